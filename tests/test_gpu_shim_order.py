"""The R-side binding (r_shim/gpu_shim.cpp) cannot be compiled here (no R, no gdsfmt): these tests replay, through
ctypes, exactly the libsnpgpu call sequence of each of its seven `.Call` routines -- same order, same memory kinds
(two page-locked reader buffers, SNPGPU_HOST_PINNED feeds of uint8 blocks, results written into caller-owned
column-major n x n / packed buffers of R's element types) -- and check the results against the reference's golden
vectors (tests/golden) and the CPU oracle."""
import ctypes
import os

import numpy as np
import pytest

import oracle as orc
from conftest import GOLDEN, synth_geno

pytestmark = pytest.mark.gpu
NA_INTEGER = -2147483648


class ShimAccumulator:
    """`struct Accumulator` of r_shim/gpu_shim.cpp: stream() = the kept reader loop, blocks of `block_snps`."""

    def __init__(self):
        from snprelate_amd import _lib
        self.L, self._lib = _lib.lib(), _lib
        self.ctx = ctypes.c_void_p()
        self.blk = [ctypes.c_void_p(), ctypes.c_void_p()]

    def stream(self, kind, bayesian, block_snps, g):
        L, lib = self.L, self._lib
        n_snp, n = g.shape
        o = lib.Opts(0, int(bayesian), 0, 0, int(block_snps), None)
        lib.check(L.snpgpu_create(int(kind), n, ctypes.byref(o), ctypes.byref(self.ctx)))
        for k in range(2):
            lib.check(L.snpgpu_host_alloc(n * block_snps, ctypes.byref(self.blk[k])))
        views = [np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)), shape=(block_snps * n,))
                 for b in self.blk]
        at, k = 0, 0
        while True:
            lib.check(L.snpgpu_host_wait(self.ctx, self.blk[k]))
            cnt = min(block_snps, n_snp - at)                 # reader.Read() / reader.Count()
            if cnt <= 0:
                break
            views[k][: cnt * n] = g[at:at + cnt].ravel()
            lib.check(L.snpgpu_feed(self.ctx, self.blk[k], cnt, lib.GENO_U8, lib.HOST_PINNED))
            at += cnt
            k ^= 1
        lib.check(L.snpgpu_sync(self.ctx))

    def close(self):
        if self.ctx:
            self.L.snpgpu_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()
        for k in range(2):
            if self.blk[k]:
                self.L.snpgpu_host_free(self.blk[k])
                self.blk[k] = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _alloc_result(n, packed):          # alloc_result(): REALSXP matrix (column-major) or packed numeric vector
    return np.empty(n * (n + 1) // 2) if packed else np.empty((n, n), order="F")


def _hapmap_block(hapmap, n_first):
    """genotypes .InitFile2 selects for the reference's golden tests: first n samples, autosomes, polymorphic"""
    from snprelate_amd import api
    from snprelate_amd.gds import unpack_2bit_rows
    ws = api._init_file2(None, hapmap, hapmap.sample_id[:n_first], None, missing_rate=float("nan"), verbose=False)
    return unpack_2bit_rows(ws["packed"], ws["n_samp"]), ws


def test_shim_gnrIBSNum_and_gnrIBSAve(hapmap):
    from snprelate_amd import _lib
    g, ws = _hapmap_block(hapmap, 90)
    n = g.shape[1]
    z = np.load(os.path.join(GOLDEN, "validate_ibs.npz"))
    with ShimAccumulator() as acc:
        acc.stream(_lib.IBS, False, 65536, g)
        m = [np.empty((n, n), dtype=np.int32, order="F") for _ in range(3)]          # three INTSXP matrices
        _lib.check(acc.L.snpgpu_ibs_num(acc.ctx, _p(m[0]), _p(m[1]), _p(m[2]), 0, _lib.HOST))
        assert (m[0][0, 1], m[1][0, 1], m[2][0, 1]) == (447, 3160, 5050)
        assert all(np.array_equal(x, x.T) for x in m)
        cnt = orc.ibs_count(g)
        assert np.array_equal(orc.tri_to_full(cnt[:, 0].astype(np.int32), n), m[0])
        for packed in (False, True):                                                 # gnrIBSAve on the same context
            out = _alloc_result(n, packed)
            _lib.check(acc.L.snpgpu_ibs_ave(acc.ctx, _p(out), int(packed), _lib.HOST))
            assert np.array_equal(orc.tri_to_full(out, n) if packed else out, z["ibs"])


def test_shim_gnrIBD_KING_Robust_and_Homo(hapmap):
    from snprelate_amd import _lib
    g, ws = _hapmap_block(hapmap, 60)
    n = g.shape[1]
    z = np.load(os.path.join(GOLDEN, "validate_king.npz"))
    fam = np.full(n, NA_INTEGER, dtype=np.int32)                                     # family.id = NULL -> all NA
    with ShimAccumulator() as acc:
        acc.stream(_lib.KING_ROBUST, False, 4096, g)                                 # several blocks
        for packed in (False, True):
            a, b = _alloc_result(n, packed), _alloc_result(n, packed)
            _lib.check(acc.L.snpgpu_king_robust(acc.ctx, _p(fam), _p(a), _p(b), int(packed), _lib.HOST))
            assert np.array_equal(orc.tri_to_full(a, n) if packed else a, z["robust_IBS0"])
            assert np.array_equal(orc.tri_to_full(b, n) if packed else b, z["robust_kinship"])
        fam2 = np.arange(n, dtype=np.int32) // 3 + 1                                 # as.integer(as.factor(.)): levels 1..k
        fam2[::7] = NA_INTEGER
        a, b = _alloc_result(n, True), _alloc_result(n, True)
        _lib.check(acc.L.snpgpu_king_robust(acc.ctx, _p(fam2), _p(a), _p(b), 1, _lib.HOST))
        r0, rk = orc.king_robust_final(orc.king_robust_count(g), n, np.where(fam2 == NA_INTEGER, -1, fam2))
        assert np.array_equal(a, r0, equal_nan=True) and np.array_equal(b, rk, equal_nan=True)
    with ShimAccumulator() as acc:
        acc.stream(_lib.KING_HOMO, False, 16384, g)
        k0, k1 = _alloc_result(n, False), _alloc_result(n, False)
        _lib.check(acc.L.snpgpu_king_homo(acc.ctx, _p(k0), _p(k1), 0, _lib.HOST))
        np.testing.assert_allclose(k0, z["homo_k0"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(k1, z["homo_k1"], rtol=1e-5, atol=2e-5)


def _append_rows(tri, n):
    """append_rows() of the shim: row i of the symmetric matrix from the packed upper triangle"""
    rows = []
    for i in range(n):
        row = np.empty(n)
        for k in range(i):
            row[k] = tri[i + k * (2 * n - k - 1) // 2]
        p = i + i * (2 * n - i - 1) // 2
        row[i:] = tri[p:p + n - i]
        rows.append(row)
    return np.stack(rows)


@pytest.mark.parametrize("method", ["GCTA", "Eigenstrat", "Corr", "EIGMIX", "IndivBeta"])
def test_shim_gnrGRM_all_methods_and_out_gds(method):
    from norms import error_figures, tri_diag_scale
    from snprelate_amd import _lib
    n, L = 333, 3000
    g = synth_geno(n, L, missing=0.03, seed=5)
    kind = {"GCTA": _lib.GRM_GCTA, "Corr": _lib.GRM_GCTA, "Eigenstrat": _lib.PCA_COV, "EIGMIX": _lib.EIGMIX,
            "IndivBeta": _lib.INDIV_BETA}[method]
    if method in ("GCTA", "Corr"):
        ref = orc.grm_gcta(g)
    elif method == "Eigenstrat":
        ref = orc.pca_cov(g)
        orc.trace_normalize(ref, n)
    elif method == "EIGMIX":
        ref = 2 * orc.eigmix(g, False)[0]
    else:
        ref, ref_avg = orc.beta_final_grm(orc.beta_count(g), n)
    ref_full = orc.tri_to_full(ref, n)
    if method == "Corr":
        sd = np.sqrt(np.diag(ref_full))
        ref_full = ref_full / np.outer(sd, sd)
        np.fill_diagonal(ref_full, 1.0)
    avg = ctypes.c_double(0)

    def finalise(acc, out, packed):
        L_ = acc.L
        if kind == _lib.GRM_GCTA:
            rc = L_.snpgpu_grm_gcta(acc.ctx, _p(out), packed, _lib.HOST)
        elif kind == _lib.PCA_COV:
            rc = L_.snpgpu_pca_cov(acc.ctx, _p(out), packed, 1, 0.0, None, _lib.HOST)
        elif kind == _lib.EIGMIX:
            rc = L_.snpgpu_eigmix(acc.ctx, 0, 2.0, _p(out), packed, _lib.HOST)
        else:
            rc = L_.snpgpu_indiv_beta(acc.ctx, 2, _p(out), ctypes.byref(avg), packed, _lib.HOST)
        _lib.check(rc)

    with ShimAccumulator() as acc:
        acc.stream(kind, False, 65536 if kind == _lib.INDIV_BETA else 1024, g)
        full = _alloc_result(n, False)
        finalise(acc, full, 0)
        if method == "Corr":                                  # the host loop of the shim
            sd = np.sqrt(np.diag(full).copy())
            for i in range(n):
                full[i, i] = 1
                full[i, i + 1:] = full[i + 1:, i] = full[i + 1:, i] / (sd[i] * sd[i + 1:])
        tol = 1e-10 if method == "IndivBeta" else 1e-5
        f = error_figures(full, ref_full, float(np.median(np.abs(np.diag(ref_full)))))
        assert f["contract"] < tol and (method in ("Corr",) or f["offdiag"] < tol), f
        assert np.array_equal(full, full.T)
        if method != "Corr":
            tri = _alloc_result(n, True)                      # useMatrix = TRUE / the out.gds path
            finalise(acc, tri, 1)
            assert np.array_equal(_append_rows(tri, n), full)                 # rows appended to the "grm" node
        if method == "IndivBeta":
            assert abs(avg.value - ref_avg) < 1e-12                           # gnrGRM_avg_val


def test_shim_gnrPCA_exact(hapmap):
    from snprelate_amd import _lib
    g, ws = _hapmap_block(hapmap, 90)
    n = g.shape[1]
    z = np.load(os.path.join(GOLDEN, "validate_pca.npz"))
    with ShimAccumulator() as acc:
        acc.stream(_lib.PCA_COV, False, 16384, g)
        genmat = np.empty((n, n), order="F")
        tr = ctypes.c_double(0)
        _lib.check(acc.L.snpgpu_pca_cov(acc.ctx, _p(genmat), 0, 1, 0.0, ctypes.byref(tr), _lib.HOST))
        scale = np.abs(z["genmat"]).mean()
        assert np.max(np.abs(genmat - z["genmat"]) / (np.abs(z["genmat"]) + scale)) < 1e-5
        assert abs(np.trace(genmat) - (n - 1)) < 1e-9 * n                     # TraceVal
        # need.genmat = FALSE: only the trace
        tr2 = ctypes.c_double(0)
        _lib.check(acc.L.snpgpu_pca_cov(acc.ctx, None, 0, 1, 0.0, ctypes.byref(tr2), _lib.HOST))
        assert tr2.value == tr.value
        w, v = np.linalg.eigh(z["genmat"])
        w, v = w[::-1], v[:, ::-1]
        for k, n_eig in ((8, 8), (n, 5)):                                     # DSPEVX: k = eigen.cnt; DSPEV: all values
            val = np.empty(n)
            vec = np.empty((n, k), order="F")
            _lib.check(acc.L.snpgpu_pca_eigen(acc.ctx, k, _p(val), _p(vec), _lib.HOST))
            val[k:] = np.nan
            np.testing.assert_allclose(val[:min(k, 40)], w[:min(k, 40)], rtol=1e-5)
            cos = np.abs(np.sum(vec[:, :n_eig] * v[:, :n_eig], axis=0))
            assert np.all(cos > 1 - 1e-6)


def test_shim_gnrPCA_randomized():
    from snprelate_amd import _lib
    from snprelate_amd.gds import pack_2bit_rows
    from test_gpu_api_golden import _structured_geno
    n, L, n_eig, aux_dim, iter_num = 150, 2500, 4, 8, 5
    g = _structured_geno(n, L, seed=n)
    g[g > 2] = 3
    rng = np.random.default_rng(3)
    packed = pack_2bit_rows(g)                                                # what the shim builds from the reader's blocks
    aux = rng.normal(size=aux_dim * n)
    Lb = _lib.lib()
    try:
        _lib.check(Lb.snpgpu_ws_set_geno(_p(packed), L, n, _lib.GENO_PACKED2, 0))
        sigma, vecs, tr2 = np.empty(n), np.empty((n, n_eig), order="F"), ctypes.c_double(0)
        _lib.check(Lb.snpgpu_gnrPCA_randomized(n_eig, aux_dim, iter_num, _p(aux), 1, 0, _p(sigma), _p(vecs), ctypes.byref(tr2)))
    finally:
        Lb.snpgpu_ws_clear()
    hsize = aux_dim * (iter_num + 1)
    vt = np.zeros((hsize, n), order="F")
    vt[:n_eig, :] = vecs.T                                                    # first eigen.cnt rows of V^T
    r_sig, r_vt, r_tr2 = orc.pca_randomized(g, aux.reshape(aux_dim, n), iter_num)
    np.testing.assert_allclose(sigma[:n_eig], r_sig[:n_eig], rtol=1e-6)
    assert abs(tr2.value - r_tr2) < 1e-9 * r_tr2
    cos = np.abs(np.sum(vt[:n_eig] * r_vt[:n_eig], axis=1))
    assert np.all(cos[:2] > 1 - 1e-6)


def test_shim_gnrPCA_exact_60000_samples_through_the_abi():
    """gnrPCA's covariance + CalcEigen (src/genPCA.cpp:1262-1346, :1355-1452) beyond any dense solver, through the C ABI
    only (what the R shim binds; no torch, no Python algebra): 60 000 samples, 2048 SNPs generated on the device, the
    28.8 GB panel stays resident and snpgpu_pca_eigen runs the block-Krylov solver of csrc/eigen.hip on it.
    Checks: the eight largest eigenvalues against the eigenvalues of the 2048 x 2048 DUAL matrix Z Z^T (n-1)/trace
    computed in fp64 numpy from the CPU twin of the generator; the residuals |C v - lambda v| with the panel product;
    orthonormality."""
    import torch                                                             # device buffers only
    from oracle.synth import synth_hash_geno
    from snprelate_amd import _lib
    L_ = _lib.lib()
    n, n_snp, blk, k, seed = 60000, 2048, 1024, 8, 20240601
    ctx = ctypes.c_void_p()
    o = _lib.Opts(0, 0, 0, 0, blk, None)
    _lib.check(L_.snpgpu_create(_lib.PCA_COV, n, ctypes.byref(o), ctypes.byref(ctx)))
    try:
        buf = torch.empty((blk, (n + 3) // 4), dtype=torch.uint8, device="cuda")
        for lo in range(0, n_snp, blk):
            _lib.check(L_.snpgpu_synth_block(ctypes.c_void_p(buf.data_ptr()), n, lo, blk, seed, 0.01, 0, 0, 0, None))
            _lib.check(L_.snpgpu_feed(ctx, ctypes.c_void_p(buf.data_ptr()), blk, _lib.GENO_PACKED2, _lib.DEVICE))
        _lib.check(L_.snpgpu_sync(ctx))
        val = np.empty(k)
        vec = np.empty((n, k), order="F")
        _lib.check(L_.snpgpu_pca_eigen(ctx, k, _p(val), _p(vec), _lib.HOST))
        tr = ctypes.c_double(0)
        _lib.check(L_.snpgpu_pca_panel_trace(ctx, ctypes.byref(tr)))
        # residuals through the panel product
        q = torch.from_numpy(np.ascontiguousarray(vec.T)).cuda()             # [k][n]
        y = torch.zeros_like(q)
        torch.cuda.synchronize()
        _lib.check(L_.snpgpu_pca_panel_matmul(ctx, (n - 1) / tr.value, ctypes.c_void_p(q.data_ptr()), k,
                                              ctypes.c_void_p(y.data_ptr())))
        res = (y - torch.from_numpy(val).cuda()[:, None] * q).norm(dim=1).cpu().numpy()
        assert np.all(res / val < 1e-8), res / val
        assert np.allclose(vec.T @ vec, np.eye(k), atol=1e-9)
    finally:
        L_.snpgpu_destroy(ctx)
    # the dual problem in fp64 on the CPU
    z = np.empty((n_snp, n))
    for lo in range(0, n_snp, 256):
        g = synth_hash_geno(np.arange(n), lo, 256, seed, missing=0.01)
        valid = g <= 2
        s, c = (g * valid).sum(1, dtype=np.int64).astype(np.float64), valid.sum(1).astype(np.float64)
        avg = s / c
        p = avg / 2
        z[lo:lo + 256] = np.where(valid, (g - avg[:, None]) / np.sqrt(p * (1 - p))[:, None], 0.0)
    trace = float((z * z).sum())
    assert abs(trace - tr.value) < 1e-6 * trace
    w = np.linalg.eigvalsh(z @ z.T)[::-1][:k] * (n - 1) / trace
    np.testing.assert_allclose(val, w, rtol=2e-6)


def test_shim_multi_device_gnrPCA_and_king_sequences(hapmap):
    """`MultiAccumulator` of r_shim/gpu_shim.cpp (options(snpgpu.devices=) with more than one entry): snpgpu_multi_create ->
    two page-locked reader buffers -> {snpgpu_multi_host_wait; Read; snpgpu_multi_feed(U8, HOST_PINNED)} per block ->
    snpgpu_multi_sync -> snpgpu_multi_pca_cov (trace / packed matrix) -> snpgpu_multi_topk_eigen, and the KING-robust
    routine with two passes gathering into one pair of packed vectors.  Two "devices" on the one test GPU."""
    from snprelate_amd import _lib
    L_ = _lib.lib()
    g, ws = _hapmap_block(hapmap, 90)
    n_snp, n = g.shape
    z = np.load(os.path.join(GOLDEN, "validate_pca.npz"))
    devs = (ctypes.c_int32 * 2)(0, 0)

    def stream(kind, block_snps, n_passes=1, q=0, ppd=2):
        m = ctypes.c_void_p()
        o = _lib.Opts(0, 0, 0, 0, int(block_snps), None)
        mo = _lib.MultiOpts(devs, 2, ppd, n_passes, q)
        _lib.check(L_.snpgpu_multi_create(int(kind), n, ctypes.byref(o), ctypes.byref(mo), ctypes.byref(m)))
        blk = [ctypes.c_void_p(), ctypes.c_void_p()]
        for k in range(2):
            _lib.check(L_.snpgpu_host_alloc(n * block_snps, ctypes.byref(blk[k])))
        views = [np.ctypeslib.as_array(ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)), shape=(block_snps * n,)) for b in blk]
        at, k = 0, 0
        while True:
            _lib.check(L_.snpgpu_multi_host_wait(m, blk[k]))
            cnt = min(block_snps, n_snp - at)
            if cnt <= 0:
                break
            views[k][: cnt * n] = g[at:at + cnt].ravel()
            _lib.check(L_.snpgpu_multi_feed(m, blk[k], cnt, _lib.GENO_U8, _lib.HOST_PINNED))
            at += cnt
            k ^= 1
        _lib.check(L_.snpgpu_multi_sync(m))
        return m, blk

    def done(m, blk):
        L_.snpgpu_multi_destroy(m)
        for b in blk:
            L_.snpgpu_host_free(b)

    m, blk = stream(_lib.PCA_COV, 2048)
    try:
        tri = np.empty(n * (n + 1) // 2)
        tr = ctypes.c_double(0)
        _lib.check(L_.snpgpu_multi_pca_cov(m, _p(tri), 1, ctypes.byref(tr), _lib.HOST))
        genmat = orc.tri_to_full(tri, n)
        scale = np.abs(z["genmat"]).mean()
        assert np.max(np.abs(genmat - z["genmat"]) / (np.abs(z["genmat"]) + scale)) < 1e-5
        val, vec = np.empty(n), np.empty((n, 8), order="F")
        _lib.check(L_.snpgpu_multi_topk_eigen(m, 0.0, 8, None, _p(val), _p(vec), _lib.HOST, None))
        w, v = np.linalg.eigh(z["genmat"])
        np.testing.assert_allclose(val[:8], w[::-1][:8], rtol=1e-5)
        assert np.all(np.abs(np.sum(vec[:, :4] * v[:, ::-1][:, :4], axis=0)) > 1 - 1e-6)
    finally:
        done(m, blk)
    # KING-robust, two passes into one pair of packed vectors (first 60 samples as in the golden file's call)
    g, ws = _hapmap_block(hapmap, 60)
    n_snp, n = g.shape
    ibs0, kin = np.full(n * (n + 1) // 2, np.nan), np.full(n * (n + 1) // 2, np.nan)
    fam = np.full(n, NA_INTEGER, np.int32)
    ppd = -1            # the shim's default: pass 0 lets the library choose, the later passes get what snpgpu_multi_get_status reports
    for q in range(2):
        m, blk = stream(_lib.KING_ROBUST, 4096, 2, q, ppd)
        if q == 0:
            st = _lib.MultiStatus()
            _lib.check(L_.snpgpu_multi_get_status(m, ctypes.byref(st)))
            ppd = int(st.panels_per_device)
            assert ppd >= 1
        try:
            _lib.check(L_.snpgpu_multi_king_robust(m, _p(fam), _p(ibs0), _p(kin), _lib.HOST))
        finally:
            done(m, blk)
    r0, rk = orc.king_robust_final(orc.king_robust_count(g), n)
    # n = 60 < 256: the plan has one non-empty panel, owned by one pass; the other pass has nothing resident and must say so
    assert np.array_equal(ibs0, r0, equal_nan=True) and np.array_equal(kin, rk, equal_nan=True)
