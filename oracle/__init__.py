"""CPU oracle bindings (TEST INFRASTRUCTURE ONLY).

ctypes wrapper over ``oracle/libsnporacle.so`` (built from ``snp_oracle.c`` by
``oracle/Makefile``).  Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and
``__graft_entry__.smoke()`` may import this package; the product path
(``snprelate_amd``) never does.

Pinning: every restatement is checked against the reference's own golden vectors in
tests/test_oracle_golden.py, except ``pca_randomized`` -- PARITY UNPINNED (the reference's tests hold no
golden for algorithm="randomized").

Genotypes are ``uint8 [L][N]`` (SNP-major, sample fastest, >2 = missing), the
layout ``CGenoReadBySNP::Read`` produces in the reference
(src/dGenGWAS.cpp:1218-1397).  Triangles are packed row-major upper with
diagonal (``CdMatTri``, src/dGenGWAS.h:511-583).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsnporacle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (used by __graft_entry__.build())."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "snp_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsnporacle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        # idle OpenMP workers sleep instead of spinning: the GPU boxes are shared (256 hardware threads, load average
        # 58 observed) and a spinning 256-thread team made a 0.1 s oracle call take 6 s
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = ctypes.CDLL(_LIB_PATH)
        i64, vp, dbl, c_int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_double, ctypes.c_int
        L.orc_snp_stats.argtypes = [vp, i64, i64, vp, vp]
        L.orc_select_snp_base.argtypes = [vp, i64, i64, c_int, dbl, dbl, vp]
        L.orc_select_snp_base.restype = c_int
        L.orc_ibs_count.argtypes = [vp, i64, i64, vp]
        L.orc_ibs_ave.argtypes = [vp, i64, vp]
        L.orc_king_robust_count.argtypes = [vp, i64, i64, vp]
        L.orc_king_robust_final.argtypes = [vp, i64, vp, vp, vp]
        L.orc_king_homo_count.argtypes = [vp, i64, i64, vp, vp]
        L.orc_king_homo_final.argtypes = [vp, vp, i64, vp, vp]
        L.orc_pca_cov.argtypes = [vp, i64, i64, c_int, vp]
        L.orc_trace_normalize.argtypes = [vp, i64]
        L.orc_trace_normalize.restype = dbl
        L.orc_grm_gcta.argtypes = [vp, i64, i64, vp]
        L.orc_mom_expect.argtypes = [vp, i64, i64, vp, vp, vp]
        L.orc_mom_final.argtypes = [vp, i64, vp, c_int, vp, vp]
        L.orc_beta_count.argtypes = [vp, i64, i64, vp]
        L.orc_beta_final_ibd.argtypes = [vp, i64, c_int, vp]
        L.orc_beta_final_ibd.restype = dbl
        L.orc_beta_final_grm.argtypes = [vp, i64, vp]
        L.orc_beta_final_grm.restype = dbl
        L.orc_eigmix.argtypes = [vp, i64, i64, c_int, vp, vp]
        L.orc_tri_to_full_f64.argtypes = [vp, i64, vp]
        L.orc_synth_hash_geno.argtypes = [vp, i64, i64, i64, ctypes.c_uint32, ctypes.c_uint32, c_int, c_int, vp]
        L.orc_num_threads.restype = c_int
        L.orc_set_num_threads.argtypes = [c_int]
        L.orc_set_num_threads(min(os.cpu_count() or 1, 32))       # DEFAULT_THREADS; cpu_baseline raises it explicitly
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _geno(g):
    g = np.ascontiguousarray(g, dtype=np.uint8)
    assert g.ndim == 2
    return g


DEFAULT_THREADS = 32     # checker runs (tests, smoke): plenty for the test sizes, robust on a loaded host


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def host_threads():
    """all hardware threads of this host (bench.py's cpu_baseline asks for them explicitly)"""
    return os.cpu_count() or 1


def num_threads():
    return lib().orc_num_threads()


def tri_size(n):
    return n * (n + 1) // 2


def tri_to_full(tri, n):
    """Packed upper triangle -> full symmetric matrix (any dtype, numpy)."""
    tri = np.asarray(tri)
    full = np.empty((n, n), dtype=tri.dtype)
    iu = np.triu_indices(n)
    full[iu] = tri
    full.T[iu] = tri
    return full


def snp_stats(g):
    g = _geno(g)
    L, N = g.shape
    s = np.empty(L, np.int32)
    c = np.empty(L, np.int32)
    lib().orc_snp_stats(_p(g), L, N, _p(s), _p(c))
    return s, c


def select_snp_base(g, remove_mono=True, maf=float("nan"), missing_rate=float("nan")):
    """gnrSelSNP_Base with the NaN -> (-1, 2) mapping of R/Internal.R:438-439."""
    g = _geno(g)
    L, N = g.shape
    if not np.isfinite(maf):
        maf = -1.0
    if not np.isfinite(missing_rate):
        missing_rate = 2.0
    sel = np.empty(L, np.uint8)
    lib().orc_select_snp_base(_p(g), L, N, int(bool(remove_mono)), float(maf),
                              float(missing_rate), _p(sel))
    return sel.astype(bool)


def ibs_count(g):
    """-> uint32 [npair, 3] (IBS0, IBS1, IBS2)."""
    g = _geno(g)
    L, N = g.shape
    out = np.empty((tri_size(N), 3), np.uint32)
    lib().orc_ibs_count(_p(g), L, N, _p(out))
    return out


def ibs_ave(cnt, n):
    out = np.empty(tri_size(n), np.float64)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    with np.errstate(all="ignore"):
        lib().orc_ibs_ave(_p(cnt), n, _p(out))
    return out


def king_robust_count(g):
    """-> uint32 [npair, 5] (IBS0, nLoci, SumSq, N1_Aa, N2_Aa)."""
    g = _geno(g)
    L, N = g.shape
    out = np.empty((tri_size(N), 5), np.uint32)
    lib().orc_king_robust_count(_p(g), L, N, _p(out))
    return out


def king_robust_final(cnt, n, family=None):
    cnt = np.ascontiguousarray(cnt, np.uint32)
    ibs0 = np.empty(tri_size(n), np.float64)
    kin = np.empty(tri_size(n), np.float64)
    fam = None
    if family is not None:
        fam = np.ascontiguousarray(family, np.int32)
    lib().orc_king_robust_final(_p(cnt), n, _p(fam) if fam is not None else None,
                                _p(ibs0), _p(kin))
    return ibs0, kin


def king_homo_count(g):
    g = _geno(g)
    L, N = g.shape
    cnt = np.empty((tri_size(N), 2), np.uint32)
    fs = np.empty((tri_size(N), 2), np.float64)
    lib().orc_king_homo_count(_p(g), L, N, _p(cnt), _p(fs))
    return cnt, fs


def king_homo_final(cnt, fs, n):
    cnt = np.ascontiguousarray(cnt, np.uint32)
    fs = np.ascontiguousarray(fs, np.float64)
    k0 = np.empty(tri_size(n), np.float64)
    k1 = np.empty(tri_size(n), np.float64)
    lib().orc_king_homo_final(_p(cnt), _p(fs), n, _p(k0), _p(k1))
    return k0, k1


def pca_cov(g, bayesian=False):
    """Raw covariance numerator (before the (N-1)/trace scaling), packed."""
    g = _geno(g)
    L, N = g.shape
    out = np.empty(tri_size(N), np.float64)
    lib().orc_pca_cov(_p(g), L, N, int(bool(bayesian)), _p(out))
    return out


def trace_normalize(cov_tri, n):
    """In place C *= (N-1)/trace; returns TraceXTX."""
    assert cov_tri.dtype == np.float64 and cov_tri.flags.c_contiguous
    return lib().orc_trace_normalize(_p(cov_tri), n)


def grm_gcta(g):
    g = _geno(g)
    L, N = g.shape
    out = np.empty(tri_size(N), np.float64)
    lib().orc_grm_gcta(_p(g), L, N, _p(out))
    return out


def mom_expect(g, in_afreq=None):
    """-> (e[5] = E00,E01,E02,E11,E12, afreq[L])  (Init_EPrIBD_IBS)."""
    g = _geno(g)
    L, N = g.shape
    e = np.empty(5, np.float64)
    af = np.empty(L, np.float64)
    ia = None if in_afreq is None else np.ascontiguousarray(in_afreq, np.float64)
    with np.errstate(all="ignore"):
        lib().orc_mom_expect(_p(g), L, N, _p(ia) if ia is not None else None, _p(e), _p(af))
    return e, af


def mom_final(ibs_cnt, n, e, constraint=False):
    cnt = np.ascontiguousarray(ibs_cnt, np.uint32)
    k0 = np.empty(tri_size(n), np.float64)
    k1 = np.empty(tri_size(n), np.float64)
    e = np.ascontiguousarray(e, np.float64)
    lib().orc_mom_final(_p(cnt), n, _p(e), int(bool(constraint)), _p(k0), _p(k1))
    return k0, k1


def beta_count(g):
    """-> uint32 [npair, 2] (ibscnt, num)."""
    g = _geno(g)
    L, N = g.shape
    out = np.empty((tri_size(N), 2), np.uint32)
    lib().orc_beta_count(_p(g), L, N, _p(out))
    return out


def beta_final_ibd(cnt, n, inbreeding=True):
    cnt = np.ascontiguousarray(cnt, np.uint32)
    out = np.empty(tri_size(n), np.float64)
    avg = lib().orc_beta_final_ibd(_p(cnt), n, int(bool(inbreeding)), _p(out))
    return out, avg


def beta_final_grm(cnt, n):
    cnt = np.ascontiguousarray(cnt, np.uint32)
    out = np.empty(tri_size(n), np.float64)
    avg = lib().orc_beta_final_grm(_p(cnt), n, _p(out))
    return out, avg


def eigmix(g, diagadj=True):
    """-> (ibd packed triangle, afreq[L])."""
    g = _geno(g)
    L, N = g.shape
    out = np.empty(tri_size(N), np.float64)
    af = np.empty(L, np.float64)
    lib().orc_eigmix(_p(g), L, N, int(bool(diagadj)), _p(out), _p(af))
    return out, af


# ---------------------------------------------------------------------------
# PCA projections (numpy restatements; small inputs only)

def pca_snp_corr(g, eigvec):
    """CPCA_SNPCorr::SNP_PC_Corr / thread_corr (src/genPCA.cpp:822-858): Pearson correlation of every SNP's
    genotypes (0/1/2, missing skipped) with each eigenvector.  g uint8 [L][n], eigvec [k][n] -> [L][k]
    (R: k x L column-major); NaN when fewer than 2 calls or a zero variance."""
    g = np.asarray(g)
    e = np.asarray(eigvec, dtype=np.float64)
    v = (g < 3).astype(np.float64)                      # [L][n]
    y = np.where(g < 3, g, 0).astype(np.float64)
    m = v.sum(axis=1)[:, None]                          # [L][1]
    XY = y @ e.T                                        # [L][k]
    X = v @ e.T
    XX = v @ (e * e).T
    Y = y.sum(axis=1)[:, None]
    YY = (y * y).sum(axis=1)[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        c1 = XX - X * X / m
        c2 = YY - Y * Y / m
        val = c1 * c2
        out = (XY - X * Y / m) / np.sqrt(val)
    out[~((m > 1) & (val > 0))] = np.nan
    return out


def pca_snp_loading(g, eigenval, eigenvect, trace_xtx, bayesian=False):
    """gnrPCASNPLoading + CPCA_SNPLoad::thread_loading (src/genPCA.cpp:1488-1531, 950-998).
    eigenvect [k][n]; returns loading [L][k] (R: k x L), avgfreq [L], scale [L]."""
    g = np.asarray(g)
    n = g.shape[1]
    ev = np.asarray(eigenvect, dtype=np.float64) * np.sqrt((n - 1) / trace_xtx / np.asarray(eigenval, dtype=np.float64))[:, None]
    valid = g < 3
    gsum = np.where(valid, g, 0).sum(axis=1).astype(np.float64)
    gnum = valid.sum(axis=1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = np.where(gnum > 0, gsum / gnum, 0.0)
        if not bayesian:
            p = avg * 0.5
            scale = np.where((0.0 < p) & (p < 1.0), 1.0 / np.sqrt(p * (1.0 - p)), 0.0)
        else:
            p = (gsum + 1) / (2 * gnum + 2)
            scale = 1.0 / np.sqrt(p * (1.0 - p))
        scale = np.where(gnum > 0, scale, 0.0)
    z = np.where(valid, (g.astype(np.float64) - avg[:, None]) * scale[:, None], 0.0)
    return z @ ev.T, avg, scale


def pca_samp_loading(g, sload, avgfreq, scale):
    """CPCA_SampleLoad::thread_loading (src/genPCA.cpp:1046-1068): sload [L][k] (already multiplied by
    sqrt(ss / eigenval) as R/PCA.R:283-285 does) -> sample eigenvectors [k][n] (R: n x k column-major)."""
    g = np.asarray(g)
    z = np.where(g < 3, (g.astype(np.float64) - np.asarray(avgfreq)[:, None]) * np.asarray(scale)[:, None], 0.0)
    return (z.T @ np.asarray(sload, dtype=np.float64)).T


def pca_randomized(g, aux_mat, iter_num):
    """CRandomPCA::Run (src/genPCA.cpp:672-792) restated with numpy / LAPACK (np.linalg.svd = dgesvd):
    g uint8 [L][n]; aux_mat [aux_dim][n] (R: rnorm(aux.dim * n.samp) as the C code reads it).
    Returns (sigma [min(hsize, n)], vt [min(hsize, n)][n], 2 * TraceXTX).
    The reference's tests hold no golden for algorithm = "randomized"; pinned through the exact PCA they do fix: on the call of
    Validate.PCA.RData (HapMap's first 90 samples) iter.num = 10 reproduces the golden genmat's top-4 eigenpairs (subspace
    6e-6 rad, eigenvalues 5e-11; tests/test_oracle_golden.py::test_pca_randomized_pinned_by_the_exact_pca_golden)."""
    g = np.asarray(g)
    L, n = g.shape
    valid = g < 3
    gsum = np.where(valid, g, 0).sum(axis=1).astype(np.float64)
    gnum = valid.sum(axis=1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        avg = np.where(gnum > 0, gsum / gnum, 0.0)
        p = avg * 0.5
        s = np.where((0 < p) & (p < 1), 1.0 / np.sqrt(2 * p * (1 - p)), 0.0)      # :507-510
    Y = np.where(valid, (g.astype(np.float64) - avg[:, None]) * s[:, None], 0.0)     # [L][n]
    trace = float((Y * Y).sum())
    G = np.asarray(aux_mat, dtype=np.float64).T.copy()                               # [n][aux_dim]
    A = G.shape[1]
    H = np.empty((L, A * (iter_num + 1)))
    for it in range(iter_num + 1):
        H[:, A * it: A * (it + 1)] = Y @ G                                           # :528-579
        if it < iter_num:
            G = Y.T @ H[:, A * it: A * (it + 1)] / L                                 # :581-613, 750
    _, _, vt = np.linalg.svd(H.T, full_matrices=False)                               # :757 (hsize x L)
    T = vt @ Y                                                                       # :763-781
    _, sig, vt2 = np.linalg.svd(T, full_matrices=False)                              # :783-784
    return sig, vt2, 2 * trace


def eigmix_snp_loading(g, eigenval, eigenvect, afreq):
    """gnrEigMixSNPLoading + CEigMix_SNPLoad::thread_loading (src/genEIGMIX.cpp:739-775, 440-470): eigenvect [k][n]
    -> loading [L][k].  PARITY UNPINNED (no golden in the reference's tests)."""
    g = np.asarray(g)
    af = np.asarray(afreq, dtype=np.float64)
    sc = 1.0 / np.sqrt(np.sum(4 * af * (1 - af)))
    ev = np.asarray(eigenvect, dtype=np.float64) * np.sqrt(1 / np.asarray(eigenval, dtype=np.float64))[:, None]
    z = np.where(g < 3, (g.astype(np.float64) - 2 * af[:, None]) * sc, 0.0)
    return z @ ev.T


def eigmix_samp_loading(g, sload, afreq):
    """gnrEigMixSampLoading + CEigMix_SampleLoad::thread_loading (src/genEIGMIX.cpp:777-803, 540-565): sload [L][k]
    (already multiplied by sqrt(1 / eigenval), R/PCA.R:289-290) -> [k][n].  PARITY UNPINNED."""
    g = np.asarray(g)
    af = np.asarray(afreq, dtype=np.float64)
    sc = 1.0 / np.sqrt(np.sum(4 * af * (1 - af)))
    z = np.where(g < 3, (g.astype(np.float64) - 2 * af[:, None]) * sc, 0.0)
    return (z.T @ np.asarray(sload, dtype=np.float64)).T


def grm_merge(grms, weight, cmd=":method = GCTA", avg_val=None):
    """gnrGRMMerge, src/genPCA.cpp:1721-1853: weighted combination of full N x N GRMs.  Returns (merged, avg_val)
    (avg_val = None unless cmd is the IndivBeta one).  Pinned by the reference's own merge property
    (inst/unitTests/test_GRM.R:14-87: merged GRM of a SNP partition == GRM of the whole set)."""
    grms = [np.asarray(g, np.float64) for g in grms]
    n = grms[0].shape[0]
    off = ~np.eye(n, dtype=bool)
    if cmd != ":method = IndivBeta":                       # :1835-1851
        out = np.zeros((n, n))
        for g, w in zip(grms, weight):
            out += w * g
        return out, None
    m = np.zeros((n, n))
    for g, w, a in zip(grms, weight, avg_val):             # :1758-1796
        mb = g[off].sum() / (n * (n - 1)) * 0.5
        mij = (g * 0.5 - mb) / (1 - mb) * (1 - a) + a
        dij = (np.diag(g) - 1 - mb) / (1 - mb) * (1 - a) + a
        mij[np.arange(n), np.arange(n)] = dij
        m += mij * w
    avg = m[off].sum() / (n * (n - 1))                     # :1798-1809
    mn = m.min()
    out = (m - mn) * (2 / (1 - mn))                        # :1811-1819
    out[np.arange(n), np.arange(n)] = out[np.arange(n), np.arange(n)] * 0.5 + 1
    return out, avg


def synth_hash_geno_c(samples, snp_begin, n_snp, seed, missing=0.0, spectrum=0, special=False):
    """C twin (OpenMP) of oracle.synth.synth_hash_geno for spectra 0, 1, 2 -- bit-identical (tests/test_cpu_host.py); the structured
    spectra 3 and 4 fall back to the numpy form."""
    if spectrum not in (0, 1, 2):
        from .synth import synth_hash_geno
        return synth_hash_geno(samples, snp_begin, n_snp, seed, missing, spectrum, special)
    samples = np.ascontiguousarray(samples, dtype=np.int64)
    out = np.empty((int(n_snp), samples.size), np.uint8)
    miss32 = int(np.floor(missing * 4294967296.0))
    lib().orc_synth_hash_geno(_p(samples), samples.size, int(snp_begin), int(n_snp), int(seed) & 0xFFFFFFFF, miss32, int(spectrum),
                              int(bool(special)), _p(out))
    return out
