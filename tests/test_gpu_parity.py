"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on seeded inputs.
Integer results bit-exact; floating-point within the stated tolerances."""
import numpy as np
import pytest

import oracle as orc
from conftest import synth_geno

pytestmark = pytest.mark.gpu


def _acc(kind, n, **kw):
    from snprelate_amd import _lib
    return _lib.Accumulator(kind, n, **kw)


def _feed_blocks(acc, g, block):
    for i in range(0, g.shape[0], block):
        acc.feed(g[i:i + block])


SIZES = [(37, 301, 100), (279, 1000, 333), (600, 2500, 1024), (1030, 4100, 4096)]


@pytest.fixture(params=["mfma_i8", "mfma_i8_no_fp4", "mfma_fp4_nomiss_only", "popcount"])
def pair_backend(request, monkeypatch):
    """The forms of the IBS/KING/beta counters: exact MFMA contractions (default: MX-fp4 for IBS / KING-robust / individual beta
    and GCTA's both-missing counts, int8 for KING-homo), the same with the int8 forms of all of them, with the int8 form of the
    general kernels only, and bit-plane popcounts.  The library reads the variables when a context is created."""
    if request.param == "mfma_i8_no_fp4":
        monkeypatch.setenv("SNPGPU_PAIR_BACKEND", "mfma_i8")
        monkeypatch.setenv("SNPGPU_PAIR_FP4", "0")
        monkeypatch.setenv("SNPGPU_GCTA_MISS_FP4", "0")
    elif request.param == "mfma_fp4_nomiss_only":     # int8 general kernels beside the fp4 two-product kernel
        monkeypatch.setenv("SNPGPU_PAIR_BACKEND", "mfma_i8")
        monkeypatch.setenv("SNPGPU_PAIR_FP4_GENERAL", "0")
    else:
        monkeypatch.setenv("SNPGPU_PAIR_BACKEND", request.param)
    return request.param


@pytest.mark.parametrize("missing", [0.05, 0.0])
@pytest.mark.parametrize("n,L,blk", SIZES)
def test_ibs_counts_bit_exact(n, L, blk, pair_backend, missing):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=missing, seed=n)
    if missing == 0.0 and L > 1500:
        g[L // 2 + 7, 3] = 3      # blocks with and without missing calls in one run (device-side variant choice)
    ref = orc.ibs_count(g)
    with _acc(_lib.IBS, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        i0, i1, i2 = a.ibs_num(packed=True)
        assert np.array_equal(i0, ref[:, 0].astype(np.int32))
        assert np.array_equal(i1, ref[:, 1].astype(np.int32))
        assert np.array_equal(i2, ref[:, 2].astype(np.int32))
        f0, f1, f2 = a.ibs_num(packed=False)
        assert np.array_equal(f0, orc.tri_to_full(i0, n))
        assert np.array_equal(f2, orc.tri_to_full(i2, n))
        ave = a.ibs_ave(packed=True)
        assert np.array_equal(ave, orc.ibs_ave(ref, n))


@pytest.mark.parametrize("missing", [0.05, 0.0])
@pytest.mark.parametrize("n,L,blk", SIZES)
def test_king_robust_bit_exact(n, L, blk, pair_backend, missing):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=missing, seed=n + 1)
    if missing == 0.0 and L > 1500:
        g[L // 2 + 7, 3] = 3      # blocks with and without missing calls in one run (binary 3-product kernel / general)
    ref = orc.king_robust_count(g)
    with _acc(_lib.KING_ROBUST, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        assert np.array_equal(a.king_robust_counts(), ref)
        fam = (np.arange(n) // 3).astype(np.int32)
        fam[::7] = -1
        for family in (None, fam):
            r0, rk = orc.king_robust_final(ref, n, family)
            g0, gk = a.king_robust(family=family, packed=True)
            assert np.array_equal(g0, r0, equal_nan=True)
            assert np.array_equal(gk, rk, equal_nan=True)


@pytest.mark.parametrize("n,L,blk", SIZES[:3])
def test_king_homo(n, L, blk, pair_backend, syrk_backend):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=0.05, seed=n + 2)
    c, fs = orc.king_homo_count(g)
    r0, r1 = orc.king_homo_final(c, fs, n)
    with _acc(_lib.KING_HOMO, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        k0, k1 = a.king_homo(packed=True)
    np.testing.assert_allclose(k0, r0, rtol=1e-5, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(k1, r1, rtol=1e-5, atol=2e-5, equal_nan=True)


@pytest.mark.parametrize("homo_uv,missing", [("1", 0.3), ("1", 0.002), ("0", 0.05)])
def test_king_homo_weight_sums_single_product_and_two_product_forms(homo_uv, missing, monkeypatch):
    """KING-homo's masked weight sums of blocks with missing calls (src/genKING.cpp:236-248): round 5's form -- totals - per-sample
    missing sums + ONE fp16 product of binary operands per weight (SNPGPU_HOMO_UV=1, default) -- at a high and at a tiny missing rate,
    several ragged blocks, a row panel; and the two-product form it replaced (SNPGPU_HOMO_UV=0) on the same data."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_HOMO_UV", homo_uv)
    n, L, blk = 700, 5000, 1900
    g = synth_geno(n, L, missing=missing, seed=91)
    c, fs = orc.king_homo_count(g)
    r0, r1 = orc.king_homo_final(c, fs, n)
    with _acc(_lib.KING_HOMO, n, max_block_snps=2048) as a:
        _feed_blocks(a, g, blk)
        k0, k1 = a.king_homo(packed=True)
    np.testing.assert_allclose(k0, r0, rtol=1e-5, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(k1, r1, rtol=1e-5, atol=2e-5, equal_nan=True)
    with _lib.Accumulator(_lib.KING_HOMO, n, row_begin=256, row_end=512, max_block_snps=2048) as a:
        _feed_blocks(a, g, blk)
        p0, p1 = a.king_homo(packed=True)
    lo, hi = 256 * n - 256 * 255 // 2, 512 * n - 512 * 511 // 2
    np.testing.assert_allclose(p0, r0[lo:hi], rtol=1e-5, atol=1e-7, equal_nan=True)
    np.testing.assert_allclose(p1, r1[lo:hi], rtol=1e-5, atol=2e-5, equal_nan=True)


@pytest.mark.parametrize("missing_blocks", ["none", "second", "alternate"])
def test_king_homo_blocks_without_missing_calls(missing_blocks, pair_backend):
    """KING-homo on blocks without missing calls: the masked weight sums of such a block are the same for every pair (the
    SYRK of both tables is skipped, two scalars carry them) and the counters come from the two-product kernel; blocks
    with missing calls in the same stream take the general kernels.  Counters bit-exact, k0 / k1 within the tolerance."""
    from snprelate_amd import _lib
    n, L, blk = 333, 3000, 500
    g = synth_geno(n, L, missing=0.0, seed=77, special=False)
    g[3] = 0; g[5] = 2; g[13] = 1          # monomorphic / all-het SNPs, no missing call anywhere yet
    rng = np.random.default_rng(78)
    for b in range(L // blk):
        if missing_blocks == "second" and b == 1 or missing_blocks == "alternate" and b % 2 == 1:
            sub = g[b * blk:(b + 1) * blk]
            sub[rng.random(sub.shape) < 0.04] = 3
    c, fs = orc.king_homo_count(g)
    r0, r1 = orc.king_homo_final(c, fs, n)
    with _acc(_lib.KING_HOMO, n, max_block_snps=512) as a:
        _feed_blocks(a, g, blk)
        for _ in range(2):                 # a second request after the rank-one terms were settled
            k0, k1 = a.king_homo(packed=True)
            np.testing.assert_allclose(k0, r0, rtol=1e-5, atol=1e-7, equal_nan=True)
            np.testing.assert_allclose(k1, r1, rtol=1e-5, atol=2e-5, equal_nan=True)


@pytest.fixture(params=["f16", "f16_uvc", "f16_uv16", "f16_uv32", "f16_x1", "f16_2w", "h3", "f32"])
def syrk_backend(request, monkeypatch):
    """The SYRK kernels behind GRM / PCA.  f16 (default): blocks without missing calls take the single-product kernel
    (syrk_uv16c_kernel on v_mfma_f32_16x16x32_f16; f16_uv32: syrk_uv_kernel, its 32 x 32 x 16 form: SNP weight = product of two fp16
    numbers, integer centres), blocks with missing calls the exact-row
    kernel (syrk_x1_kernel: exact row operand x hi / lo-split column operand); f16_x1: the exact-row kernel for every
    block (SNPGPU_SYRK_UV=0); f16_2w: the same arithmetic at two waves per SIMD (syrk_h3_kernel<2, true>,
    SNPGPU_SYRK_X1=0); h3: the round-1 three-product split (SNPGPU_SYRK=h3); f32: fp32 MFMAs (SNPGPU_SYRK=f32)."""
    if request.param in ("f16_uvc", "f16_uv16"):   # round 6: f16_uv16 = syrk_uv16_kernel (operands looked up in LDS tables, one launch of (tile, run)
        monkeypatch.setenv("SNPGPU_SYRK", "f16")        # items); f16_uvc = syrk_uv16c_kernel as (tile, run) items without the LDS carry (SNPGPU_SYRK_UV16=2);
        monkeypatch.setenv("SNPGPU_SYRK_UV16", "1" if request.param == "f16_uv16" else "2")      # the default f16 walks a tile's runs itself
    elif request.param == "f16_uv32":       # round 6: the 32 x 32 x 16 form of the single-product kernel (default: 16 x 16 x 32, syrk_uv16_kernel)
        monkeypatch.setenv("SNPGPU_SYRK", "f16")
        monkeypatch.setenv("SNPGPU_SYRK_UV16", "0")
    elif request.param == "f16_2w":
        monkeypatch.setenv("SNPGPU_SYRK", "f16")
        monkeypatch.setenv("SNPGPU_SYRK_X1", "0")
    elif request.param == "f16_x1":
        monkeypatch.setenv("SNPGPU_SYRK", "f16")
        monkeypatch.setenv("SNPGPU_SYRK_UV", "0")
    else:
        monkeypatch.setenv("SNPGPU_SYRK", request.param)
    return request.param


def _rel_err(got, ref, n=None):
    """Error of a packed-triangle result in the ONE norm of the floating-point path (tests/norms.py):
    max |got - ref| / (|ref| + median(diag)) <= 1e-5.  The tighter off-diagonal-floor figure the kernels are
    engineered to (floor = median |entry|) must hold as well at these sizes: the larger of the two is returned."""
    from norms import error_figures, tri_diag_scale
    if n is None:                      # packed triangle of n samples: n(n+1)/2 entries
        n = int((np.sqrt(8 * ref.size + 1) - 1) / 2 + 0.5)
    f = error_figures(got, ref, tri_diag_scale(ref, n))
    return max(f["contract"], f["offdiag"])


def _err_figures(got, ref):
    from norms import error_figures, tri_diag_scale
    n = int((np.sqrt(8 * ref.size + 1) - 1) / 2 + 0.5)
    return error_figures(got, ref, tri_diag_scale(ref, n))


@pytest.mark.parametrize("n,L,blk", SIZES)
@pytest.mark.parametrize("missing", [0.0, 0.05])
def test_grm_gcta(n, L, blk, missing, syrk_backend, pair_backend):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=missing, seed=n + 3)
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        got = a.grm_gcta(packed=True)
        nsnp, nloc = a.counts()
        assert nsnp == L
    # tolerance: 1e-5 relative (north_star) with the off-diagonal scale as floor
    assert _rel_err(got, ref) < 1e-5
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)


@pytest.mark.parametrize("panels", [1, 3])
@pytest.mark.parametrize("mode", ["2", "3", "3-nopace"])
def test_converted_operand_kernel_whole_tiles_vs_oracle(mode, panels, monkeypatch):
    """syrk_uv16c_kernel on tiles that are NOT split along K (SNPGPU_I8_TAIL_PARTS=1: what every tile of a large panel is; the sizes
    above only reach the split tiles of a last, partially filled round).  Mode 3 then walks the fp32 runs of a tile inside the work item
    and carries half of each wave's sub-tile sums in LDS between runs -- two runs per block here, blocks of four and of two table
    chunks; as one context, and as three row panels (a panel with its own column offset, panels that end in padding rows)."""
    from snprelate_amd import _lib
    from snprelate_amd.dist import slab_range
    monkeypatch.setenv("SNPGPU_SYRK", "f16")
    monkeypatch.setenv("SNPGPU_SYRK_UV16", mode[0])
    if mode.endswith("nopace"):
        monkeypatch.setenv("SNPGPU_UVC_PACE", "0")
    monkeypatch.setenv("SNPGPU_I8_TAIL_PARTS", "1")
    n, L = 700, 5000
    g = synth_geno(n, L, missing=0.0, seed=977)
    ref = orc.grm_gcta(g)
    got = np.zeros_like(ref)
    bounds = [0, n] if panels == 1 else [0, 256, 512, n]      # (panel rows start at multiples of 256)
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        lo, hi = slab_range(n, r0, r1)
        with _acc(_lib.GRM_GCTA, n, row_begin=r0, row_end=r1, max_block_snps=4096) as a:
            _feed_blocks(a, g, 4096)
            got[lo:hi] = a.grm_gcta(packed=True)
    assert _rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("n,L,blk", SIZES[:3])
@pytest.mark.parametrize("bayesian", [False, True])
def test_pca_cov(n, L, blk, bayesian, syrk_backend):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=0.03, seed=n + 4, special=not bayesian)
    ref = orc.pca_cov(g, bayesian)
    tr_ref = orc.trace_normalize(ref, n)
    with _acc(_lib.PCA_COV, n, bayesian=bayesian, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        got, tr = a.pca_cov(packed=True, normalize=True)
        full, _ = a.pca_cov(packed=False, normalize=True)
    assert abs(tr - tr_ref) / tr_ref < 1e-6
    assert _rel_err(got, ref) < 1e-5
    assert np.array_equal(full, orc.tri_to_full(got, n))


@pytest.mark.parametrize("bayesian", [False, True])
@pytest.mark.parametrize("n,L,blk", [(600, 2500, 1024), (1030, 4100, 4096), (150, 700, 256)])
def test_pca_cov_blocks_without_missing_calls(n, L, blk, bayesian, syrk_backend):
    """PCA covariance (Eigenstrat and Bayesian normalisation) when no block holds a missing call: the default path is the
    single-product kernel with its row / column / constant terms, rare variants (here: every SNP of the 150-sample case,
    a few of the others) through the sparse fp64 kernel; monomorphic SNPs planted."""
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=0.0, seed=n + 40, special=False)
    g[3] = 0; g[5] = 2; g[17] = 1
    g[19] = 0; g[19, :3] = 1                  # three carriers among n
    g[23] = 2; g[23, 5] = 0                   # the same from the other allele
    ref = orc.pca_cov(g, bayesian)
    tr_ref = orc.trace_normalize(ref, n)
    with _acc(_lib.PCA_COV, n, bayesian=bayesian, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        got, tr = a.pca_cov(packed=True, normalize=True)
    assert abs(tr - tr_ref) / tr_ref < 1e-6
    f = _err_figures(got, ref)
    assert f["contract"] < 1e-5 and f["offdiag"] < 1e-5, f


@pytest.mark.parametrize("missing_blocks", ["none", "alternate"])
@pytest.mark.parametrize("diagadj", [True, False])
def test_eigmix_blocks_without_missing_calls(missing_blocks, diagadj, syrk_backend):
    """EIGMIX numerator on blocks without missing calls: weight 1 = 1 x 1, so the single-product kernel (default) is exact up
    to fp32 accumulation; blocks with missing calls in the same stream keep the three-product kernel + the weighted
    both-missing sums."""
    from snprelate_amd import _lib
    n, L, blk = 500, 3000, 500
    g = synth_geno(n, L, missing=0.0, seed=91, special=False)
    g[3] = 0; g[5] = 2; g[13] = 1; g[19] = 0; g[19, :2] = 1
    if missing_blocks == "alternate":
        rng = np.random.default_rng(92)
        for b in range(1, L // blk, 2):
            sub = g[b * blk:(b + 1) * blk]
            sub[rng.random(sub.shape) < 0.04] = 3
    ref, _ = orc.eigmix(g, diagadj)
    with _acc(_lib.EIGMIX, n, max_block_snps=512) as a:
        _feed_blocks(a, g, blk)
        got = a.eigmix(diagadj=diagadj, packed=True)
    assert _rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("n,L,blk", SIZES[:3])
def test_beta_mom_eigmix_synthetic(n, L, blk, pair_backend, syrk_backend):
    from snprelate_amd import _lib
    g = synth_geno(n, L, missing=0.05, seed=n + 9)
    # individual beta counters -> all three finalisers
    cnt = orc.beta_count(g)
    with _acc(_lib.INDIV_BETA, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, blk)
        for mode, ref in ((1, orc.beta_final_ibd(cnt, n, True)), (0, orc.beta_final_ibd(cnt, n, False)),
                          (2, orc.beta_final_grm(cnt, n))):
            got, avg = a.indiv_beta(mode=mode, packed=True)
            np.testing.assert_allclose(got, ref[0], rtol=1e-10, atol=1e-12, equal_nan=True)
            np.testing.assert_allclose(avg, ref[1], rtol=1e-11)
    # PLINK MoM on the IBS context
    e, _ = orc.mom_expect(g)
    for cons in (False, True):
        r0, r1 = orc.mom_final(orc.ibs_count(g), n, e, cons)
        with _acc(_lib.IBS, n, max_block_snps=4096) as a:
            _feed_blocks(a, g, blk)
            k0, k1 = a.ibd_mom(e, constraint=cons, packed=True)
        np.testing.assert_allclose(k0, r0, rtol=1e-12, atol=1e-14, equal_nan=True)
        np.testing.assert_allclose(k1, r1, rtol=1e-12, atol=1e-14, equal_nan=True)
    # EIGMIX
    for diagadj in (True, False):
        ref, _ = orc.eigmix(g, diagadj)
        with _acc(_lib.EIGMIX, n, max_block_snps=4096) as a:
            _feed_blocks(a, g, blk)
            got = a.eigmix(diagadj=diagadj, packed=True)
        assert _rel_err(got, ref) < 1e-5


def test_ragged_blocks_and_forced_tail_split(monkeypatch):
    """Feed blocks of awkward sizes (1 SNP, around the 16/32/64-SNP k-steps, around the 512-SNP table chunk
    and the 4096-SNP fp64 flush) and force the K-split of the scheduler's tail round to an odd part count:
    the int8 pair kernel stays bit-exact, the split-fp16 SYRK within tolerance."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_I8_TAIL_PARTS", "7")
    n = 531
    sizes = [1, 15, 16, 17, 31, 33, 63, 64, 65, 511, 513, 1000, 4095, 4097, 2]
    L = sum(sizes)
    g = synth_geno(n, L, missing=0.07, seed=77)
    g[40] = 3          # an all-missing SNP
    g[41] = 1          # a monomorphic SNP (all het)
    cuts = np.cumsum([0] + sizes)
    ibs_ref, king_ref, grm_ref = orc.ibs_count(g), orc.king_robust_count(g), orc.grm_gcta(g)
    # uint8 genotypes > 3 are missing too (vec_u8_geno_valid, src/dGenGWAS.cpp:1388)
    g_odd = g.copy()
    miss = np.argwhere(g == 3)
    rng = np.random.default_rng(3)
    pick = miss[rng.random(len(miss)) < 0.5]
    g_odd[pick[:, 0], pick[:, 1]] = rng.choice(np.array([4, 7, 8, 9, 64, 128, 254, 255], dtype=np.uint8), size=len(pick))
    with _acc(_lib.IBS, n, max_block_snps=4160) as a:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a.feed(g_odd[lo:hi])
        i0, i1, i2 = a.ibs_num(packed=True)
    assert np.array_equal(i0, ibs_ref[:, 0]) and np.array_equal(i1, ibs_ref[:, 1]) and np.array_equal(i2, ibs_ref[:, 2])
    with _acc(_lib.KING_ROBUST, n, max_block_snps=4160) as a:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a.feed(g[lo:hi])
        assert np.array_equal(a.king_robust_counts(), king_ref)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=4160) as a:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a.feed(g[lo:hi])
        got = a.grm_gcta(packed=True)
    assert _rel_err(got, grm_ref) < 1e-5


@pytest.mark.parametrize("kind", ["GRM_GCTA", "PCA_COV", "PCA_COV_BAYES", "EIGMIX"])
def test_syrk_blocks_with_and_without_missing_calls(kind, monkeypatch):
    """The SYRK variant is chosen per block on the device (missing flag): feed a run in which complete blocks and
    blocks with missing calls alternate, with ragged sizes around the 512-SNP table chunk / 4096-SNP flush, a
    monomorphic and an all-missing SNP, and a forced odd K split of the tail round."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_I8_TAIL_PARTS", "5")
    n = 531
    sizes = [700, 64, 513, 1, 4097, 1000, 511, 2000, 16, 4160]
    with_missing = [False, True, False, False, False, True, False, False, True, False]
    L = sum(sizes)
    g = synth_geno(n, L, missing=0.0, seed=123, special=False)
    cuts = np.cumsum([0] + sizes)
    rng = np.random.default_rng(5)
    for (lo, hi), m in zip(zip(cuts[:-1], cuts[1:]), with_missing):
        if m:
            blk = g[lo:hi]
            blk[rng.random(blk.shape) < 0.06] = 3
    g[5] = 2                      # monomorphic SNP inside a complete block
    g[cuts[1] + 3] = 3            # all-missing SNP inside a block with missing calls
    assert (g[cuts[0]:cuts[1]] <= 2).all() and (g[cuts[1]:cuts[2]] > 2).any()
    bayes = kind.endswith("BAYES")
    K = getattr(_lib, kind.replace("_BAYES", ""))
    with _acc(K, n, max_block_snps=4160, **({"bayesian": True} if bayes else {})) as a:
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a.feed(g[lo:hi])
        if kind == "GRM_GCTA":
            got, ref = a.grm_gcta(packed=True), orc.grm_gcta(g)
        elif kind == "EIGMIX":
            got, ref = a.eigmix(diagadj=True, packed=True), orc.eigmix(g, True)[0]
        else:
            got, ref = a.pca_cov(packed=True, normalize=True)[0], orc.pca_cov(g, bayes)
            orc.trace_normalize(ref, n)
    assert _rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("kind", ["IBS", "KING_ROBUST"])
def test_results_between_feeds(kind):
    """Asking for results between feeds: the rank-one het terms of the binary kernel (blocks without missing calls)
    are folded in once per request, and further feeds keep accumulating."""
    from snprelate_amd import _lib
    n, L = 300, 2400
    g = synth_geno(n, L, missing=0.0, seed=9, special=False)
    g[1500:1800][np.random.default_rng(1).random((300, n)) < 0.05] = 3
    with _acc(getattr(_lib, kind), n, max_block_snps=1024) as a:
        for hi in (600, 1200, 2400):
            _feed_blocks(a, g[(0 if hi == 600 else hi // 2):hi], 300)
            if kind == "IBS":
                ref = orc.ibs_count(g[:hi])
                for _ in range(2):
                    i0, i1, i2 = a.ibs_num(packed=True)
                    assert np.array_equal(i0, ref[:, 0]) and np.array_equal(i1, ref[:, 1]) and np.array_equal(i2, ref[:, 2])
            else:
                ref = orc.king_robust_count(g[:hi])
                for _ in range(2):
                    assert np.array_equal(a.king_robust_counts(), ref)


def _spectrum_geno(n, L, kind, seed):
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.01, 0.5, L) if kind == "array" else np.maximum(0.5 * rng.random(L) ** 3, 2.0 / n)
    p = np.where(rng.random(L) < 0.5, maf, 1 - maf)[:, None]
    return ((rng.random((L, n)) < p).astype(np.uint8) + (rng.random((L, n)) < p).astype(np.uint8))


@pytest.mark.parametrize("kind", ["array", "rare"])
def test_grm_allele_frequency_spectra(kind, syrk_backend):
    """GRM on allele-frequency spectra unlike the bench's U(0.05, 0.95): array-like MAF ~ U(0.01, 0.5) and a
    rare-variant heavy one (MAF = 0.5 u^3): the exact-row SYRK centres its row operand per SNP, so the tolerance holds
    whatever the frequencies (a fixed centre failed this test at 3.5e-4)."""
    from snprelate_amd import _lib
    n, L = 700, 12288
    g = _spectrum_geno(n, L, kind, 17)
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=4096) as a:
        _feed_blocks(a, g, 4096)
        got = a.grm_gcta(packed=True)
    assert np.isfinite(got).all()
    f = _err_figures(got, ref)
    assert f["contract"] < 1e-5, f
    # The off-diagonal-floor figure: 1e-5 for every kernel.  (Round 2's single-product kernel carried each SNP's weight as ONE
    # product of two fp16 numbers, up to 6.6e-6 off, and reached 1.6e-5 here; the SNPs with the largest factorisation error now
    # got a second slot in round 3; since round 4 every fp32 run of a block carries its own weight target.  SNPGPU_SYRK_FAST=1 is
    # round 2's kernel.)
    assert f["offdiag"] < 1e-5, f


def test_grm_rare_variants_take_the_sparse_fp64_path(monkeypatch):
    """Blocks without missing calls: SNPs with at most 128 copies of the minor allele leave the dense fp16 product and are
    added pair by pair in fp64 (uv_sparse_kernel).  With 60 samples that is every SNP, so the GRM must agree with the fp64
    oracle to rounding -- for both allele orientations, on the diagonal, and on a row panel that starts past sample 0."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_SYRK", "f16")
    n, L = 60, 900
    rng = np.random.default_rng(31)
    p = np.where(rng.random(L) < 0.5, rng.uniform(0.02, 0.5, L), rng.uniform(0.5, 0.98, L))[:, None]
    g = ((rng.random((L, n)) < p).astype(np.uint8) + (rng.random((L, n)) < p).astype(np.uint8))
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=512) as a:
        _feed_blocks(a, g, 300)
        got = a.grm_gcta(packed=True)
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-12)
    # the singleton test below, restricted to its rare SNPs, on a panel of rows 256..511 of 700 samples
    n = 700
    g = np.zeros((64, n), np.uint8)
    for k in range(64):
        g[k, rng.choice(n, size=1 + k % 7, replace=False)] = 1 + (k % 5 == 0)
    g[::3] = 2 - g[::3]
    full = orc.tri_to_full(orc.grm_gcta(g), n)
    with _acc(_lib.GRM_GCTA, n, row_begin=256, row_end=512, max_block_snps=64) as a:
        a.feed(g)
        slab = a.grm_gcta(packed=True)
    want = np.concatenate([full[r, r:] for r in range(256, 512)])
    np.testing.assert_allclose(slab, want, rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("kind", ["GRM_GCTA", "PCA_COV", "PCA_COV_BAYES"])
def test_rare_variants_in_blocks_with_missing_calls(kind, monkeypatch):
    """Blocks WITH missing calls keep their rare variants (<= 128 copies of the minor allele) in the exact-row product with
    every called genotype replaced by the non-carrier's; uv_sparse_kernel adds what the carriers' pairs lack -- their products,
    their row / column terms, and those terms back at the cells (carrier, sample with a missing call) -- in fp64.
    (a) 2100 samples (the path needs 384: below that two runs stay bit-identical), singletons ... 4 carriers per SNP, both allele orientations, 3 % missing calls, full triangle and a row
        panel: against the fp64 oracle with and without the sparse path the same tolerance class; the two device results differ
        (the path is taken) and agree to the dense kernel's own accuracy.
    (b) a rare-variant heavy spectrum with missing calls at 18 000 samples: the off-diagonal figure."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_SYRK", "f16")
    bayes = kind.endswith("_BAYES")          # Bayesian allele frequencies (snpgdsPCA(bayesian=TRUE)): weights from (sum + 1) / (2 num + 2)
    kind = kind.replace("_BAYES", "")
    k_id = getattr(_lib, kind)
    rng = np.random.default_rng(41)
    n, L = 2100, 512                                     # weights 1 / (p (1 - p)) from 4200 (singleton) down to 525 (four carriers of 2)
    g = np.zeros((L, n), np.uint8)
    for k in range(L):
        g[k, rng.choice(n, size=1 + k % 4, replace=False)] = 1 + (k % 5 == 0)
    g[::3] = 2 - g[::3]
    g[rng.random((L, n)) < 0.03] = 3

    def run(rows=None):
        kw = dict(row_begin=rows[0], row_end=rows[1]) if rows else {}
        with _acc(k_id, n, max_block_snps=256, bayesian=bayes, **kw) as a:
            _feed_blocks(a, g, 256)
            return a.grm_gcta(packed=True) if kind == "GRM_GCTA" else a.pca_cov(packed=True, normalize=False)[0]
    ref = orc.grm_gcta(g) if kind == "GRM_GCTA" else orc.pca_cov(g, bayesian=bayes)
    got = run()
    monkeypatch.setenv("SNPGPU_X1_SPARSE", "0")
    dense = run()
    monkeypatch.delenv("SNPGPU_X1_SPARSE")
    scale = np.abs(ref).max()
    assert np.abs(got - dense).max() > 0                                 # the sparse path ran ...
    assert np.abs(got - ref).max() < 2e-6 * scale and np.abs(dense - ref).max() < 2e-5 * scale, \
        (np.abs(got - ref).max() / scale, np.abs(dense - ref).max() / scale)
    assert np.abs(got - ref).max() <= np.abs(dense - ref).max()          # ... and does not lose to the dense product
    full = orc.tri_to_full(ref, n)
    slab = run(rows=(512, 1280))
    want = np.concatenate([full[r, r:] for r in range(512, 1280)])
    assert np.abs(slab - want).max() < 2e-6 * scale
    if kind == "PCA_COV":
        return
    # (b)
    n, L = 18000, 512
    g = _spectrum_geno(n, L, "rare", 43)
    g[rng.random((L, n)) < 0.02] = 3
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=512) as a:
        a.feed(g)
        got = a.grm_gcta(packed=True)
    f = _err_figures(got, ref)
    assert f["contract"] < 1e-5 and f["offdiag"] < 1e-5, f


def test_grm_singletons_many_samples():
    """Singleton / doubleton SNPs among 18 000 samples: y^2 = 1 / (p (1 - p)) reaches 36 000 and the column operand
    y^2 (g - avg) would leave fp16's range without the power-of-two balance between row and column operand."""
    from snprelate_amd import _lib
    n, L = 18000, 96
    rng = np.random.default_rng(23)
    g = _spectrum_geno(n, L, "array", 29)
    for k in range(0, 48):                      # half of the SNPs: 1, 2 or 3 carriers only
        g[k] = 0
        g[k, rng.choice(n, size=1 + k % 3, replace=False)] = 1 + (k % 5 == 0)
    for flip in range(0, 48, 4):                # ... some of them counted from the other allele
        g[flip] = 2 - g[flip]
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=64) as a:
        _feed_blocks(a, g, 48)
        got = a.grm_gcta(packed=True)
    assert np.isfinite(got).all()
    f = _err_figures(got, ref)
    # 162 million entries built from 96 SNPs: the maximum of the off-diagonal-floor figure sits 7 sigma out (round 2's
    # single-product kernel with ONE weight target: 1.1e-5; the exact-row kernel: 3e-6; blocks of one table chunk now run as two
    # half-empty chunks with two targets)
    assert f["contract"] < 1e-5 and f["offdiag"] < 1e-5, f


@pytest.mark.parametrize("n,missing,spectrum,special", [(10, 0.1, 0, True), (1001, 0.0, 1, False), (4099, 0.05, 2, True),
                                                        (2050, 0.02, 3, False), (1001, 0.0, 4, False), (515, 0.03, 4, True)])
def test_synth_block_device_matches_numpy_twin(n, missing, spectrum, special):
    """snpgpu_synth_block (the generator of bench.py and of the full-size tests) against oracle/synth.py, bit for bit."""
    import torch
    from oracle.synth import synth_hash_block_packed
    from snprelate_amd import _lib
    lo, m = 990, 1100                      # crosses snp % 997 == 3 / 5 / 7
    buf = torch.empty((m, (n + 3) // 4), dtype=torch.uint8, device="cuda")
    _lib.synth_block(buf.data_ptr(), n, lo, m, 20240601, missing=missing, spectrum=spectrum, special=special)
    assert np.array_equal(buf.cpu().numpy(), synth_hash_block_packed(n, lo, m, 20240601, missing, spectrum, special))


@pytest.mark.parametrize("x1", ["1", "0"])
def test_grm_several_fp32_runs_per_block(x1, monkeypatch):
    """SNPGPU_H3_PROMOTE shorter than the feed block: the one-wave kernel is then launched once per fp32 run (its flush
    sits after the K loop), the two-wave kernel flushes inside its loop -- both must give the same sums as one run."""
    from snprelate_amd import _lib
    n, L = 700, 4096 + 640
    g = synth_geno(n, L, missing=0.02, seed=77)
    ref = orc.grm_gcta(g)
    monkeypatch.setenv("SNPGPU_SYRK_X1", x1)
    outs = []
    for promote in ("1024", "16384"):
        monkeypatch.setenv("SNPGPU_H3_PROMOTE", promote)
        with _acc(_lib.GRM_GCTA, n, max_block_snps=8192) as a:
            a.feed(g)
            outs.append(a.grm_gcta(packed=True))
    assert _rel_err(outs[0], ref) < 1e-5 and _rel_err(outs[1], ref) < 1e-5
    assert np.nanmax(np.abs(outs[0] - outs[1])) < 1e-6


@pytest.mark.parametrize("kind", ["GRM_GCTA", "PCA_COV"])
@pytest.mark.parametrize("L,promote", [(5000, 2048), (3072, 1024), (9000, 1024), (2500, 1024)])
def test_grm_weight_targets_per_fp32_run(kind, L, promote, monkeypatch):
    """Blocks without missing calls that span several fp32 runs: every run carries its own weight target (flush factor
    1 - q / 4096) and the block's SNPs are dealt to the runs (uv_factor / uv_assign / uv_tables kernels).  3, 3, 9 (more than
    UV_QMAX: one target again) and 3 runs with a ragged last block; monomorphic SNPs and rare variants (fp64 sparse path) own
    no slot.  Against the fp64 oracle, and against the same runs with ONE target (SNPGPU_UV_TARGETS=0): the targets must
    not cost accuracy anywhere."""
    from snprelate_amd import _lib
    n = 900
    g = synth_geno(n, L, missing=0.0, seed=L + promote)
    g[5] = 0; g[77] = 2                          # monomorphic
    g[100] = 0; g[100, [3, 500]] = 1             # a rare variant: fp64 path
    ref = orc.grm_gcta(g) if kind == "GRM_GCTA" else orc.pca_cov(g, False)
    monkeypatch.setenv("SNPGPU_H3_PROMOTE", str(promote))
    outs = {}
    for targets in ("1", "0"):
        monkeypatch.setenv("SNPGPU_UV_TARGETS", targets)
        with _acc(getattr(_lib, kind), n, max_block_snps=16384) as a:
            a.feed(g)
            outs[targets] = a.grm_gcta(packed=True) if kind == "GRM_GCTA" else a.pca_cov(packed=True, normalize=False)[0]
    f1, f0 = _err_figures(outs["1"], ref), _err_figures(outs["0"], ref)
    assert f1["offdiag"] < 1e-5 and f0["offdiag"] < 1e-5, (f1, f0)
    assert f1["offdiag"] < 2.0 * f0["offdiag"] + 1e-6, (f1, f0)


@pytest.mark.parametrize("missing", [0.004, 0.02, 0.06])
@pytest.mark.parametrize("sparse", ["1", "0"])
def test_grm_gcta_denominators_sparse_and_dense_routes(missing, sparse, monkeypatch):
    """GCTA's both-missing counts (src/genPCA.cpp:1201-1224): blocks with few missing calls (here: up to 3 %) count them from
    per-SNP SETS of samples (missmask256_kernel + pair_sparse_miss_kernel), blocks above that with the dense int8 product -- the
    route is picked per block on the device.  1300 samples (five full groups of 256 + a ragged one), ragged blocks, a sample without a
    single call, monomorphic and all-missing SNPs (which GCTA does not count), a row panel that starts past sample 0; a
    denominator off by one would move an entry by 5e-4."""
    from snprelate_amd import _lib
    monkeypatch.setenv("SNPGPU_GCTA_SPARSE", sparse)
    monkeypatch.setenv("SNPGPU_GCTA_SPARSE_MAX_RATE", "0.03")      # (default 0.003: where the sparse form is the faster one)
    n, L = 1300, 2300
    g = synth_geno(n, L, missing=missing, seed=int(missing * 1000) + 3)
    g[:, 77] = 3                                  # a sample that is missing everywhere
    g[1200:1500][np.random.default_rng(4).random((300, n)) < 0.08] = 3       # one block above the threshold among sparse ones
    ref = orc.grm_gcta(g)
    with _acc(_lib.GRM_GCTA, n, max_block_snps=1024) as a:
        for i in range(0, L, 700):
            a.feed(g[i:i + 700])
        got = a.grm_gcta(packed=True)
    fin = np.isfinite(ref)                        # (pairs with the all-missing sample: 0 / 0)
    assert np.array_equal(np.isfinite(got), fin)
    assert _rel_err(np.where(fin, got, 0.0), np.where(fin, ref, 0.0), n) < 1e-5
    from snprelate_amd.dist import slab_range
    with _acc(_lib.GRM_GCTA, n, max_block_snps=1024, row_begin=512, row_end=1024) as a:
        for i in range(0, L, 1024):
            a.feed(g[i:i + 1024])
        part = a.grm_gcta(packed=True)
    lo, hi = slab_range(n, 512, 1024)
    f2 = np.isfinite(ref[lo:hi])
    assert np.array_equal(np.isfinite(part), f2) and np.nanmax(np.abs(part[f2] - ref[lo:hi][f2])) < 1e-5 * np.nanmax(np.abs(ref[fin]))


@pytest.mark.parametrize("n", [1008, 1030, 2048])
@pytest.mark.parametrize("missing", [0.0, 0.03])
def test_counters_from_2bit_rows_one_pass_prepass(n, missing, monkeypatch):
    """IBS / KING-robust fed with GDS-style 2-bit rows: the one-pass pre-pass (transpose2_direct_kernel: n % 16 == 0; a
    partial last 64-sample chunk at n = 1008) and the two-kernel form (n = 1030, or SNPGPU_PREP_TWO_PASS=1) must give the
    oracle's counters bit for bit, for blocks with and without missing calls and a ragged last block."""
    from snprelate_amd import _lib
    from snprelate_amd.gds import pack_2bit_rows
    L = 2100
    g = synth_geno(n, L, missing=missing, seed=n + 5, special=missing > 0)
    if missing == 0.0:
        g[1500, 7] = 3                     # one block with a missing call among blocks without
    packed = pack_2bit_rows(g)
    ref_i, ref_k = orc.ibs_count(g), orc.king_robust_count(g)
    for two_pass in (False, True):
        if two_pass:
            monkeypatch.setenv("SNPGPU_PREP_TWO_PASS", "1")
        with _acc(_lib.IBS, n, max_block_snps=1024) as a:
            for i in range(0, L, 1000):
                a.feed(packed[i:i + 1000], fmt=_lib.GENO_PACKED2)
            i0, i1, i2 = a.ibs_num(packed=True)
        assert np.array_equal(np.stack([i0, i1, i2], 1).astype(np.uint32), ref_i)
        with _acc(_lib.KING_ROBUST, n, max_block_snps=1024) as a:
            for i in range(0, L, 1000):
                a.feed(packed[i:i + 1000], fmt=_lib.GENO_PACKED2)
            assert np.array_equal(a.king_robust_counts(), ref_k)


def test_counters_direct_and_two_pass_prepass_alternate_within_one_context():
    """The one-pass pre-pass of the IBS / KING-robust counters reads the caller's 2-bit rows with dword loads, so a feed whose
    DEVICE pointer is not 4-byte aligned takes the two-kernel form instead: blocks of one context may alternate between the two
    (and between blocks with and without missing calls) and the counters must not notice."""
    import torch
    from snprelate_amd import _lib
    from snprelate_amd.gds import pack_2bit_rows
    n, L, blk = 1008, 3000, 700
    g = synth_geno(n, L, missing=0.0, seed=91, special=False)
    g[1500:2200][np.random.default_rng(2).random((700, n)) < 0.03] = 3
    packed = torch.from_numpy(pack_2bit_rows(g)).cuda()
    rb = packed.shape[1]
    raw = torch.empty(blk * rb + 8, dtype=torch.uint8, device="cuda")
    ref_i, ref_k = orc.ibs_count(g), orc.king_robust_count(g)
    with _acc(_lib.IBS, n, max_block_snps=1024) as a, _acc(_lib.KING_ROBUST, n, max_block_snps=1024) as k:
        for t, lo in enumerate(range(0, L, blk)):
            m = min(blk, L - lo)
            off = (t % 2) * 1                                   # every second block at an odd address
            view = raw[off: off + m * rb]
            view.copy_(packed[lo:lo + m].reshape(-1))
            assert (view.data_ptr() & 3) == (1 if off else 0)
            a.feed_device(view.data_ptr(), m)
            k.feed_device(view.data_ptr(), m)
            a.sync(); k.sync()                                  # the staging buffer is rewritten by the next block
        i0, i1, i2 = a.ibs_num(packed=True)
        assert np.array_equal(np.stack([i0, i1, i2], 1).astype(np.uint32), ref_i)
        assert np.array_equal(k.king_robust_counts(), ref_k)
