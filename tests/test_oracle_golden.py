"""Pin the CPU oracle against the reference's own golden vectors
(inst/unitTests/test_rel.R; fixtures made by tests/golden/make_golden.py) and
the known answers recorded in SURVEY.md 8(c)."""
import os

import numpy as np

import oracle as orc
from conftest import GOLDEN, synth_geno


def _autosome(f):
    return (f.snp_chromosome >= 1) & (f.snp_chromosome <= 22)


def _subset(f, n_samp, missing_rate=float("nan")):
    auto = _autosome(f)
    g = f.read_genotype(snp_sel=auto, samp_sel=np.arange(n_samp))
    sel = orc.select_snp_base(g, True, float("nan"), missing_rate)
    return np.ascontiguousarray(g[sel]), f.snp_id[auto][sel]


def test_gds_fixture_reads(hapmap):
    assert (hapmap.n_snp, hapmap.n_samp) == (9088, 279)
    g = hapmap.read_genotype()
    # value counts from SURVEY.md Appendix A
    assert np.bincount(g.ravel()).tolist() == [920248, 657267, 947899, 10138]
    assert int((hapmap.snp_chromosome == 23).sum()) == 365


def test_ibs_golden(hapmap):
    z = np.load(os.path.join(GOLDEN, "validate_ibs.npz"))
    g, ids = _subset(hapmap, 90)
    assert np.array_equal(ids, z["snp_id"])           # .InitFile2 filter restated
    cnt = orc.ibs_count(g)
    assert cnt[1].tolist() == [447, 3160, 5050]       # pair (1,2), SURVEY 8(c)
    assert cnt[0].tolist() == [0, 0, 8668]
    ibs = orc.tri_to_full(orc.ibs_ave(cnt, 90), 90)
    assert np.array_equal(ibs, z["ibs"])              # bit-for-bit


def test_king_golden(hapmap):
    z = np.load(os.path.join(GOLDEN, "validate_king.npz"))
    g, ids = _subset(hapmap, 60)
    assert np.array_equal(ids, z["snp_id"])
    c = orc.king_robust_count(g)
    assert c[1].tolist() == [447, 8622, 4948, 2579, 2447]
    i0, kin = orc.king_robust_final(c, 60)
    assert np.array_equal(orc.tri_to_full(i0, 60), z["robust_IBS0"])
    assert np.array_equal(orc.tri_to_full(kin, 60), z["robust_kinship"])
    c, fs = orc.king_homo_count(g)
    k0, k1 = orc.king_homo_final(c, fs, 60)
    np.testing.assert_allclose(orc.tri_to_full(k0, 60), z["homo_k0"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(orc.tri_to_full(k1, 60), z["homo_k1"], rtol=1e-12, atol=1e-13)


def test_pca_genmat_golden(hapmap):
    z = np.load(os.path.join(GOLDEN, "validate_pca.npz"))
    g, _ = _subset(hapmap, 90)
    cov = orc.pca_cov(g)
    orc.trace_normalize(cov, 90)
    np.testing.assert_allclose(orc.tri_to_full(cov, 90), z["genmat"], rtol=1e-12, atol=1e-13)


def _principal_angles(a, b):
    """principal angles between the column spaces of a and b (radians, ascending)"""
    qa, qb = np.linalg.qr(a)[0], np.linalg.qr(b)[0]
    return np.arccos(np.clip(np.linalg.svd(qa.T @ qb, compute_uv=False), -1.0, 1.0))


def test_pca_randomized_pinned_by_the_exact_pca_golden(hapmap):
    """snpgdsPCA(algorithm = "randomized") has no golden of its own in the reference's tests.  What the reference does fix is the
    exact PCA of the same call -- Validate.PCA.RData$genmat (test_rel.R:128-142), HapMap's first 90 samples -- and a randomised
    PCA with iter.num = 10 and aux.dim = 2 * eigen.cnt (R/PCA.R:54-62, 80-89) must reproduce that matrix's leading eigenpairs:
    the restatement of CRandomPCA::Run (src/genPCA.cpp:672-792) does, top-4 subspace to < 1e-3 rad (measured 6e-6) and
    eigenvalues to 1e-8 relative (measured 5e-11; the algorithm's own error after 10 iterations at these spectral gaps)."""
    z = np.load(os.path.join(GOLDEN, "validate_pca.npz"))
    n, k = 90, 4
    g, _ = _subset(hapmap, n)
    w, v = np.linalg.eigh(z["genmat"])
    w, v = w[::-1][:k], v[:, ::-1][:, :k]
    aux = np.random.default_rng(2024).normal(size=(2 * k, n))
    sig, vt, tr2 = orc.pca_randomized(g, aux, 10)
    val = (n - 1) * 2 * sig[:k] ** 2 / tr2                      # R/PCA.R:86: eigenval = sigma^2 * (n - 1) / TraceXTX
    np.testing.assert_allclose(val, w, rtol=1e-8)
    assert _principal_angles(vt[:k].T, v).max() < 1e-3
    # four iterations (the accuracy actually depends on iter.num): the two structural components still agree, the rest do not yet
    sig4, vt4, _ = orc.pca_randomized(g, aux, 4)
    np.testing.assert_allclose((n - 1) * 2 * sig4[:2] ** 2 / tr2, w[:2], rtol=1e-8)
    assert _principal_angles(vt4[:2].T, v[:, :2]).max() < 1e-4 and _principal_angles(vt4[:k].T, v).max() > 1e-3


def _sign_fix(got, gold, axis):
    """Eigenvectors are defined up to sign: align every component with the golden one."""
    s = np.sign(np.nansum(got * gold, axis=axis, keepdims=True))
    s[s == 0] = 1
    return got * s


def test_pca_projections_golden(hapmap):
    """snpgdsPCACorr / snpgdsPCASNPLoading / snpgdsPCASampLoading goldens of Validate.PCA.RData
    (inst/unitTests/test_rel.R:20-44, 144-160): rounded to 3 / 3 / 4 decimals by the reference."""
    z = np.load(os.path.join(GOLDEN, "validate_pca.npz"))
    n = 90
    g, ids = _subset(hapmap, n)                         # snpgdsPCA(sample.id=first 90, missing.rate=NaN)
    cov = orc.pca_cov(g)
    trace = orc.trace_normalize(cov, n)
    w, v = np.linalg.eigh(orc.tri_to_full(cov, n))
    w, v = w[::-1][:8], v[:, ::-1][:, :8].T.copy()      # eigenval [8], eigenvect [8][n]
    # SNP loadings over the PCA's own SNP set
    load, avg, scale = orc.pca_snp_loading(g, w, v, trace)
    got = _sign_fix(load.T, z["snploading"], axis=1)
    assert np.abs(got - z["snploading"]).max() < 5.0001e-4
    # SNP correlation with eigenvectors 1:2 over ALL 9088 SNPs (snp.id = NULL -> .InitFile, no filter)
    g_all = hapmap.read_genotype(samp_sel=np.arange(n))
    corr = orc.pca_snp_corr(g_all, v[:2]).T
    assert np.array_equal(np.isnan(corr), np.isnan(z["corr"]))
    got = _sign_fix(corr, z["corr"], axis=1)
    assert np.nanmax(np.abs(got - z["corr"])) < 5.0001e-4
    # sample loadings of the first 100 samples from the SNP loadings
    sload = load * np.sqrt((n - 1) / trace / w)[None, :]
    sel = np.isin(hapmap.snp_id, ids)
    g100 = hapmap.read_genotype(snp_sel=sel, samp_sel=np.arange(100))
    sl = orc.pca_samp_loading(g100, sload, avg, scale)  # [8][100]
    got = _sign_fix(sl.T, z["samploading"], axis=0)
    assert np.abs(got - z["samploading"]).max() < 5.0001e-5
    # the first 90 projected samples are the PCA's own samples: projection reproduces their eigenvectors
    np.testing.assert_allclose(np.abs(sl[:, :n]), np.abs(v), atol=1e-10)


def test_gcta_known_answers(hapmap):
    # snpgdsGRM(f) defaults: autosome, remove.monosnp, missing.rate=0.01 -> 279 x 8039
    auto = _autosome(hapmap)
    g = hapmap.read_genotype(snp_sel=auto)
    sel = orc.select_snp_base(g, True, float("nan"), 0.01)
    assert int(sel.sum()) == 8039
    g = np.ascontiguousarray(g[sel])
    assert int((g > 2).sum()) == 1583
    grm = orc.tri_to_full(orc.grm_gcta(g), 279)
    # values recorded in SURVEY.md 8(c) from the reference's own CGCTA_AlgArith
    np.testing.assert_allclose(grm[0, 0], 1.3925009439720786, rtol=1e-13)
    np.testing.assert_allclose(grm[1, 1], 1.3087163456244955, rtol=1e-13)
    np.testing.assert_allclose(grm[0, 1], 0.2421021559446208, rtol=1e-13)
    np.testing.assert_allclose(grm[0, 278], -0.09733479032489507, rtol=1e-12)
    np.testing.assert_allclose(np.trace(grm), 305.28199493848285, rtol=1e-13)


def test_grm_merge_self_consistency(hapmap):
    """inst/unitTests/test_GRM.R:14-87 -- on SNPs without missing calls the merge (gnrGRMMerge restatement) of the
    GRMs of a SNP partition equals the all-SNP GRM, for GCTA and for IndivBeta (which the merge has to
    back-transform and re-baseline)."""
    g = hapmap.read_genotype(snp_sel=_autosome(hapmap))
    g = g[(g > 2).sum(axis=1) == 0]
    g = np.ascontiguousarray(g[orc.select_snp_base(g, True)])
    n = g.shape[1]
    parts = [np.ascontiguousarray(p) for p in (g[:1000], g[1000:3000], g[3000:])]
    w = np.array([len(p) for p in parts], float) / len(g)
    merged, _ = orc.grm_merge([orc.tri_to_full(orc.grm_gcta(p), n) for p in parts], w)
    np.testing.assert_allclose(merged, orc.tri_to_full(orc.grm_gcta(g), n), rtol=1e-10, atol=1e-12)

    sub = [orc.beta_final_grm(orc.beta_count(p), n) for p in parts]
    merged, avg = orc.grm_merge([orc.tri_to_full(t, n) for t, _ in sub], w, ":method = IndivBeta", [a for _, a in sub])
    whole, wavg = orc.beta_final_grm(orc.beta_count(g), n)
    np.testing.assert_allclose(merged, orc.tri_to_full(whole, n), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(avg, wavg, rtol=1e-10)


def test_oracle_vs_definitions_on_synthetic():
    """Independent numpy restatement straight from the definitions."""
    g = synth_geno(37, 301, missing=0.05, seed=3)
    n = g.shape[1]
    gi = g.astype(np.int64)
    valid = g <= 2
    iu = np.triu_indices(n)
    both = (valid[:, :, None] & valid[:, None, :])
    d = np.abs(gi[:, :, None] - gi[:, None, :])
    ibs = np.stack([((d == k) & both).sum(0)[iu] for k in (2, 1, 0)], axis=1)
    assert np.array_equal(orc.ibs_count(g), ibs.astype(np.uint32))
    het = (gi == 1) & valid
    king = np.stack([((d == 2) & both).sum(0)[iu], both.sum(0)[iu], ((d * d) * both).sum(0)[iu],
                     (het[:, :, None] & both).sum(0)[iu], (het[:, None, :] & both).sum(0)[iu]], axis=1)
    assert np.array_equal(orc.king_robust_count(g), king.astype(np.uint32))
    # GCTA from the formula in man/snpgdsGRM.Rd:50-70 with per-pair missing handling
    s = (gi * valid).sum(1).astype(float)
    c = valid.sum(1).astype(float)
    with np.errstate(all="ignore"):
        avg = np.where(c > 0, s / np.maximum(c, 1), 0)
    p = avg / 2
    poly = (s > 0) & (s < 2 * c)
    scale = np.where((p > 0) & (p < 1), 1 / np.sqrt(np.where((p > 0) & (p < 1), p * (1 - p), 1)), 0)
    Z = np.where(valid, (gi - avg[:, None]) * scale[:, None], 0.0)
    num = Z.T @ Z
    den = 2 * (both & poly[:, None, None]).sum(0)
    with np.errstate(all="ignore"):
        ref = (num / den)[iu]
    got = orc.grm_gcta(g)
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(orc.pca_cov(g), num[iu], rtol=1e-11, atol=1e-11)


def test_mom_beta_eigmix_goldens(hapmap):
    """test.PLINK.MoM :193-224, test.IndivBeta :277-304, test.EIGMIX :308-327 of test_rel.R"""
    g, ids = _subset(hapmap, 90)
    z = np.load(os.path.join(GOLDEN, "validate_mom.npz"))
    assert np.array_equal(ids, z["snp_id"])
    e, af = orc.mom_expect(g)
    k0, k1 = orc.mom_final(orc.ibs_count(g), 90, e)
    np.testing.assert_allclose(orc.tri_to_full(k0, 90), z["k0"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(orc.tri_to_full(k1, 90), z["k1"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(af, z["afreq"], rtol=1e-14)
    z = np.load(os.path.join(GOLDEN, "validate_beta.npz"))
    b, _ = orc.beta_final_ibd(orc.beta_count(g), 90, True)
    np.testing.assert_allclose(orc.tri_to_full(b, 90), z["beta"], rtol=1e-12, atol=1e-14)
    z = np.load(os.path.join(GOLDEN, "validate_eigmix.npz"))
    ibd, _ = orc.eigmix(g, True)
    np.testing.assert_allclose(orc.tri_to_full(ibd, 90), z["ibd"], rtol=1e-12, atol=1e-14)


def test_beta_counts_from_ibs_identity():
    """ibscnt = IBS1 + 2*IBS2 - #(both het): ties the beta counters to the IBS counters."""
    g = synth_geno(41, 500, missing=0.06, seed=8)
    ibs = orc.ibs_count(g).astype(np.int64)
    beta = orc.beta_count(g).astype(np.int64)
    het = (g == 1)
    iu = np.triu_indices(41)
    hh = (het[:, :, None] & het[:, None, :]).sum(0)[iu]
    assert np.array_equal(beta[:, 1], ibs.sum(1))
    assert np.array_equal(beta[:, 0], ibs[:, 1] + 2 * ibs[:, 2] - hh)
