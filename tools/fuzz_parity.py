#!/usr/bin/env python3
"""Randomised parity sweep (GPU): random sample counts, SNP counts, feed-block sizes, missing rates and panel
splits for IBS / KING-robust counters (bit-exact), the GCTA GRM (1e-5), and on single-panel cases KING-homo (1e-5), the PCA covariance (plain / Bayesian, 1e-5), the EIGMIX matrix (1e-5) and the
individual-beta estimates (1e-10; integer counters underneath) against the CPU oracle.
    tools/fuzz_parity.py [n_cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle as orc  # noqa: E402
from oracle.synth import synth_geno  # noqa: E402
from snprelate_amd import _lib  # noqa: E402
from snprelate_amd.dist import panel_rows, slab_range  # noqa: E402
from snprelate_amd.gds import pack_2bit_rows  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.integers(2, 1400))
    L = int(rng.integers(1, 6000))
    blk = int(rng.choice([64, 100, 512, 1000, 4096]))
    miss = float(rng.choice([0.0, 0.0, 0.01, 0.1, 0.5]))
    world = int(rng.choice([1, 1, 2, 3]))
    packed2 = bool(rng.random() < 0.5)
    if packed2 and rng.random() < 0.5:
        n = max(16, n // 16 * 16)
    big_rare = rng.random() < 0.25          # panels of >= 384 samples with rare variants: with missing calls their carriers'
    if big_rare:                           # pairs are added in fp64 beside the exact-row product (uv_sparse_kernel, missing_blocks)
        n = int(rng.integers(384, 3400))
        L = int(rng.integers(64, 1500))
    g = synth_geno(n, L, missing=miss, seed=int(rng.integers(1 << 30)))
    if big_rare:
        for k in rng.choice(L, size=L // 2, replace=False):      # half of the SNPs: 1 .. 12 carriers, either allele, calls kept missing
            m3 = g[k] == 3
            g[k] = 0
            g[k, rng.choice(n, size=int(rng.integers(1, 13)), replace=False)] = int(rng.integers(1, 3))
            if rng.random() < 0.5:
                g[k] = 2 - g[k]
            g[k, m3] = 3
    if L > 10 and rng.random() < 0.5:
        g[rng.integers(0, L)] = 3
        g[rng.integers(0, L)] = int(rng.integers(0, 3))
    ibs_ref, king_ref, grm_ref = orc.ibs_count(g), orc.king_robust_count(g), orc.grm_gcta(g)
    b = panel_rows(n, world)
    ibs = np.zeros_like(ibs_ref); king = np.zeros_like(king_ref); grm = np.zeros_like(grm_ref)
    for r in range(world):
        if b[r + 1] <= b[r]:
            continue
        lo, hi = slab_range(n, b[r], b[r + 1])
        kw = dict(max_block_snps=max(blk, 64))
        if world > 1:
            kw.update(row_begin=b[r], row_end=b[r + 1])
        for kind in (_lib.IBS, _lib.KING_ROBUST, _lib.GRM_GCTA):
            with _lib.Accumulator(kind, n, **kw) as a:
                for i in range(0, L, blk):
                    if packed2:      # GDS-style 2-bit rows (the one-pass pre-pass of the counters when n % 16 == 0)
                        a.feed(pack_2bit_rows(g[i:i + blk]), fmt=_lib.GENO_PACKED2)
                    else:
                        a.feed(g[i:i + blk])
                if kind == _lib.IBS:
                    i0, i1, i2 = a.ibs_num(packed=True)
                    ibs[lo:hi] = np.stack([i0, i1, i2], 1)
                elif kind == _lib.KING_ROBUST:
                    king[lo:hi] = a.king_robust_counts()
                else:
                    grm[lo:hi] = a.grm_gcta(packed=True)
    ok_x = True
    if world == 1 and n >= 3:
        kw = dict(max_block_snps=max(blk, 64))
        with _lib.Accumulator(_lib.KING_HOMO, n, **kw) as a:
            for i in range(0, L, blk):
                a.feed(pack_2bit_rows(g[i:i + blk]), fmt=_lib.GENO_PACKED2) if packed2 else a.feed(g[i:i + blk])
            k0, k1 = a.king_homo(packed=True)
        c, fs = orc.king_homo_count(g)
        r0, r1 = orc.king_homo_final(c, fs, n)
        ok_x &= bool(np.allclose(k0, r0, rtol=1e-5, atol=1e-7, equal_nan=True) and np.allclose(k1, r1, rtol=1e-5, atol=2e-5, equal_nan=True))
        # PCA covariance numerator (plain and Bayesian allele frequencies) and the EIGMIX matrix, same norm as the GRM
        for bayes in (False, True):
            with _lib.Accumulator(_lib.PCA_COV, n, bayesian=bayes, **kw) as a:
                for i in range(0, L, blk):
                    a.feed(pack_2bit_rows(g[i:i + blk]), fmt=_lib.GENO_PACKED2) if packed2 else a.feed(g[i:i + blk])
                got = a.pca_cov(packed=True, normalize=False)[0]
            ref = orc.pca_cov(g, bayesian=bayes)
            sc = np.median(np.abs(ref))
            e = float(np.max(np.abs(got - ref) / (np.abs(ref) + sc))) if sc > 0 else float(np.max(np.abs(got - ref)))
            if not e < 1e-5:
                print("   PCA covariance (bayesian=%s): %.2e" % (bayes, e))
                ok_x = False
        with _lib.Accumulator(_lib.EIGMIX, n, **kw) as a:
            for i in range(0, L, blk):
                a.feed(pack_2bit_rows(g[i:i + blk]), fmt=_lib.GENO_PACKED2) if packed2 else a.feed(g[i:i + blk])
            got = a.eigmix(diagadj=True, packed=True)
        ref = orc.eigmix(g, diagadj=True)
        ref = ref[0] if isinstance(ref, tuple) else ref
        fin_e = np.isfinite(ref)
        if fin_e.any():
            sc = np.median(np.abs(ref[fin_e]))
            e = float(np.max(np.abs(got[fin_e] - ref[fin_e]) / (np.abs(ref[fin_e]) + sc))) if sc > 0 else 0.0
            if not (e < 1e-5 and np.array_equal(np.isfinite(got), fin_e)):
                print("   EIGMIX: %.2e" % e)
                ok_x = False
        with _lib.Accumulator(_lib.INDIV_BETA, n, **kw) as a:
            for i in range(0, L, blk):
                a.feed(pack_2bit_rows(g[i:i + blk]), fmt=_lib.GENO_PACKED2) if packed2 else a.feed(g[i:i + blk])
            got, avg = a.indiv_beta(mode=2, packed=True)
        ref = orc.beta_final_grm(orc.beta_count(g), n)
        ok_x &= bool(np.allclose(got, ref[0], rtol=1e-10, atol=1e-12, equal_nan=True))
    ok_i = np.array_equal(ibs, ibs_ref)
    ok_k = np.array_equal(king, king_ref)
    fin = np.isfinite(grm_ref)
    err = 0.0
    if fin.any():
        scale = np.median(np.abs(grm_ref[fin]))
        err = float(np.nanmax(np.abs(grm[fin] - grm_ref[fin]) / (np.abs(grm_ref[fin]) + scale))) if scale > 0 else 0.0
    ok_g = err < 1e-5 and np.array_equal(np.isfinite(grm), fin)
    print("case %2d n=%4d L=%4d blk=%4d miss=%.2f panels=%d %s  IBS %s KING %s GRM %s (%.1e) HOMO+BETA+PCA+EIGMIX %s" %
          (case, n, L, blk, miss, world, "2bit" if packed2 else "u8  ", ok_i, ok_k, ok_g, err, ok_x), flush=True)
    bad += not (ok_i and ok_k and ok_g and ok_x)
print("FAILED cases: %d" % bad)
sys.exit(1 if bad else 0)
