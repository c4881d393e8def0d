#!/usr/bin/env python3
"""Randomised sweep of the top-k eigen solver (GPU): random sample counts (not multiples of anything), SNP counts, missing rates,
row-panel splits, matrix kinds (PCA covariance, GCTA GRM / EIGMIX matrix finalised in place) and k; the block-Krylov solver
behind snpgpu_panels_topk_eigen against numpy's eigh (LAPACK) of the device's OWN gathered matrix: eigenvalues, residuals.
    tools/fuzz_eigen.py [n_cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import oracle as orc  # noqa: E402
from oracle.synth import synth_geno  # noqa: E402
from snprelate_amd import _lib  # noqa: E402
from snprelate_amd.dist import panel_rows  # noqa: E402
from snprelate_amd.eigen import PanelOperator, topk_eigen  # noqa: E402

import torch  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
bad = 0
for case in range(cases):
    n = int(rng.integers(40, 2600))
    L = int(rng.integers(200, 2500))
    miss = float(rng.choice([0.0, 0.02, 0.2]))
    kind = str(rng.choice(["PCA_COV", "GRM_GCTA", "EIGMIX"]))
    world = int(rng.choice([1, 1, 2, 3]))
    k = int(rng.integers(1, max(2, min(33, n // 3))))
    blk = int(rng.choice([256, 1000, 4096]))
    g = synth_geno(n, L, missing=miss, seed=int(rng.integers(1 << 30)))
    # structure: shift the allele frequencies of a third of the samples on half of the SNPs
    grp = rng.random(n) < 0.33
    flip = rng.random(L) < 0.5
    sub = g[np.ix_(flip, grp)]
    sub[(sub == 0) & (rng.random(sub.shape) < 0.4)] = 1
    g[np.ix_(flip, grp)] = sub
    b = panel_rows(n, world)
    panels = []
    for r in range(world):
        if b[r + 1] > b[r]:
            a = _lib.Accumulator(getattr(_lib, kind), n, row_begin=b[r], row_end=b[r + 1] if world > 1 else 0, max_block_snps=max(blk, 64))
            for i in range(0, L, blk):
                a.feed(g[i:i + blk])
            panels.append(a)
    # the device's own matrix, gathered
    tri = np.zeros(n * (n + 1) // 2)
    from snprelate_amd.dist import slab_range
    scale = 1.0
    for a, r in zip(panels, [r for r in range(world) if b[r + 1] > b[r]]):
        lo, hi = slab_range(n, b[r], b[r + 1])
        if kind == "PCA_COV":
            tri[lo:hi] = a.pca_cov(packed=True, normalize=False)[0]
        elif kind == "GRM_GCTA":
            tri[lo:hi] = a.grm_gcta(packed=True)
        else:
            tri[lo:hi] = a.eigmix(packed=True)
    full = orc.tri_to_full(tri, n)
    if kind != "PCA_COV":
        for a in panels:
            a.finalize_inplace()
    else:
        scale = (n - 1) / np.trace(full)
        full = full * scale
    ok = np.isfinite(full).all()
    err = res = float("nan")
    if ok:
        op = PanelOperator(panels, n, dev, normalize=(kind == "PCA_COV"))
        w, v, info = topk_eigen(op, k)
        w, v = w.cpu().numpy(), v.cpu().numpy()
        wr = np.linalg.eigvalsh(full)[::-1][:k]
        err = float(np.max(np.abs(w - wr) / np.abs(wr[0])))
        res = float(np.max(np.linalg.norm(full @ v - v * w, axis=0) / np.abs(w)))
        ok = err < 1e-9 and res < 1e-7
    for a in panels:
        a.close()
    print("case %2d %-8s n=%4d L=%4d k=%2d miss=%.2f panels=%d  eigenvalues %.1e residual %.1e fp32 products %d of %d  %s" %
          (case, kind, n, L, k, miss, world, err, res, info["matmuls_fp32"] if ok or err == err else -1, info["matmuls"] if ok or err == err else -1,
           "ok" if ok else "FAILED"), flush=True)
    bad += not ok
print("FAILED cases: %d" % bad)
