import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import oracle as orc
from oracle.synth import synth_geno
def rel_err(got, ref):
    floor = np.median(np.abs(ref))
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), floor)))
for n, L, blk, miss in [(600, 2500, 1024, 0.05), (1030, 8200, 4096, 0.0), (1030, 65536, 16384, 0.02)]:
    g = synth_geno(n, L, missing=miss, seed=n)
    ref = orc.grm_gcta(g)
    for be in ("f16", "h3", "f32"):
        os.environ["SNPGPU_SYRK"] = be
        from snprelate_amd import _lib
        with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=blk) as a:
            for i in range(0, L, blk): a.feed(g[i:i + blk])
            got = a.grm_gcta(packed=True)
        print(n, L, miss, be, "max rel err %.3e" % rel_err(got, ref))
# longer accumulation: one 16384-SNP block per feed, more SNPs
for n, L, blk, miss in [(2050, 131072, 16384, 0.01), (2050, 131072, 16384, 0.0)]:
    g = synth_geno(n, L, missing=miss, seed=n)
    ref = orc.grm_gcta(g)
    for be in ("f16", "h3", "f32"):
        os.environ["SNPGPU_SYRK"] = be
        with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=blk) as a:
            for i in range(0, L, blk): a.feed(g[i:i + blk])
            got = a.grm_gcta(packed=True)
        d = np.abs(got - ref)
        print(n, L, miss, be, "max rel err %.3e" % rel_err(got, ref), "diag max rel %.3e" % float(np.max(d[[i * n - i * (i - 1) // 2 for i in range(n)]] / np.abs(ref[[i * n - i * (i - 1) // 2 for i in range(n)]]))))
