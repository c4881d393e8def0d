"""Seeded synthetic genotype generator shared by tests, smoke() and bench.py's cpu_baseline
(test infrastructure)."""
import numpy as np


def synth_geno(n_samp, n_snp, missing=0.02, seed=1, special=True):
    """Seeded synthetic genotypes uint8 [n_snp][n_samp]: per-SNP p ~ U(0.05,0.95),
    Binomial(2,p), iid missing; with `special`, a few monomorphic, all-missing
    and single-sample-valid SNPs are planted (edge cases of SURVEY.md 8d)."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.05, 0.95, size=(n_snp, 1))
    g = (rng.random((n_snp, n_samp)) < p).astype(np.uint8) + \
        (rng.random((n_snp, n_samp)) < p).astype(np.uint8)
    if missing > 0:
        g[rng.random((n_snp, n_samp)) < missing] = 3
    if special and n_snp >= 16:
        g[3] = 0                     # monomorphic (all AA... 0 copies)
        g[5] = 2                     # monomorphic
        g[7] = 3                     # all missing
        g[11] = 3
        g[11, 0] = 1                 # a single valid call
        g[13, :] = 1                 # all heterozygous (p = 0.5, polymorphic)
    return np.ascontiguousarray(g)
