"""Seeded synthetic genotype generator shared by tests, smoke() and bench.py's cpu_baseline
(test infrastructure)."""
import math

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


# ---------------------------------------------------------------------------
# Counter-based generator: the numpy twin of snpgpu_synth_block (kernels_prep.hip: synth_block_kernel).
# Every cell is a pure integer function of (seed, snp, sample): full-size GPU runs (N = 100 000 .. 500 000,
# L = 1 000 000) are checked by recomputing a handful of samples here.
_M32 = np.uint32(0xFFFFFFFF)


def _mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def synth_hash_keys(snps, seed):
    """Per-SNP key of the generator for the SNP indices `snps`."""
    s = np.asarray(snps, dtype=np.int64).astype(np.uint32)
    with np.errstate(over="ignore"):
        return _mix32(np.uint32(seed) ^ _mix32(s + np.uint32(0x9E3779B9)))


def synth_hash_threshold(snps, seed, spectrum=0):
    """16-bit allele-frequency threshold t of each SNP (p = t / 65536)."""
    ks = synth_hash_keys(snps, seed)
    u = (_mix32(ks ^ np.uint32(0xA5A5A5A5)) >> np.uint32(16)).astype(np.uint64)
    if spectrum == 1:                       # (spectra 3 and 4 -- population structure, linkage disequilibrium -- build on spectrum 0)
        t = (u * u * u) >> np.uint64(33)
    elif spectrum == 2:
        t = np.uint64(655) + ((u * np.uint64(32113)) >> np.uint64(16))
    else:
        t = np.uint64(3277) + ((u * np.uint64(58982)) >> np.uint64(16))
    return t.astype(np.uint32)


def synth_hash_geno(samples, snp_begin, n_snp, seed, missing=0.0, spectrum=0, special=False):
    """uint8 [n_snp][len(samples)] genotypes (3 = missing) of the listed samples for SNPs
    [snp_begin, snp_begin + n_snp) -- bit-identical to what snpgpu_synth_block writes."""
    samples = np.asarray(samples, dtype=np.int64)
    snps = np.arange(snp_begin, snp_begin + n_snp, dtype=np.int64)
    ks = synth_hash_keys(snps, seed)[:, None]
    t = synth_hash_threshold(snps, seed, spectrum)[:, None]
    with np.errstate(over="ignore"):
        sm = samples.astype(np.uint32) * np.uint32(0x9E3779B1)
    h = _mix32(ks ^ sm[None, :])
    if spectrum == 3:
        # three sub-populations (sample % 3), Fst ~ 0.1: per-population thresholds around the ancestral one
        t0 = synth_hash_threshold(snps, seed, 0).astype(np.int64)
        r = np.array([math.isqrt(int(v)) for v in (t0 * (65536 - t0)) // 10], dtype=np.int64)
        tk = np.empty((len(snps), 3), np.int64)
        for k in range(3):
            with np.errstate(over="ignore"):
                hk = _mix32(ks[:, 0] ^ np.uint32((0x0051ED27 + k * 0x01234567) & 0xFFFFFFFF)).astype(np.int64)
            zi = (hk & 0xFF) + ((hk >> 8) & 0xFF) + ((hk >> 16) & 0xFF) + (hk >> 24) - 510
            tk[:, k] = np.clip(t0 + (zi * r) // 148, 655, 64880)
        tt = tk[:, samples % 3].astype(np.uint32)
        g = ((h & np.uint32(0xFFFF)) < tt).astype(np.uint8) + ((h >> np.uint32(16)) < tt).astype(np.uint8)
    elif spectrum == 4:
        # LD blocks of 48 SNPs, 6 founder haplotypes per block, 2 % of the haplotype alleles drawn independently
        t0 = synth_hash_threshold(snps, seed, 0)[:, None]
        founders = np.zeros((len(snps), 1), np.uint32)
        for f in range(6):
            with np.errstate(over="ignore"):
                c = np.uint32((f * 0x85EBCA6B + 0x1B873593) & 0xFFFFFFFF)
            founders |= ((_mix32(ks ^ c) & np.uint32(0xFFFF)) < t0).astype(np.uint32) << np.uint32(f)
        with np.errstate(over="ignore"):
            kb = _mix32(np.uint32(seed) ^ _mix32((snps // 48).astype(np.uint32) + np.uint32(0x7F4A7C15)))[:, None]
        fb = _mix32(kb ^ sm[None, :])
        nz = _mix32(h ^ np.uint32(0x3C6EF372))
        a1 = np.where((nz & np.uint32(0xFFFF)) < np.uint32(1311), (h & np.uint32(0xFFFF)) < t0, ((founders >> ((fb & np.uint32(0xFFFF)) % np.uint32(6))) & np.uint32(1)) != 0)
        a2 = np.where((nz >> np.uint32(16)) < np.uint32(1311), (h >> np.uint32(16)) < t0,
                      ((founders >> ((fb >> np.uint32(16)) % np.uint32(6))) & np.uint32(1)) != 0)
        g = a1.astype(np.uint8) + a2.astype(np.uint8)
    else:
        g = ((h & np.uint32(0xFFFF)) < t).astype(np.uint8) + ((h >> np.uint32(16)) < t).astype(np.uint8)
    miss32 = int(np.floor(missing * 4294967296.0))
    if miss32:
        g[_mix32(h ^ np.uint32(0x68E31DA4)) < np.uint32(miss32)] = 3
    if special:
        m = snps % 997
        g[m == 3] = 0
        g[m == 5] = 2
        g[m == 7] = 3
    return g


def synth_hash_block_packed(n_samp, snp_begin, n_snp, seed, missing=0.0, spectrum=0, special=False):
    """The whole block as SNPGPU_GENO_PACKED2 rows [n_snp][ceil(n_samp/4)] (small sizes only)."""
    g = synth_hash_geno(np.arange(n_samp), snp_begin, n_snp, seed, missing, spectrum, special)
    nb = (n_samp + 3) // 4
    pad = np.full((n_snp, nb * 4), 3, np.uint8)
    pad[:, :n_samp] = g
    q = pad.reshape(n_snp, nb, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)
