"""fp64 anchor of the whole-panel accuracy figures (test infrastructure; used by tests/test_gpu_fullsize.py and by
tools/panel_error_distribution.py).

The whole-panel distributions compare device results with a device-side reference (the exact-row kernel promoted to fp64 every
1024 SNPs): a systematic error common to both would be invisible.  Here >= 1e5 entries of the SAME panel are recomputed on the CPU
in fp64 from the counter-based generator's numpy twin (oracle/synth.py), straight from the definitions the reference follows
(centring / scaling with missing := mean, src/genPCA.h:93-108, src/genPCA.cpp:98-181; GCTA's denominator walk, :1201-1236),
so that both the device reference and the shipped kernels are measured against something that shares no code, no tables
and no arithmetic with them.

Usage: a = Fp64Anchor(n, r0, r1, n_rows, n_cols, kind); per block a.add(snp_begin, n_snp, s, c) with the block's per-SNP
(sum, number of calls) over ALL n samples; a.finish() -> (packed-slab indices relative to the panel's first entry, fp64 values).
"""
import numpy as np


def tri_index(n, i, j):
    """index of (i, j), i <= j, in the packed upper triangle with diagonal (CdMatTri, src/dGenGWAS.h:556-561)"""
    return j + i * (2 * n - i - 1) // 2


def block_stats_torch(blk):
    """per-SNP (sum of called genotypes, number of calls) of a packed 2-bit device block, int64, by plain torch ops"""
    import torch
    s = torch.zeros(blk.shape[0], dtype=torch.int64, device=blk.device)
    c = torch.zeros_like(s)
    for k in range(4):
        code = (blk >> (2 * k)) & 3
        valid = code != 3
        c += valid.sum(1, dtype=torch.int64)
        s += (code * valid).sum(1, dtype=torch.int64)
    return s.cpu().numpy(), c.cpu().numpy()


class Fp64Anchor:
    def __init__(self, n, r0, r1, n_rows=328, n_cols=328, kind="PCA_COV", seed=20240601, missing=0.0, spectrum=0):
        self.n, self.r0, self.r1, self.kind = n, r0, r1, kind
        self.seed, self.missing, self.spectrum = seed, missing, spectrum
        # rows spread over the panel's rows; columns: a few next to the diagonal of the panel's first rows, the rest spread from
        # the panel's first column to the last sample (entries with column >= row count)
        self.rows = np.unique(np.linspace(r0, r1 - 1, n_rows).astype(np.int64))
        near = np.arange(r0, min(r0 + 24, n))
        self.cols = np.unique(np.r_[near, np.linspace(r0, n - 1, max(n_cols - len(near), 2)).astype(np.int64)])
        self.samp = np.r_[self.rows, self.cols]
        self.num = np.zeros((len(self.rows), len(self.cols)))
        self.den_miss = np.zeros_like(self.num)
        self.n_locus = 0

    def add(self, snp_begin, n_snp, s, c):
        from oracle import synth_hash_geno_c
        g = synth_hash_geno_c(self.samp, snp_begin, n_snp, self.seed, missing=self.missing, spectrum=self.spectrum)
        s = s[:n_snp].astype(np.float64)
        c = c[:n_snp].astype(np.float64)
        avg = np.where(c > 0, s / np.maximum(c, 1), 0.0)
        p = avg / 2
        ok = (p > 0) & (p < 1)
        scale = np.where(ok, 1 / np.sqrt(np.where(ok, p * (1 - p), 1.0)), 0.0)
        z = np.where(g <= 2, (g.astype(np.float64) - avg[:, None]) * scale[:, None], 0.0)
        nr = len(self.rows)
        self.num += z[:, :nr].T @ z[:, nr:]
        if self.kind == "GRM_GCTA":
            poly = (s > 0) & (s < 2 * c)
            self.n_locus += int(poly.sum())
            if self.missing > 0:
                mr = ((g[:, :nr] > 2) & poly[:, None]).astype(np.float64)
                mc = ((g[:, nr:] > 2) & poly[:, None]).astype(np.float64)
                self.den_miss += mr.sum(0)[:, None] + mc.sum(0)[None, :] - mr.T @ mc          # i or j missing

    def finish(self):
        ref = self.num / (2.0 * (self.n_locus - self.den_miss)) if self.kind == "GRM_GCTA" else self.num
        keep = self.cols[None, :] >= self.rows[:, None]
        base = tri_index(self.n, self.r0, self.r0)
        idx = (tri_index(self.n, self.rows[:, None], self.cols[None, :]) - base)[keep]
        return idx, ref[keep]
