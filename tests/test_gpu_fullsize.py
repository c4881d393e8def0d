"""Parity at BASELINE.json's full per-step sizes through size-independent properties: sampled
pairs recomputed from the definitions in fp64/int64 numpy, count identities, block additivity.
(The CPU oracle cannot cover N = 10 000 .. 100 000 exhaustively in test time.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _synth(n, L, missing, seed):
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.05, 0.95, size=(L, 1)).astype(np.float32)
    g = (rng.random((L, n), dtype=np.float32) < p).astype(np.uint8)
    g += (rng.random((L, n), dtype=np.float32) < p).astype(np.uint8)
    if missing > 0:
        g[rng.random((L, n), dtype=np.float32) < missing] = 3
    return g


def _tri(n, i, j):
    return j + i * (2 * n - i - 1) // 2


def test_ibs_and_king_10000_sampled_pairs_and_identities():
    from snprelate_amd import _lib
    n, L = 10000, 16384                       # configs[1] sample count, one feed block
    g = _synth(n, L, 0.02, 1)
    gi = g.astype(np.int16)
    valid = g <= 2
    rng = np.random.default_rng(2)
    pairs = [(int(min(a, b)), int(max(a, b))) for a, b in rng.integers(0, n, size=(200, 2))]
    pairs += [(0, 0), (n - 1, n - 1), (0, n - 1), (255, 256), (4095, 4096), (31, 9999)]
    with _lib.Accumulator(_lib.IBS, n, max_block_snps=L) as a:
        a.feed(g)
        i0, i1, i2 = a.ibs_num(packed=True)
    with _lib.Accumulator(_lib.KING_ROBUST, n, max_block_snps=8192) as a:   # two blocks: additivity
        a.feed(g[:8192]); a.feed(g[8192:])
        kc = a.king_robust_counts()
    for i, j in pairs:
        both = valid[:, i] & valid[:, j]
        d = np.abs(gi[:, i] - gi[:, j])
        k = _tri(n, i, j)
        ref = [int(((d == 2) & both).sum()), int(((d == 1) & both).sum()), int(((d == 0) & both).sum())]
        assert [int(i0[k]), int(i1[k]), int(i2[k])] == ref, (i, j)
        het_i, het_j = (gi[:, i] == 1) & both, (gi[:, j] == 1) & both
        refk = [ref[0], int(both.sum()), int((d.astype(np.int64) ** 2 * both).sum()), int(het_i.sum()), int(het_j.sum())]
        assert kc[k].tolist() == refk, (i, j)
    # identities over ALL 50 005 000 pairs
    nvalid = valid.sum(0).astype(np.int64)
    tot = i0.astype(np.int64) + i1 + i2
    assert np.array_equal(tot, kc[:, 1].astype(np.int64))                  # IBS0+IBS1+IBS2 = nLoci (two kernels agree)
    diag = np.array([_tri(n, i, i) for i in range(n)])
    assert np.array_equal(i2[diag], nvalid) and not i0[diag].any() and not i1[diag].any()
    assert np.array_equal(kc[:, 2].astype(np.int64), i1.astype(np.int64) + 4 * i0)   # SumSq = IBS1 + 4 IBS0
    assert int(tot.max()) <= L and np.array_equal(i0.astype(np.uint32), kc[:, 0])


def _grm_block_ref(g, rows, cols):
    """GCTA entries for sample sets rows x cols with allele frequencies over ALL samples."""
    valid = g <= 2
    s = (g * valid).sum(1, dtype=np.int64).astype(np.float64)
    c = valid.sum(1, dtype=np.int64).astype(np.float64)
    avg = np.where(c > 0, s / np.maximum(c, 1), 0.0)
    p = avg / 2
    poly = (s > 0) & (s < 2 * c)
    ok = (p > 0) & (p < 1)
    scale = np.where(ok, 1 / np.sqrt(np.where(ok, p * (1 - p), 1.0)), 0.0)
    zr = np.where(valid[:, rows], (g[:, rows].astype(np.float64) - avg[:, None]) * scale[:, None], 0.0)
    zc = np.where(valid[:, cols], (g[:, cols].astype(np.float64) - avg[:, None]) * scale[:, None], 0.0)
    num = zr.T @ zc
    den = 2.0 * ((valid[:, rows] & poly[:, None]).astype(np.float64).T @ valid[:, cols].astype(np.float64))
    return num / den


@pytest.mark.parametrize("missing", [0.0, 0.01])
def test_grm_100000_panels_vs_fp64_definition(missing):
    """configs[2] sample count: two 256-row panels of the 100 000 x 100 000 triangle."""
    from snprelate_amd import _lib
    from snprelate_amd.dist import slab_range
    n, L = 100000, 4096
    g = _synth(n, L, missing, 3)
    cols = np.r_[np.arange(0, 64), np.arange(50170, 50234), np.arange(n - 64, n)]
    for r0 in (0, 50176):
        with _lib.Accumulator(_lib.GRM_GCTA, n, row_begin=r0, row_end=r0 + 256, max_block_snps=2048) as a:
            a.feed(g[:2048]); a.feed(g[2048:])
            slab = a.grm_gcta(packed=True)
            assert a.slab_size() == slab_range(n, r0, r0 + 256)[1] - slab_range(n, r0, r0 + 256)[0]
        rows = np.arange(r0, r0 + 256, 5)
        ref = _grm_block_ref(g, rows, cols)
        base = _tri(n, r0, r0)
        worst = 0.0
        for a_i, i in enumerate(rows):
            for b_j, j in enumerate(cols):
                if j < i:
                    continue
                got = slab[_tri(n, i, j) - base]
                worst = max(worst, abs(got - ref[a_i, b_j]) / (abs(ref[a_i, b_j]) + 0.02))
        assert worst < 1e-5, worst


def test_grm_full_100000_device_output_sampled():
    """The full configs[2] context (120 GB of accumulators) for one block; the 40 GB packed result
    stays on the device and a random sample of entries is checked."""
    import torch
    from snprelate_amd import _lib
    n, L = 100000, 2048
    g = _synth(n, L, 0.0, 4)
    out = torch.empty(n * (n + 1) // 2, dtype=torch.float64, device="cuda")
    with _lib.Accumulator(_lib.GRM_GCTA, n, max_block_snps=L) as a:
        a.feed(g)
        a.grm_gcta(packed=True, out_ptr=out.data_ptr())
    torch.cuda.synchronize()
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(n, 48, replace=False))
    cols = np.sort(rng.choice(n, 48, replace=False))
    ref = _grm_block_ref(g, rows, cols)
    idx, want = [], []
    for a_i, i in enumerate(rows):
        for b_j, j in enumerate(cols):
            lo, hi = (i, j) if i <= j else (j, i)
            idx.append(_tri(n, int(lo), int(hi)))
            want.append(ref[a_i, b_j])
    got = out[torch.tensor(idx, device="cuda")].cpu().numpy()
    want = np.array(want)
    assert np.max(np.abs(got - want) / (np.abs(want) + 0.02)) < 1e-5
    # trace of the GRM from the device result: sum over the diagonal is finite and ~ n
    diag = out[torch.tensor([_tri(n, i, i) for i in range(0, n, 997)], device="cuda")].cpu().numpy()
    assert np.all(np.isfinite(diag)) and 0.8 < diag.mean() < 1.2
