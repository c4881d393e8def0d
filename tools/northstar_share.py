#!/usr/bin/env python3
"""One rank's share of the north_star job (GRM / PCA covariance + top-32 eigenvectors, 500 000 samples x 1 000 000 SNPs on
8 GPUs), measured on ONE MI355X: rank `--rank` of the `--world`-rank equal-area row-panel plan accumulates a few
16 384-SNP blocks (all ranks see every block; there is no collective on the data path, so the 8-GPU step time is the
slowest rank's step time), then the same panel is used for the eigen solver's building block, Y += C Q with a
40-column block (snpgpu_pca_panel_matmul: one pass over the fp64 panel).  Prints one JSON line.
    python tools/northstar_share.py --rank 0 --world 8 --steps 4
"""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--panels-per-rank", type=int, default=1)
    ap.add_argument("--kind", default="PCA_COV", choices=["PCA_COV", "GRM_GCTA", "KING_ROBUST"])
    ap.add_argument("--block", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--matmul-cols", type=int, default=40)
    a = ap.parse_args()
    import torch
    from snprelate_amd import _lib
    from snprelate_amd.dist import panel_plan, pass_plan
    n, B = a.n, a.block
    if a.kind == "KING_ROBUST":      # configs[4]: four passes over the SNP stream (20 B of counters per pair)
        bounds, owned_q, _ = pass_plan(n, a.world, a.panels_per_rank, 4, 20.0)
        owned = owned_q[0]
    else:
        bounds, owned = panel_plan(n, a.world, a.panels_per_rank)
    panels = owned[a.rank]
    accs = []
    for p in panels:
        r0, r1 = bounds[p], bounds[p + 1]
        accs.append(_lib.Accumulator(getattr(_lib, a.kind), n, row_begin=r0, row_end=r1 if r1 < n or r0 > 0 else 0,
                                     max_block_snps=B))
    free, total = torch.cuda.mem_get_info()
    buf = [torch.empty((B, (n + 3) // 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for i, t in enumerate(buf):
        _lib.synth_block(t.data_ptr(), n, i * B, B, 20240601, missing=a.missing)

    def step(i):
        for acc in accs:
            acc.feed_device(buf[i % 2].data_ptr(), B)

    step(0)
    for acc in accs:
        acc.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i + 1)
    for acc in accs:
        acc.sync()
    dt = (time.perf_counter() - t0) / a.steps
    pairs = sum((bounds[p + 1] - bounds[p]) * n - (bounds[p] + bounds[p + 1] - 1) * (bounds[p + 1] - bounds[p]) / 2.0 for p in panels)
    out = {"kind": a.kind, "n": n, "world": a.world, "rank": a.rank, "panels": [[bounds[p], bounds[p + 1]] for p in panels],
           "pairs_of_this_rank": pairs, "share_of_triangle": pairs / (n * (n + 1) / 2.0), "block_snps": B,
           "missing": a.missing, "ms_per_block": dt * 1e3, "pair_genotypes_per_s_this_gpu": pairs * B / dt,
           "hbm_used_GiB": (total - free) / 2**30,
           "projected_s_for_1e6_snps": dt * 1e6 / B}
    if a.kind == "PCA_COV" and a.matmul_cols > 0:
        m = a.matmul_cols
        q = torch.randn((m, n), dtype=torch.float64, device="cuda")        # column-major n x m
        y = torch.zeros_like(q)
        for acc in accs:
            acc.pca_panel_matmul(1.0, q.data_ptr(), m, y.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            for acc in accs:
                acc.pca_panel_matmul(1.0, q.data_ptr(), m, y.data_ptr())
        torch.cuda.synchronize()
        mm = (time.perf_counter() - t0) / reps
        out.update({"panel_matmul_ms": mm * 1e3, "panel_matmul_cols": m,
                    "panel_matmul_TBps": sum(a_.slab_size() for a_ in accs) * 8 / mm / 1e12})
    print(json.dumps(out))
    for acc in accs:
        acc.close()


if __name__ == "__main__":
    main()
