#!/usr/bin/env python3
"""Whole-panel distribution of the off-diagonal error figure of the shipped GRM / PCA kernels at configs[2]'s real size
and at the benchmarked block size (65 536-SNP feed blocks since round 4; --block 32768: rounds 2-3).

Several contexts accumulate the SAME `--rows`-row panel of the 100 000 x 100 000 triangle over every block of the
1 000 000-SNP synthetic data set:
    default   the path bench.py times (single-product kernel, fp32 runs of <= 11 264 slots -- six per 65 536-SNP block -- each with
              its own weight target, for blocks without missing calls; exact-row kernel with 8192-SNP runs otherwise)
    exact_row SNPGPU_SYRK_UV=0 (exact-row kernel for every block)
    fast      SNPGPU_SYRK_FAST=1 (round 2's default: one 32 768-SNP fp32 run per block, one weight target)
    one_target  SNPGPU_UV_TARGETS=0 (the default's run length with ONE weight target for every run)
    ref       SNPGPU_SYRK_UV=0 with SNPGPU_H3_PROMOTE=1024: the exact-row arithmetic (exact row operand x 22-bit column
              operand) promoted to fp64 every 1024 SNPs -- fp32 accumulation error ~ sqrt(1024 / 32768) of the
              shipped kernel's, i.e. a device-side stand-in for the fp64 definition that covers EVERY entry of the panel
and the figure  |x - ref| / (|ref| + median |ref|)  (tests/norms.py `offdiag`) is reduced on the device over all
~4e8 entries: maximum, 99.999th / 99.99th / 99.9th percentile, rms.  Prints one JSON line (and writes it to --out).

Round 5: the FP64 ANCHOR.  The device reference shares tables and operand formats with the kernels it judges, so --anchor K
(default 328) also recomputes K x K sampled entries of the same panel (~1e5) on the CPU in fp64 from the generator's numpy twin
and the reference's definitions (tests/fp64_anchor.py) and reports, with the same denominator, `ref_vs_fp64_max` (how good the
device reference is) and per variant `vs_fp64_max` / `vs_fp64_rms` on those entries next to the whole-panel figures.
    python tools/panel_error_distribution.py --rows 8192 --missing 0
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--snps", type=int, default=1000000)
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--row0", type=int, default=50176)
    ap.add_argument("--block", type=int, default=65536)
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--spectrum", type=int, default=0)
    ap.add_argument("--kind", default="PCA_COV", choices=["PCA_COV", "GRM_GCTA"])
    ap.add_argument("--variants", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--anchor", type=int, default=328, help="K: K x K sampled entries of the panel recomputed in fp64 on the CPU (0: off)")
    ap.add_argument("--only", default="", help="comma-separated subset of the contexts to run next to `ref` (default: all)")
    a = ap.parse_args()
    import torch
    from snprelate_amd import _lib
    n, B, r0, r1 = a.n, a.block, a.row0, min(a.row0 + a.rows, a.n)

    def make(env):
        keep = {k: os.environ.get(k) for k in ("SNPGPU_SYRK_UV", "SNPGPU_H3_PROMOTE", "SNPGPU_SYRK_FAST", "SNPGPU_UV_TARGETS", "SNPGPU_X1_SPARSE")}
        for k in keep:
            os.environ.pop(k, None)
        os.environ.update(env)
        acc = _lib.Accumulator(getattr(_lib, a.kind), n, row_begin=r0, row_end=r1 if (r1 < n or r0 > 0) else 0, max_block_snps=B)
        for k, v in keep.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        return acc

    accs = {"default": make({}), "exact_row": make({"SNPGPU_SYRK_UV": "0"}),
            "fast": make({"SNPGPU_SYRK_FAST": "1"}),                              # round 2's default: one 32 768-SNP run, one weight target
            "one_target": make({"SNPGPU_UV_TARGETS": "0"}),
            "ref": make({"SNPGPU_SYRK_UV": "0", "SNPGPU_H3_PROMOTE": "1024"})}
    if a.missing > 0:                        # rare variants of blocks with missing calls wholly in the dense exact-row product
        accs["rare_variants_dense"] = make({"SNPGPU_X1_SPARSE": "0"})
    if a.only:
        for k in [k for k in accs if k != "ref" and k not in a.only.split(",")]:
            accs.pop(k).close()
    for v in a.variants.split(","):          # e.g. "uv:8192,x1:8192,uv:1024": kernel : fp32 run length
        if v:
            kern, run = v.split(":")
            accs["%s_run%s" % (kern, run)] = make({"SNPGPU_SYRK_UV": "1" if kern == "uv" else "0", "SNPGPU_H3_PROMOTE": run})
    anchor = None
    if a.anchor > 0:
        from fp64_anchor import Fp64Anchor, block_stats_torch
        anchor = Fp64Anchor(n, r0, r1, a.anchor, a.anchor, a.kind, 20240601, a.missing, a.spectrum)
    buf = torch.empty((B, (n + 3) // 4), dtype=torch.uint8, device="cuda")
    for lo in range(0, a.snps, B):
        m = min(B, a.snps - lo)
        _lib.synth_block(buf.data_ptr(), n, lo, m, 20240601, missing=a.missing, spectrum=a.spectrum)
        for acc in accs.values():
            acc.feed_device(buf.data_ptr(), m)
        if anchor is not None:               # (the CPU recomputation of this block's sampled entries runs under the kernels)
            sc = block_stats_torch(buf[:m])
            anchor.add(lo, m, *sc)
    for acc in accs.values():               # the feeds are asynchronous: the block buffer must outlive every context's pre-pass
        acc.sync()
    del buf
    slabs = {}
    for k, acc in accs.items():
        out = torch.empty(acc.slab_size(), dtype=torch.float64, device="cuda")
        if a.kind == "GRM_GCTA":
            acc.grm_gcta(packed=True, out_ptr=out.data_ptr())
        else:
            acc.pca_cov(packed=True, normalize=False, out_ptr=out.data_ptr())
        acc.close()
        slabs[k] = out
    torch.cuda.synchronize()
    ref = slabs["ref"]
    aref = ref.abs()
    sub = aref[torch.randint(0, aref.numel(), (8_000_000,), device="cuda")]
    med = float(sub.median())                                   # median |entry| from an 8e6-entry random sample
    res = {"n": n, "snps": a.snps, "panel_rows": [r0, r1], "entries": int(ref.numel()), "block_snps": B,
           "missing": a.missing, "spectrum": a.spectrum, "kind": a.kind, "median_abs_ref": med,
           "reference": "exact-row kernel promoted to fp64 every 1024 SNPs (SNPGPU_SYRK_UV=0 SNPGPU_H3_PROMOTE=1024)"}
    denom = aref + med
    if anchor is not None:
        idx, f64 = anchor.finish()
        it = torch.from_numpy(idx).to("cuda")
        f64t = torch.from_numpy(f64).to("cuda")
        den64 = f64t.abs() + med
        res["anchor"] = {"entries": int(idx.size), "what": "fp64 on the CPU from oracle/synth.py and the reference's definitions (tests/fp64_anchor.py)",
                         "ref_vs_fp64_max": float(((ref[it] - f64t).abs() / den64).max()),
                         "ref_vs_fp64_rms": float(((ref[it] - f64t) / den64).pow(2).mean().sqrt())}
    for k in [x for x in slabs if x != "ref"]:
        fig = (slabs[k] - ref).abs_() / denom
        ent = fig.numel()
        top = torch.topk(fig, max(1, ent // 1000)).values        # the largest 0.1 %, descending
        res[k] = {"offdiag_max": float(top[0]), "p99_999": float(top[max(0, ent // 100000 - 1)]),
                  "p99_99": float(top[max(0, ent // 10000 - 1)]), "p99_9": float(top[-1]),
                  "rms": float(fig.pow(2).mean().sqrt()), "above_1e-5": int((fig > 1e-5).sum())}
        if anchor is not None:
            d64 = (slabs[k][it] - f64t).abs() / den64
            res[k].update({"vs_fp64_max": float(d64.max()), "vs_fp64_rms": float(d64.pow(2).mean().sqrt()),
                           "vs_ref_max_on_the_anchor_entries": float(fig[it].max())})
        del fig, top
    line = json.dumps(res, sort_keys=True)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
