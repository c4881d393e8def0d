#!/usr/bin/env python3
"""bench.py -- throughput of the pairwise hot path on MI355X.

Metric (BASELINE.json): SNP-pair-genotypes/sec = N^2 * L / 2 / t.

A "step" is one pass of the hot path over one feed block of B SNPs for ALL sample
pairs (pre-pass + pair kernel + accumulation into the resident N x N panel);
the K timed steps therefore process K*B SNPs of the named configuration.  Inputs
(2-bit packed synthetic genotypes) are resident in HBM before the timed region.

Default workload = BASELINE.json configs[2]: snpgdsGRM method="GCTA", synthetic
N = 100 000 samples (x 1 000 000 SNPs = 62 steps of 16 384 SNPs; the default K
times a slice of that job, every step is identical work).  Other workloads:
  --workload ibs    configs[1]  snpgdsIBSNum   N = 10 000
  --workload king   snpgdsIBDKING robust       N = 10 000, 5 % missing
  --workload pca    snpgdsPCA covariance       N = 100 000
Multi-GPU (--gpus N under torch.distributed.run): the output triangle is cut into
equal-area row panels, one per rank, no collective on the data path; the total
problem is fixed => "strong" scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    #            kind           N        B      missing  metric kernel (0 popcount / 1 syrk)
    "grm":  dict(kind="GRM_GCTA", n=100000, b=16384, missing=0.0, which=1,
                 name="snpgdsGRM method=GCTA, synthetic 100000 x 1000000 (configs[2]), fed in blocks of 16384 SNPs"),
    "pca":  dict(kind="PCA_COV", n=100000, b=16384, missing=0.0, which=1,
                 name="snpgdsPCA covariance, synthetic 100000 samples, blocks of 16384 SNPs"),
    # counter kernels: 65536-SNP feed blocks (the upper clamp of the reference's own block size, src/genIBS.cpp:286-289):
    # one HBM counter update per block, 5.9e14 instead of 5.1e14 (IBS) at 16384
    "ibs":  dict(kind="IBS", n=10000, b=65536, missing=0.0, which=0,
                 name="snpgdsIBSNum, synthetic 10000 x 500000 (configs[1]), fed in blocks of 65536 SNPs"),
    "king": dict(kind="KING_ROBUST", n=10000, b=65536, missing=0.05, which=0,
                 name="snpgdsIBDKING KING-robust, synthetic 10000 samples, 5% missing, blocks of 65536 SNPs"),
}
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2516.6         # dense fp16 MFMA: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz (guide: ~2.5 PF)
PEAK_VALU_TLANEOPS = 78.6             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
POP_OPS = {"IBS": 8, "KING_ROBUST": 11}   # VALU bit-ops per 32 SNP pairs (popcount backend, kernels_pair.hip)
# Measured on this part (tools/mfma_power.sh -> profiles/r01_mfma_power.txt): a register-only MFMA stream is held
# back by the socket power limit as soon as the operands are not zeros (zeros: 2470 TFLOP/s / 4940 TOP/s at 2.39 GHz).
SUSTAINED_F16_TFLOPS = {2: 1840.0, 3: 1689.0}   # row operand in {-1,0,1} (1.85 GHz) / both operands real-valued (1.71 GHz)
SUSTAINED_I8_TOPS = {False: 4129.0, True: 4911.0}  # operands in {-1,0,1}: 2.06 GHz / binary (blocks without missing calls): 2.39 GHz
PEAK_I8_MFMA_TOPS = 5033.0            # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz (= 2x the dense bf16 peak)
I8_SLOTS = {"IBS": 4, "KING_ROBUST": 5}   # int8 dot products per pair-genotype (I8Scheme<> in kernels_pair.hip)


def pmc_traffic(workload, n, b):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_hbm_traffic.json; FETCH_SIZE and WRITE_SIZE collected in separate runs of this
    same command).  None when no measurement for this exact workload/size is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
            tab = json.load(f)
        key = "%s_n%d_b%d" % (workload, n, b)
        return tab[key]["hbm_bytes_per_launch_raw"] if key in tab else None
    except Exception:
        return None


def synth_block_torch(n, b, missing, seed, device):
    """2-bit packed synthetic genotypes [b][ceil(n/4)] on the device (SURVEY.md 8d generator:
    per-SNP p ~ U(0.05, 0.95), Binomial(2, p), iid missing)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nb = (n + 3) // 4
    out = torch.empty((b, nb), dtype=torch.uint8, device=device)
    chunk = max(1, min(b, (1 << 27) // max(n, 1)))
    for s in range(0, b, chunk):
        e = min(b, s + chunk)
        p = torch.rand((e - s, 1), generator=g, device=device) * 0.9 + 0.05
        geno = (torch.rand((e - s, nb * 4), generator=g, device=device) < p).to(torch.uint8)
        geno += (torch.rand((e - s, nb * 4), generator=g, device=device) < p).to(torch.uint8)
        if missing > 0:
            geno[torch.rand((e - s, nb * 4), generator=g, device=device) < missing] = 3
        geno[:, n:] = 3
        geno = geno.view(e - s, nb, 4)
        out[s:e] = geno[:, :, 0] | (geno[:, :, 1] << 2) | (geno[:, :, 2] << 4) | (geno[:, :, 3] << 6)
    return out


def cpu_baseline(kind, target_s=12.0):
    """The CPU oracle (a restatement of the reference's algorithm, 'port') timed on this host's
    cores on a bounded sample of the same workload: a short calibration run picks the number of
    SNPs so that the timed run is ~target_s seconds of CPU work."""
    import oracle as orc
    from oracle.synth import synth_geno
    fn = {"GRM_GCTA": orc.grm_gcta, "PCA_COV": orc.pca_cov, "IBS": orc.ibs_count,
          "KING_ROBUST": orc.king_robust_count}[kind]
    missing = 0.05 if kind == "KING_ROBUST" else 0.0
    n = 6000
    cal = synth_geno(n, 512, missing=missing, seed=7, special=False)
    fn(cal[:64])                      # warm the library / thread pool
    t0 = time.perf_counter()
    fn(cal)
    rate = n * n * 512 / 2 / (time.perf_counter() - t0)
    L = int(min(max(target_s * rate / (n * n / 2), 1024), 262144))
    L = (L + 255) // 256 * 256
    g = synth_geno(n, L, missing=missing, seed=8, special=False)
    t0 = time.perf_counter()
    fn(g)
    dt = time.perf_counter() - t0
    return {"value": n * n * L / 2 / dt, "unit": "SNP-pair-genotypes/s",
            "cores": orc.num_threads(), "kind": "port",
            "sample": "oracle %s (C + OpenMP restatement of the reference algorithm) on synthetic "
                      "%d samples x %d SNPs, %.1f s" % (fn.__name__, n, L, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 8 (grm, pca) / 50 (ibs, king)")
    ap.add_argument("--warmup", type=int, default=None, help="default 2 (grm, pca) / 20 (ibs, king: ms-scale "
                    "steps, the first ~10 ms on an idle GPU run at a lower clock)")
    ap.add_argument("--workload", default="grm", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", "--n", dest="n", type=int, default=0, help="override the number of samples (not the named config)")
    ap.add_argument("--block", type=int, default=0, help="override SNPs per step")
    ap.add_argument("--missing", type=float, default=None, help="override the missing-call rate of the synthetic data")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--feed", default="device", choices=["device", "pinned_u8", "pinned_2bit"],
                    help="device: blocks resident in HBM (the metric). pinned_*: blocks come from page-locked host "
                         "memory through snpgpu_feed(SNPGPU_HOST_PINNED) -- the PCIe-inclusive rate of the R reader path")
    args = ap.parse_args()
    quick = args.workload in ("ibs", "king")
    if args.steps is None:
        args.steps = 50 if quick else 8
    if args.warmup is None:
        args.warmup = 20 if quick else 2

    import torch
    from snprelate_amd import _lib
    from snprelate_amd.dist import panel_rows

    wl = dict(WORKLOADS[args.workload])
    if args.missing is not None:
        wl["missing"] = float(args.missing)
    if args.n:
        wl["n"] = args.n
        wl["name"] += " [OVERRIDE n=%d]" % args.n
    if args.block:
        wl["b"] = args.block
    n, B = wl["n"], wl["b"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_multiproc.py runs 2 ranks on the single test GPU over gloo)
    backend = os.environ.get("SNPGPU_BENCH_BACKEND", "nccl")
    if "SNPGPU_BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["SNPGPU_BENCH_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    bounds = panel_rows(n, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    kind = getattr(_lib, wl["kind"])
    acc = _lib.Accumulator(kind, n, device=local, row_begin=r0, row_end=(r1 if r1 != n or r0 != 0 else 0),
                           max_block_snps=B) if r1 > r0 else None

    n_blocks = max(1, min(args.steps + args.warmup, 3))
    blocks = [synth_block_torch(n, B, wl["missing"], 20240601 + i, device) for i in range(n_blocks)]
    torch.cuda.synchronize()

    pinned = []
    if args.feed != "device":
        from snprelate_amd.gds import unpack_2bit_rows
        for blk in blocks[:2]:
            h = blk.cpu().numpy()
            if args.feed == "pinned_u8":
                h = unpack_2bit_rows(h, n)
            pb = _lib.PinnedBuffer(h.shape)
            pb.array[:] = h
            pinned.append(pb)

    def step(i):
        if acc is None:
            return
        if pinned:
            pb = pinned[i % len(pinned)]
            acc.host_wait(pb)
            acc.feed_pinned(pb, B, _lib.GENO_U8 if args.feed == "pinned_u8" else _lib.GENO_PACKED2)
        else:
            acc.feed_device(blocks[i % n_blocks].data_ptr(), B)

    def fence():
        if acc is not None:
            acc.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()
    if acc is not None:
        acc.set_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    kms, klaunch = acc.get_timing(wl["which"]) if acc is not None else (0.0, 0)
    if acc is not None:
        acc.set_timing(False)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # one finalise (+ gather when sharded) outside the timed region, reported for information
    fin_ms = None
    if acc is not None and n <= 20000:
        t1 = time.perf_counter()
        if wl["kind"] == "IBS":
            acc.ibs_num(packed=True)
        elif wl["kind"] == "KING_ROBUST":
            acc.king_robust(packed=True)
        elif wl["kind"] == "GRM_GCTA":
            acc.grm_gcta(packed=True)
        else:
            acc.pca_cov(packed=True, normalize=acc.full)
        fin_ms = (time.perf_counter() - t1) * 1e3

    pairs_total = n * n / 2.0
    value = pairs_total * B * args.steps / dt
    out = None
    if rank == 0:
        # roofline of the dominant kernel on this rank's panel
        my_pairs = (r1 - r0) * n - (r0 + r1 - 1) * (r1 - r0) / 2.0
        per_launch_ms = kms / max(klaunch, 1)
        if wl["which"] == 1:
            flops = 2.0 * my_pairs * B                       # 2 flop per pair-genotype
            achieved = flops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
            if os.environ.get("SNPGPU_SYRK", "") == "f32":
                peak, kname, extra = PEAK_F32_MFMA_TFLOPS, "syrk_mfma_kernel", {}
            else:
                # split-fp16 SYRK: blocks without missing calls run (g - 1) (hi + lo) -> 2 executed MFMA flops per
                # algorithmic flop; blocks with missing calls (or SNPGPU_SYRK=h3) hi hi' + hi lo' + lo hi' -> 3
                execd = 2 if (wl["missing"] == 0.0 and os.environ.get("SNPGPU_SYRK", "") != "h3") else 3
                peak, kname = PEAK_F16_MFMA_TFLOPS, ("syrk_h3_kernel<2, true>" if execd == 2 else "syrk_h3_kernel<3, false>")
                # what a register-only stream of the same MFMA sustains under the socket power cap with operands
                # like this kernel's (tools/mfma_power.sh, profiles/r01_mfma_power.txt); zero operands reach `peak`
                sustained = SUSTAINED_F16_TFLOPS[execd]
                extra = {"executed_per_algorithmic": execd, "executed_frac": execd * achieved / peak,
                         "sustained_peak_measured": sustained, "executed_frac_of_sustained": execd * achieved / sustained,
                         "algorithmic_vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS}
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak,
                    "traffic": pmc_traffic(args.workload, n, B) if world == 1 else None,
                    "kernel": kname, "ms_per_launch": per_launch_ms, "launches": klaunch}
            roof.update(extra)
        elif os.environ.get("SNPGPU_PAIR_BACKEND", "") == "popcount":
            ops = POP_OPS[wl["kind"]] * my_pairs * B / 32.0  # VALU lane-ops per launch
            achieved = ops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
            roof = {"bound": "valu", "achieved": achieved, "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
                    "frac": achieved / PEAK_VALU_TLANEOPS, "traffic": None,
                    "kernel": "pair_popcount_kernel", "ms_per_launch": per_launch_ms, "launches": klaunch}
        else:
            slots = I8_SLOTS[wl["kind"]]
            if wl["missing"] == 0.0 and "SNPGPU_I8_NO_NOMISS" not in os.environ:
                slots = 3                                    # blocks without missing calls: binary h.h', e0.e2', e2.e0' 
            ops = 2.0 * slots * my_pairs * B                 # int8 multiply-adds x 2 per launch
            achieved = ops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s",
                    "frac": achieved / PEAK_I8_MFMA_TOPS,
                    "traffic": pmc_traffic(args.workload, n, B) if world == 1 else None,
                    "kernel": "pair_mfma_i8_kernel", "ms_per_launch": per_launch_ms, "launches": klaunch,
                    "products_per_pair_genotype": slots,
                    # register-only i8 MFMA stream with like operands under the power cap (tools/mfma_power.sh)
                    "sustained_peak_measured": SUSTAINED_I8_TOPS[slots == 3],
                    "frac_of_sustained": achieved / SUSTAINED_I8_TOPS[slots == 3]}
        out = {
            "metric": "SNP-pair-genotypes/sec (N^2*L/2/t)", "value": value, "unit": "SNP-pair-genotypes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": ("f32 MFMA + f64 accumulate" if os.environ.get("SNPGPU_SYRK", "") == "f32" else
                                             "f16 hi/lo split operands (22-bit) MFMA + f32/f64 accumulate") if wl["which"] == 1 else "i8 MFMA + i32 accumulate"
            if os.environ.get("SNPGPU_PAIR_BACKEND", "") != "popcount" else "u32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "n_samples": n, "snps_per_step": B,
                       "missing_rate": wl["missing"], "parallelism": "row-panel x%d" % world, "feed": args.feed,
                       "finalize_ms": fin_ms},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl["kind"])
        print(json.dumps(out))
    if acc is not None:
        acc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
