#!/usr/bin/env python3
"""Condensed round-3 measurements gpurun_out/r03prof/ (tools/profile_r03.sh) -> profiles/r03_* (tracked)."""
import json
import shutil

D = "gpurun_out/r03prof/"
shutil.copy(D + "kernel_trace.txt", "profiles/r03_kernel_trace.txt")
note = ("raw counter bytes (L2 <-> fabric requests; Infinity-Cache hits included), KiB counters x 1024.  FETCH_SIZE reports half "
        "the bytes of the access widths used here (calibration in DESIGN.md 4.2: fin_kernel<FinGcta> reads 60 GB and reports 30-36 GB), "
        "WRITE_SIZE is exact; the read half of the atomic flushes does not appear in FETCH_SIZE.  One feed block of 32768 SNPs = "
        "`launches_per_feed` launches of the kernel (one per fp32 run, each ending in an fp64 flush of the 5e9-element panel = 40 GB).")
out = {}
for key, w, kern in (("grm_n100000_b32768", "grm", "syrk_uv_kernel"), ("grm_missing_n100000_b32768", "grmmiss", "syrk_x1_kernel")):
    f = json.load(open(D + "pmc_%s_FETCH_SIZE.json" % w))[kern]["FETCH_SIZE"]
    wr = json.load(open(D + "pmc_%s_WRITE_SIZE.json" % w))[kern]["WRITE_SIZE"]
    feeds = 3                                      # --steps 2 --warmup 1
    per_feed = f["launches"] // feeds
    e = {"kernel": kern, "launches_per_feed": per_feed,
         "FETCH_SIZE_KiB_per_launch": f["mean"], "WRITE_SIZE_KiB_per_launch": wr["mean"],
         "hbm_bytes_per_launch_raw": (f["mean"] + wr["mean"]) * 1024 * per_feed,
         "hbm_bytes_per_kernel_launch_raw": (f["mean"] + wr["mean"]) * 1024, "note": note}
    if w == "grmmiss":
        mf = json.load(open(D + "pmc_grmmiss_FETCH_SIZE.json"))["void pair_mfma_i8_kernel<3>"]["FETCH_SIZE"]
        mw = json.load(open(D + "pmc_grmmiss_WRITE_SIZE.json"))["void pair_mfma_i8_kernel<3>"]["WRITE_SIZE"]
        e["both_missing_product_pair_mfma_i8_kernel<3>_bytes_per_launch_raw"] = (mf["mean"] + mw["mean"]) * 1024
    out[key] = e
    print(key, "%.4g bytes per feed block (%d launches)" % (e["hbm_bytes_per_launch_raw"], per_feed))
json.dump(out, open("profiles/r03_pmc_hbm_traffic.json", "w"), indent=1)
u = {}
for k in range(4):
    d = json.load(open(D + "util_%d.json" % k))
    for kern, cs in d.items():
        if kern == "syrk_uv_kernel":
            for c, x in cs.items():
                u[c] = x["mean"]
u["launches_profiled"] = 15
u["derived"] = {"matrix_pipe_busy": u["SQ_VALU_MFMA_BUSY_CYCLES"] / (u["GRBM_GUI_ACTIVE"] / 8 * 1024),
                "valu_per_mfma": (u["SQ_INSTS_VALU"] - u["SQ_INSTS_MFMA"]) / u["SQ_INSTS_MFMA"],
                "lds_per_mfma": u["SQ_INSTS_LDS"] / u["SQ_INSTS_MFMA"],
                "waves_waiting_frac": u["SQ_WAIT_INST_ANY"] / u["SQ_WAVE_CYCLES"],
                "lds_bank_conflict_frac": u["SQ_LDS_BANK_CONFLICT"] / max(u["SQ_LDS_IDX_ACTIVE"], 1)}
json.dump({"syrk_uv_kernel (headline: GRM GCTA, N = 100000, 32768-SNP feed blocks = 5 launches of 8192 slots)": u},
          open("profiles/r03_mfma_util_counters.json", "w"), indent=1)
print(u["derived"])
with open("profiles/r03_bench_lines_profiled.jsonl", "w") as f:
    for w in ("grm", "grmmiss", "ibs", "king", "eig"):
        try:
            f.write(open(D + "%s_trace.json" % w).read().strip() + "\n")
        except Exception:
            pass
