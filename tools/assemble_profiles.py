#!/usr/bin/env python3
"""Copy one round's condensed measurements from gpurun_out/<tag>/ (tools/profile_round.sh, clock_watch.sh,
profile_mfma_util.sh) into profiles/ (tracked):  python tools/assemble_profiles.py <tag>"""
import glob
import json
import re
import shutil
import statistics as st
import sys

tag = sys.argv[1]
D = "gpurun_out/%s/" % tag
p = "profiles/r01_pmc_hbm_traffic.json"
d = json.load(open(p))
m = {"grm": ("grm_n100000_b16384", "void syrk_h3_kernel<2, true>"), "ibs": ("ibs_n10000_b65536", "void pair_mfma_i8_kernel<5>"),
     "king": ("king_n10000_b65536", "void pair_mfma_i8_kernel<1>")}
for w, (key, k) in m.items():
    d.setdefault(key, {})
    f = json.load(open(D + "pmc_%s_FETCH_SIZE.json" % w))[k]["FETCH_SIZE"]
    wr = json.load(open(D + "pmc_%s_WRITE_SIZE.json" % w))[k]["WRITE_SIZE"]
    e = d[key]
    e["kernel"] = k.replace("void ", "")
    e["FETCH_SIZE_KiB_per_launch"], e["FETCH_SIZE_launches"] = f["mean"], f["launches"]
    e["WRITE_SIZE_KiB_per_launch"], e["WRITE_SIZE_launches"] = wr["mean"], wr["launches"]
    e["hbm_bytes_per_launch_raw"] = (f["mean"] + wr["mean"]) * 1024
    print(key, "%.4g bytes per launch" % e["hbm_bytes_per_launch_raw"])
json.dump(d, open(p, "w"), indent=1)
with open("profiles/r01_clock_power.txt", "w") as out:
    out.write("# tools/clock_watch.sh: shader clock / socket power (rocm-smi, 0.2 s period) while `bench.py --workload W` runs; "
              "ramp-up samples dropped\n")
    for f in sorted(glob.glob(D + "clocks_*.txt")):
        sclk, pw = [], []
        for line in open(f):
            i = line.find("card0")
            if i < 0:
                continue
            v = line[i:].strip().split(",")
            mm = re.search(r"(\d+)Mhz", v[5])
            try:
                watts = float(v[9])
            except (ValueError, IndexError):
                continue
            if mm and (int(mm.group(1)) > 1200 or "idle" in f):
                sclk.append(int(mm.group(1)))
                pw.append(watts)
        if "idle" not in f:
            sclk, pw = sclk[3:-1], pw[3:-1]
        if sclk:
            line = "%s: sclk MHz median %.0f (min %.0f max %.0f) | power W median %.0f max %.0f | samples %d\n" % (
                f.split("/")[-1].replace("clocks_", "").replace(".txt", ""), st.median(sclk), min(sclk), max(sclk),
                st.median(pw), max(pw), len(sclk))
            out.write(line)
            print(line, end="")
    out.write("# earlier run, three-product SYRK for every block (SNPGPU_SYRK=h3): grm sclk median 1815, power median 1380 max 1393\n")
    out.write("# earlier run, IBS without missing calls with {-1,0,1} operands: sclk median 2244-2291, power 1376-1396 (at the cap)\n")
shutil.copy(D + "kernel_trace.txt", "profiles/r01_kernel_trace_final.txt")
shutil.copy(D + "bench_lines.jsonl", "profiles/r01_bench_lines_final.jsonl")
shutil.copy(D + "mfma_util.json", "profiles/r01_mfma_util_counters.json")
for l in open("profiles/r01_bench_lines_final.jsonl"):
    b = json.loads(l)
    r = b["roofline"]
    print(b["config"]["workload"][:40], "%.4g" % b["value"], "%.2f ms/step" % b["ms_per_step"],
          {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
u = json.load(open("profiles/r01_mfma_util_counters.json"))
for k, v in u.items():
    if v.get("GRBM_GUI_ACTIVE", 0) > 1e6:
        print(k, "mfma busy %.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)),
              "valu/mfma %.2f" % ((v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"]),
              "lds/mfma %.2f" % (v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"]))
