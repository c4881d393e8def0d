#!/bin/bash
# utilisation counters of the two-product counter kernel (IBS, blocks without missing calls) + clock / power while it runs
set -u
OUT=$PWD/gpurun_out/r02ibs
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
i=0
for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    name=ibs_$i
    rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --workload ibs --steps 5 --warmup 2 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"; i=$((i+1))
done
cd "$REPO"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r02ibs/ibs_*.json")):
    d = json.load(open(f))
    for k, cs in d.items():
        if "pair_mfma_i8_kernel<5>" in k: print(f.split("/")[-1], k[:40], {c: v["mean"] for c, v in cs.items()})
PY
rm -f "$OUT"/*.log
bash tools/clock_watch.sh r02ibs_clk 2>&1 | grep "clocks_ibs\|clocks_king" 
