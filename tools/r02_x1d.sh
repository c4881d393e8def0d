#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r02l
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
bash tools/bench_env.sh "--no-sub-results --steps 8 --warmup 2" "SNPGPU_X1_SUPER=4" "SNPGPU_X1_SUPER=2" "SNPGPU_X1_SUPER=8" "SNPGPU_X1_SUPER=1"
cd /tmp
i=0
for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "FETCH_SIZE"; do
    name=ibs_$i
    rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --workload ibs --steps 5 --warmup 2 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"; i=$((i+1))
done
cd "$REPO"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r02l/ibs_*.json")):
    d = json.load(open(f))
    for k, cs in d.items():
        if "pair_mfma_i8_kernel<5>" in k: print(f.split("/")[-1], k, {c: v["mean"] for c, v in cs.items()})
PY
rm -f "$OUT"/*.log
