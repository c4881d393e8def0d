#!/bin/bash
# Round 6: counters of syrk_uv_kernel against syrk_uv16_kernel (separate --pmc passes, --kernel-trace only) -> gpurun_out/r06_uv16_pmc.txt
set -u
OUT=$PWD/gpurun_out/r06_uv16_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS")
for v in 0 1; do i=0
  for s in "${SETS[@]}"; do name=uv16_${v}_$i
    SNPGPU_SYRK_UV16=$v rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --no-pmc --no-probe --no-telemetry --samples ${N:-50000} --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"; i=$((i+1))
  done
done
cd "$REPO"
python - "$OUT" <<'PY' | tee gpurun_out/r06_uv16_pmc.txt
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/uv16_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print("bad", f, e); continue
    for k, cs in d.items():
        if "syrk_uv" in k:
            print(f.split("/")[-1], k[:40], {c: round(v["mean"]) for c, v in cs.items()})
PY
