#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r02g
mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
SNPGPU_LIB=$PWD/snprelate_amd/libsnpgpu_c512.so python -m pytest tests/test_gpu_parity.py -q -x -k "grm_gcta or pca_cov or ragged" 2>&1 | tail -3
bash tools/bench_lib.sh "--no-sub-results --steps 8 --warmup 2" libsnpgpu.so libsnpgpu_c512.so
cd /tmp
i=0
for s in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    name=x1_$i
    rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --workload grm --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"; i=$((i+1))
done
cd "$REPO"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r02g/x1_*.json")):
    d = json.load(open(f))
    for k, cs in d.items():
        if "syrk" in k: print(f.split("/")[-1], k, {c: v["mean"] for c, v in cs.items()})
PY
rm -f "$OUT"/*.log
