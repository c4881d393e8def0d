#!/bin/bash
# MFMA / LDS utilisation counters of the three dominant kernels (run on the GPU box through gpurun):
#   tools/profile_mfma_util.sh <tag>     -> gpurun_out/<tag>/util_<workload>_<set>.json
# Counter passes carry --kernel-trace only (no other trace domain), one small counter set per pass.
set -u
TAG=${1:-util}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"
      "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD")
for w in grm ibs king; do
    if [ $w = grm ]; then a="--steps 2 --warmup 1"; else a="--steps 5 --warmup 2"; fi
    i=0
    for s in "${SETS[@]}"; do
        name=util_${w}_$i
        rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --workload $w $a > "$OUT/$name.log" 2>&1
        python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
        rm -rf "$OUT/$name"
        i=$((i+1))
    done
done
rm -f "$OUT"/*.log
cd "$REPO"
python - "$OUT" <<'PY'
import glob, json, sys
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/util_*.json")):
    w = f.split("util_")[-1].split("_")[0]
    try:
        d = json.load(open(f))
    except Exception as e:
        print("bad", f, e); continue
    for k, cs in d.items():
        if not ("syrk_h3" in k or "pair_mfma_i8_kernel<5>" in k or "pair_mfma_i8_kernel<1>" in k):
            continue
        for c, v in cs.items():
            out.setdefault(w + ":" + k, {})[c] = v["mean"]
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[1] + "/mfma_util.json", "w"), indent=1)
PY
