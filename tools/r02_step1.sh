#!/bin/bash
# round 2, first measurement pass: parity suite, A/B of the exact-row SYRK changes, LDS / MFMA counters of the GRM step
set -u
OUT=$PWD/gpurun_out/${1:-r02a}
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$OUT/pytest.txt"
cat "$OUT/pytest.txt"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$1', '| value %.4g | ms/step %.3f | ms/launch %.3f | %s' % (d['value'], d['ms_per_step'], r['ms_per_launch'], r['kernel']))"; }
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | line "grm default"

  python bench.py --no-cpu-baseline --steps 8 --warmup 2 --missing 0.02 2>/dev/null | tail -1 | line "grm miss0.02 exact-row"
  SNPGPU_SYRK_MISS3=1 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --missing 0.02 2>/dev/null | tail -1 | line "grm miss0.02 three-product"
done | tee "$OUT/ab.txt"
cd /tmp
i=0
for s in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
    name=util_grm_$i
    rocprofv3 --kernel-trace --pmc $s -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --workload grm --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"
    i=$((i+1))
done
for c in FETCH_SIZE WRITE_SIZE; do
    name=grm_$c
    rocprofv3 --kernel-trace --pmc $c -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --workload grm --steps 2 --warmup 1 > "$OUT/$name.log" 2>&1
    python "$REPO/tools/pmc_summary.py" "$OUT/$name/${name}_results.db" > "$OUT/pmc_$name.json" 2>> "$OUT/$name.log" || tail -5 "$OUT/$name.log"
    rm -rf "$OUT/$name"
done
cd "$REPO"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r02b/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "bad", e); continue
    for k, cs in d.items():
        if "syrk_h3" in k:
            print(f.split("/")[-1], k, {c: v["mean"] for c, v in cs.items()})
PY
rm -f "$OUT"/*.log
