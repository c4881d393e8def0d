#!/bin/bash
# A/B of the super-tile edge of the exact-row SYRK's work list: step time and L2->fabric word traffic (FETCH_SIZE)
set -u
OUT=$PWD/gpurun_out/${1:-r02d}
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
for S in 4 8 2 16; do
  for rep in 1 2; do
    SNPGPU_H3_SUPER=$S python bench.py --no-cpu-baseline --no-sub-results --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('super $S | ms/step %.3f | ms/launch %.3f' % (d['config']['steps_only_ms_per_step'], d['roofline']['ms_per_launch']))"
  done
  ( cd /tmp; SNPGPU_H3_SUPER=$S rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f$S" -o f$S -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --steps 2 --warmup 1 > /dev/null 2>&1 )
  python tools/pmc_summary.py "$OUT/f$S/f${S}_results.db" | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    if 'syrk' in k: print('super $S FETCH_SIZE KiB', v['FETCH_SIZE']['mean'])"
  rm -rf "$OUT/f$S"
done 2>&1 | tee "$OUT/super_ab.txt"
