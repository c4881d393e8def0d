#!/bin/bash
# Kernel traces of the variants that run on blocks WITH missing calls (GPU box):  tools/profile_missing.sh <tag>
#   -> gpurun_out/<tag>/kernel_trace_missing.txt
set -u
TAG=${1:-miss}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/grm_m" -o grm_m -- python "$REPO/bench.py" --no-cpu-baseline --workload grm --missing 0.02 --steps 3 --warmup 1 > "$OUT/grm_m.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/ibs_m" -o ibs_m -- python "$REPO/bench.py" --no-cpu-baseline --workload ibs --missing 0.02 --steps 20 --warmup 10 > "$OUT/ibs_m.log" 2>&1
cd "$REPO"
{
    echo "# bench.py --workload grm --missing 0.02 --steps 3 --warmup 1"
    python tools/rocprof_summary.py "$OUT/grm_m/grm_m_results.db"
    grep '^{' "$OUT/grm_m.log" | tail -1 | cut -c1-420
    echo "# bench.py --workload ibs --missing 0.02 --steps 20 --warmup 10"
    python tools/rocprof_summary.py "$OUT/ibs_m/ibs_m_results.db"
    grep '^{' "$OUT/ibs_m.log" | tail -1 | cut -c1-420
} > "$OUT/kernel_trace_missing.txt"
find "$OUT" -name "*.db" -delete
rm -f "$OUT"/*.log
cat "$OUT/kernel_trace_missing.txt"
