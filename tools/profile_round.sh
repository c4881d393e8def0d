#!/bin/bash
# Round profile set (run on the GPU box through gpurun):  tools/profile_round.sh <tag>
#   kernel traces (rocprofv3 --kernel-trace --stats) and HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE in
#   separate runs, kernel-trace only) of the bench.py workloads; results under gpurun_out/<tag>/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    rocprofv3 "${pargs[@]}" -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1
    grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json"
}
run grm_trace  --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1
run ibs_trace  --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20
run king_trace --kernel-trace --stats -- --workload king --steps 40 --warmup 20
for c in FETCH_SIZE WRITE_SIZE; do
    run grm_$c  --kernel-trace --pmc $c -- --workload grm  --steps 2 --warmup 1
    run ibs_$c  --kernel-trace --pmc $c -- --workload ibs  --steps 5 --warmup 2
    run king_$c --kernel-trace --pmc $c -- --workload king --steps 5 --warmup 2
done
# plain bench lines (no profiler attached)
cd "$REPO"
for w in grm ibs king; do
    if [ $w = grm ]; then a="--steps 8 --warmup 2"; else a="--steps 50 --warmup 30"; fi
    python bench.py --workload $w $a > "$OUT/bench_$w.log" 2>&1
    grep '^{' "$OUT/bench_$w.log" | tail -1 >> "$OUT/bench_lines.jsonl"
done
# condense on the box (the result databases are too large to travel back), then drop them
{
    for w in grm ibs king; do python tools/rocprof_summary.py "$OUT/${w}_trace/${w}_trace_results.db"; done
} > "$OUT/kernel_trace.txt"
for w in grm ibs king; do
    for c in FETCH_SIZE WRITE_SIZE; do
        python tools/pmc_summary.py "$OUT/${w}_$c/${w}_${c}_results.db" > "$OUT/pmc_${w}_$c.json"
    done
done
find "$OUT" -name "*.db" -delete
rm -f "$OUT"/*.log
cat "$OUT/kernel_trace.txt"
