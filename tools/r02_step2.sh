#!/bin/bash
# round 2: default bench line (with sub-results + CPU baseline), kernel trace + HBM counters of the workloads, 2-rank tests
set -u
TAG=${1:-r02c}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
( time python bench.py --steps 20 --warmup 5 ) > "$OUT/bench_default.log" 2>&1
grep '^{' "$OUT/bench_default.log" | tail -1 > "$OUT/bench_default.json"
tail -5 "$OUT/bench_default.log"
python -m pytest tests/test_gpu_multiproc.py -x -q 2>&1 | tail -15
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    rocprofv3 "${pargs[@]}" -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results "$@" > "$OUT/$name.log" 2>&1
}
run grm_trace  --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1
run grmmiss_trace --kernel-trace --stats -- --workload grm --missing 0.02 --steps 3 --warmup 1
run ibs_trace  --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20
run king_trace --kernel-trace --stats -- --workload king --steps 40 --warmup 20
SNPGPU_SYRK=f32 run grmf32_trace --kernel-trace --stats -- --workload grm --steps 2 --warmup 1
for c in FETCH_SIZE WRITE_SIZE; do
    run grm_$c  --kernel-trace --pmc $c -- --workload grm  --steps 2 --warmup 1
    run grmmiss_$c --kernel-trace --pmc $c -- --workload grm --missing 0.02 --steps 2 --warmup 1
    run ibs_$c  --kernel-trace --pmc $c -- --workload ibs  --steps 5 --warmup 2
    run king_$c --kernel-trace --pmc $c -- --workload king --steps 5 --warmup 2
done
cd "$REPO"
{
    for w in grm grmmiss ibs king grmf32; do echo "== $w"; python tools/rocprof_summary.py "$OUT/${w}_trace/${w}_trace_results.db"; done
} > "$OUT/kernel_trace.txt"
for w in grm grmmiss ibs king; do
    for c in FETCH_SIZE WRITE_SIZE; do
        python tools/pmc_summary.py "$OUT/${w}_$c/${w}_${c}_results.db" > "$OUT/pmc_${w}_$c.json"
    done
done
find "$OUT" -name "*.db" -delete
find "$OUT" -type d -empty -delete
cat "$OUT/kernel_trace.txt" | head -80
