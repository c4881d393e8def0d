#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun):  tools/profile_r03.sh
#   rocprofv3 --kernel-trace --stats of the bench.py workloads (no other trace domain), HBM-traffic PMC passes (FETCH_SIZE and
#   WRITE_SIZE in separate runs), matrix-pipe / LDS counters of the headline kernel; condensed on the box into
#   gpurun_out/r03prof/ (the result databases are too large to travel back).
set -u
OUT=$PWD/gpurun_out/r03prof
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    rocprofv3 "${pargs[@]}" -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results "$@" > "$OUT/$name.log" 2>&1
    grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json"
}
run grm_trace      --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1
run grmmiss_trace  --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1 --missing 0.02
run ibs_trace      --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20
run king_trace     --kernel-trace --stats -- --workload king --steps 40 --warmup 20
for c in FETCH_SIZE WRITE_SIZE; do
    run grm_$c      --kernel-trace --pmc $c -- --workload grm --steps 2 --warmup 1
    run grmmiss_$c  --kernel-trace --pmc $c -- --workload grm --steps 2 --warmup 1 --missing 0.02
done
i=0
for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    run util_$i --kernel-trace --pmc $s -- --workload grm --steps 2 --warmup 1
    i=$((i+1))
done
# the eigen solver's panel product on rank 0's panel of the N = 500 000 plan
rocprofv3 --kernel-trace --stats -d "$OUT/eig_trace" -o eig_trace -- python "$REPO/tools/northstar_share.py" --kind PCA_COV --block 4096 --steps 1 --matmul-cols 48 > "$OUT/eig_trace.log" 2>&1
grep '^{' "$OUT/eig_trace.log" | tail -1 > "$OUT/eig_trace.json"
cd "$REPO"
{
    for w in grm grmmiss ibs king eig; do echo "### $w"; python tools/rocprof_summary.py "$OUT/${w}_trace/${w}_trace_results.db"; done
} > "$OUT/kernel_trace.txt"
for w in grm grmmiss; do
    for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py "$OUT/${w}_$c/${w}_${c}_results.db" > "$OUT/pmc_${w}_$c.json"; done
done
for k in 0 1 2 3; do python tools/pmc_summary.py "$OUT/util_$k/util_${k}_results.db" > "$OUT/util_$k.json"; done
find "$OUT" -name "*.db" -delete
find "$OUT" -type d -empty -delete
cat "$OUT/kernel_trace.txt" | head -60
