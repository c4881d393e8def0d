#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun):  bash tools/profile_r04.sh [part ...]   (parts: bench trace pmc util acc fullsize eigen; default all)
#   rocprofv3 --kernel-trace --stats of the bench.py workloads (no other trace domain), HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in
#   separate runs), matrix-pipe / LDS counters of the headline kernel, whole-panel accuracy distributions on six spectra, the full-size parity
#   tests' error figures, the north-star rehearsal; condensed on the box into gpurun_out/r04prof/ (the result databases are too large to travel).
#   Every output is tied to the tree it came from by gpurun_out/r04prof/stamp.txt = `python bench.py --stamp` ON THE BOX;
#   tools/assemble_profiles_r04.py refuses to copy anything into profiles/ unless that equals the local tree's stamp.
set -u
OUT=$PWD/gpurun_out/r04prof
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
PARTS="${*:-bench trace pmc util acc fullsize eigen}"
python bench.py --stamp > "$OUT/stamp.txt"
sha256sum snprelate_amd/libsnpgpu.so | cut -c1-16 > "$OUT/so_sha16.txt"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    ( cd /tmp && rocprofv3 "${pargs[@]}" -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --no-pmc "$@" > "$OUT/$name.log" 2>&1 )
    grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json"
}
if has bench; then
    # the driver's command (defaults) and its usual step counts: the line as the driver will see it, traffic measured by the run itself
    python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.log" 2> "$OUT/bench_default.err"
    grep '^{' "$OUT/bench_default.log" | tail -1 > "$OUT/bench_default.json"
fi
if has trace; then
    run grm_trace      --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1
    run grmmiss_trace  --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1 --missing 0.02
    run ibs_trace      --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20
    run king_trace     --kernel-trace --stats -- --workload king --steps 40 --warmup 20
    run ibsmiss_trace  --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20 --missing 0.02
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/eig_trace" -o eig_trace -- python "$REPO/tools/northstar_share.py" --kind PCA_COV --block 4096 --steps 1 --matmul-cols 48 > "$OUT/eig_trace.log" 2>&1 )
    grep '^{' "$OUT/eig_trace.log" | tail -1 > "$OUT/eig_trace.json"
    { for w in grm grmmiss ibs ibsmiss king eig; do echo "### $w"; python tools/rocprof_summary.py "$OUT/${w}_trace/${w}_trace_results.db"; done; } > "$OUT/kernel_trace.txt"
fi
if has pmc; then
    for c in FETCH_SIZE WRITE_SIZE; do
        run grm_$c      --kernel-trace --pmc $c -- --workload grm --steps 2 --warmup 1
        run grmmiss_$c  --kernel-trace --pmc $c -- --workload grm --steps 2 --warmup 1 --missing 0.02
    done
    for w in grm grmmiss; do
        for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_summary.py "$OUT/${w}_$c/${w}_${c}_results.db" > "$OUT/pmc_${w}_$c.json"; done
    done
fi
if has util; then
    i=0
    for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
        run util_$i --kernel-trace --pmc $s -- --workload grm --steps 2 --warmup 1
        python tools/pmc_summary.py "$OUT/util_$i/util_${i}_results.db" > "$OUT/util_$i.json"
        i=$((i+1))
    done
    i=0
    for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
        run util_ibs_$i --kernel-trace --pmc $s -- --workload ibs --steps 10 --warmup 5
        python tools/pmc_summary.py "$OUT/util_ibs_$i/util_ibs_${i}_results.db" > "$OUT/util_ibs_$i.json"
        i=$((i+1))
    done
fi
if has acc; then
    # whole-panel error distributions (3.7e8 entries each) at configs[2]'s size: the bench spectrum, rare variants, array-like, the two with
    # missing calls that were the thinnest in round 3, and the two structured generators of round 4 with and without missing calls
    for spec in "0 0" "1 0" "2 0" "2 0.02" "1 0.02" "3 0" "3 0.02" "4 0" "4 0.02"; do
        set -- $spec
        python tools/panel_error_distribution.py --rows 8192 --spectrum $1 --missing $2 --kind GRM_GCTA --variants "uv:8192" \
            --out "$OUT/acc_panel_s$1_m$2.json" > /dev/null 2>> "$OUT/acc.err"
    done
    python tools/panel_error_distribution.py --rows 8192 --row0 0 --spectrum 0 --missing 0 --kind GRM_GCTA --out "$OUT/acc_panel_s0_m0_rows0.json" > /dev/null 2>> "$OUT/acc.err"
    python tools/panel_error_distribution.py --rows 8192 --row0 91904 --spectrum 0 --missing 0 --kind GRM_GCTA --out "$OUT/acc_panel_s0_m0_rows91904.json" > /dev/null 2>> "$OUT/acc.err"
fi
if has fullsize; then
    SNPGPU_REPORT_DIR="$OUT/fullsize" python -m pytest tests/test_gpu_fullsize.py -x -q > "$OUT/fullsize_pytest.log" 2>&1
    tail -3 "$OUT/fullsize_pytest.log"
fi
if has eigen; then
    python tools/northstar_rehearsal.py --mode whole --out "$OUT/northstar_whole_150000.json" > "$OUT/northstar_whole.log" 2>&1
    python tools/northstar_rehearsal.py --mode whole --missing 0.02 --out "$OUT/northstar_whole_150000_missing0.02.json" > "$OUT/northstar_whole_miss.log" 2>&1
    python tools/northstar_rehearsal.py --mode share --out "$OUT/northstar_share_500000.json" > "$OUT/northstar_share.log" 2>&1
    python tools/northstar_rehearsal.py --mode check --out "$OUT/northstar_check_12000.json" > "$OUT/northstar_check.log" 2>&1
fi
find "$OUT" -name "*.db" -delete
find "$OUT" -type d -empty -delete
ls "$OUT" | head -80
[ -f "$OUT/kernel_trace.txt" ] && head -40 "$OUT/kernel_trace.txt"
[ -f "$OUT/bench_default.json" ] && python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("bench value %.4g ms/step %.2f frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic")))
for k,v in d.get("sub_results",{}).items(): print("  ",k, "%.4g"%v["value"] if "value" in v else v)
PY
exit 0
