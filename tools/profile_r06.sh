#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun):  bash tools/profile_r06.sh [part ...]   (parts: bench trace pmc util acc fullsize eigen ubench multi uvc; default all)
#   rocprofv3 --kernel-trace --stats of the bench.py workloads (no other trace domain), HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in
#   separate runs), matrix-pipe / LDS counters of the headline kernel, whole-panel accuracy distributions on six spectra, the full-size parity
#   tests' error figures, the north-star rehearsal; condensed on the box into gpurun_out/r06prof/ (the result databases are too large to travel).
#   Every output is tied to the tree it came from by gpurun_out/r06prof/stamp.txt = `python bench.py --stamp` ON THE BOX;
#   tools/assemble_profiles_r06.py refuses to copy anything into profiles/ unless that equals the local tree's stamp.
set -u
OUT=$PWD/gpurun_out/r06prof
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
PARTS="${*:-bench trace pmc util acc fullsize eigen ubench multi uvc}"
python bench.py --stamp > "$OUT/stamp.txt"
sha256sum snprelate_amd/libsnpgpu.so | cut -c1-16 > "$OUT/so_sha16.txt"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
run() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    ( cd /tmp && rocprofv3 "${pargs[@]}" -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --no-cpu-baseline --no-sub-results --no-pmc --no-probe "$@" > "$OUT/$name.log" 2>&1 )
    grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json"
}
if has bench; then
    # the driver's command (defaults) and its usual step counts: the line as the driver will see it, traffic measured by the run itself
    python bench.py --steps 20 --warmup 5 --details "$OUT/bench_details.json" > "$OUT/bench_default.log" 2> "$OUT/bench_default.err"
    grep '^{' "$OUT/bench_default.log" | tail -1 > "$OUT/bench_default.json"
fi
if has trace; then
    run grm_trace      --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1
    run grmmiss_trace  --kernel-trace --stats -- --workload grm  --steps 3  --warmup 1 --missing 0.02
    run ibs_trace      --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20
    run king_trace     --kernel-trace --stats -- --workload king --steps 40 --warmup 20
    run ibsmiss_trace  --kernel-trace --stats -- --workload ibs  --steps 40 --warmup 20 --missing 0.02
    run homo_trace     --kernel-trace --stats -- --workload king_homo --steps 20 --warmup 10
    ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/eig_trace" -o eig_trace -- python "$REPO/tools/northstar_share.py" --kind PCA_COV --block 4096 --steps 1 --matmul-cols 48 > "$OUT/eig_trace.log" 2>&1 )
    grep '^{' "$OUT/eig_trace.log" | tail -1 > "$OUT/eig_trace.json"
    { for w in grm grmmiss ibs ibsmiss king homo eig; do echo "### $w"; python tools/rocprof_summary.py "$OUT/${w}_trace/${w}_trace_results.db"; done; } > "$OUT/kernel_trace.txt"
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
    # the counter kernels: the two-product kernel (ibs), and the general kernels of blocks with missing calls (ibsmiss: IBS 2 %, king: KING-robust 5 %)
    for w in "ibs ibs 0" "ibsmiss ibs 0.02" "king king 0.05"; do
        set -- $w
        i=0
        for s in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
            run util_$1_$i --kernel-trace --pmc $s -- --workload $2 --steps 10 --warmup 5 --missing $3
            python tools/pmc_summary.py "$OUT/util_$1_$i/util_$1_${i}_results.db" > "$OUT/util_$1_$i.json"
            i=$((i+1))
        done
    done
fi
if has acc; then
    # whole-panel error distributions (3.7e8 entries each) at configs[2]'s size, spectra 0-4 without and with 2 % missing calls, every
    # figure ALSO against ~1e5 entries recomputed in fp64 on the CPU (--anchor, tests/fp64_anchor.py; round 5)
    for spec in "0 0" "0 0.02" "1 0" "1 0.02" "2 0" "2 0.02" "3 0" "3 0.02" "4 0" "4 0.02"; do
        set -- $spec
        anchor=328; [ "$1" -ge 3 ] && anchor=128      # (the structured generators have no C twin: their fp64 anchor is 128 x 128 entries)
        python tools/panel_error_distribution.py --rows 8192 --spectrum $1 --missing $2 --kind GRM_GCTA --only default,exact_row,fast --anchor $anchor \
            --out "$OUT/acc_panel_s$1_m$2.json" > /dev/null 2>> "$OUT/acc.err"
    done
    python tools/panel_error_distribution.py --rows 8192 --row0 0 --spectrum 0 --missing 0 --kind GRM_GCTA --only default --out "$OUT/acc_panel_s0_m0_rows0.json" > /dev/null 2>> "$OUT/acc.err"
    python tools/panel_error_distribution.py --rows 8192 --row0 91904 --spectrum 0 --missing 0 --kind GRM_GCTA --only default --out "$OUT/acc_panel_s0_m0_rows91904.json" > /dev/null 2>> "$OUT/acc.err"
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
    python tools/northstar_rehearsal.py --mode whole --n 20000 --devices 0,0,0,0,0,0,0,0 --missing 0.01 --out "$OUT/northstar_one_command_20000_x8.json" > "$OUT/northstar_x8.log" 2>&1
fi
if has ubench; then
    # round 6: the K-loop models (32x32x16 / 16x16x32 / int8 form of KING-homo's both-missing contraction) and the rounding premise
    ( cd tools/ubench && [ -x r06_kloop_ubench ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w r06_kloop_ubench.hip -o r06_kloop_ubench )
    tools/ubench/r06_kloop_ubench 40000 > "$OUT/kloop_ubench.txt" 2>&1
    # what the matrix pipe sustains per operand class and instruction shape (snpgpu_diag_mfma_rate), three seconds each
    python - > "$OUT/probe_mfma_shapes.txt" 2>&1 <<PY
from snprelate_amd import _lib
for name, m in (("f16_zero 32x32x16", _lib.DIAG_F16_ZERO), ("f16_uv 32x32x16", _lib.DIAG_F16_UV), ("f16_uv 16x16x32", _lib.DIAG_F16_UV_16X16X32),
                ("f16_exact_row 32x32x16", _lib.DIAG_F16_EXACT_ROW), ("f16_exact_row 16x16x32", _lib.DIAG_F16_EXACT_ROW_16X16X32),
                ("fp4 32x32x64", _lib.DIAG_FP4), ("fp4 16x16x128", _lib.DIAG_FP4_16X16X128)):
    r, mhz = _lib.diag_mfma_rate(m, 3.0)
    print("%-24s %8.1f TFLOP/s  implied shader clock %6.0f MHz" % (name, r, mhz))
PY
fi
if has multi; then
    # round 6: the self-launching multi-rank bench on the one GPU (gloo: RCCL needs one device per rank) -- 2 and 8 ranks, defaults otherwise
    SNPGPU_BENCH_BACKEND=gloo SNPGPU_BENCH_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 4 --warmup 1 --samples 40000 > "$OUT/bench_gpus2_one_device.log" 2> "$OUT/bench_gpus2.err"
    grep '^{' "$OUT/bench_gpus2_one_device.log" | tail -1 > "$OUT/bench_gpus2_one_device.json"
    SNPGPU_BENCH_BACKEND=gloo SNPGPU_BENCH_FORCE_DEVICE=0 python bench.py --gpus 8 --steps 4 --warmup 1 --samples 40000 > "$OUT/bench_gpus8_one_device.log" 2> "$OUT/bench_gpus8.err"
    grep '^{' "$OUT/bench_gpus8_one_device.log" | tail -1 > "$OUT/bench_gpus8_one_device.json"
fi
if has uvc; then
    # round 6 (session 3): the 16x16x32 forms of the single-product kernel on one box, interleaved -- SNPGPU_SYRK_UV16 = 1 syrk_uv16_kernel
    # (operands from LDS tables), 2 syrk_uv16c_kernel (operands converted from nibble words, (tile, run) items), 3 ... walking a tile's fp32 runs
    # itself with half its sums carried in LDS (default), 3 with SNPGPU_UVC_PACE=0 (no pace-maker fetches) -- with clock / power, then their
    # L2 <-> fabric counters (separate --pmc passes) and busy cycles / instruction counts
    {
        echo "# configs[2], 8 steps + 2 warm-up per run, interleaved on one box"
        for rep in 1 2; do for v in 1 2 3 30; do
            SNPGPU_UVC_PACE=$([ $v = 30 ] && echo 0 || echo 1) SNPGPU_SYRK_UV16=${v:0:1} python bench.py --no-sub-results --no-cpu-baseline --no-pmc --no-probe --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']; c = d['config']
print('SNPGPU_SYRK_UV16=${v:0:1}$([ $v = 30 ] && echo " SNPGPU_UVC_PACE=0") %-18s value %.4g  ms_per_step %.2f  kernel_ms %.2f  sclk_mhz_median %s  power_w_median %s' % (r['kernel'], d['value'], d['ms_per_step'], r['ms_per_launch'], c.get('sclk_mhz_median'), c.get('power_w_median')))"
        done; done
        echo "# per feed block of 65536 SNPs (KiB counters x 1024 x launches per block; FETCH_SIZE raw, the guide's x 2 not applied), 2 steps + 1 warm-up"
        for v in 1 2 3 30; do for c in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCP_TCC_READ_REQ_sum"; do
            name=uvc_${v}_$(echo $c | cut -d" " -f1)
            ( cd /tmp && SNPGPU_UVC_PACE=$([ $v = 30 ] && echo 0 || echo 1) SNPGPU_SYRK_UV16=${v:0:1} rocprofv3 --kernel-trace --pmc $c -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --workload grm --steps 2 --warmup 1 --no-sub-results --no-cpu-baseline --no-pmc --no-probe --no-telemetry > "$OUT/$name.log" 2>&1 )
            python tools/pmc_summary.py "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log"
            python - "$OUT/$name.json" "${v:0:1}$([ $v = 30 ] && echo " SNPGPU_UVC_PACE=0")" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
for k, cs in d.items():
    if "syrk_uv16" in k:
        print("SNPGPU_SYRK_UV16=%s %-18s" % (sys.argv[2], k.split("(")[0]), "  ".join("%s %.6g x %d launches" % (c, x["mean"] * (1024 if c.endswith("_SIZE") else 1), x["launches"]) for c, x in sorted(cs.items())))
PY
        done; done
        echo "# super-tile edge of the work list (SNPGPU_H3_SUPER, default 8: 4 x 4 tiles for these kernels), forms 1 and 3: step time and FETCH_SIZE (raw) per block"
        for sup in 8 16; do for v in 1 3; do
            SNPGPU_H3_SUPER=$sup SNPGPU_SYRK_UV16=$v python bench.py --no-sub-results --no-cpu-baseline --no-pmc --no-probe --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('SNPGPU_H3_SUPER=$sup SNPGPU_SYRK_UV16=$v  ms_per_step %.2f  kernel_ms %.2f' % (d['ms_per_step'], r['ms_per_launch']))"
            name=uvc_sup${sup}_${v}
            ( cd /tmp && SNPGPU_H3_SUPER=$sup SNPGPU_SYRK_UV16=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/$name" -o "$name" -- python "$REPO/bench.py" --workload grm --steps 2 --warmup 1 --no-sub-results --no-cpu-baseline --no-pmc --no-probe --no-telemetry > "$OUT/$name.log" 2>&1 )
            python tools/pmc_summary.py "$OUT/$name/${name}_results.db" > "$OUT/$name.json" 2>> "$OUT/$name.log"
            python - "$OUT/$name.json" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
for k, cs in d.items():
    if "syrk_uv16" in k:
        print("    %-18s" % k.split("(")[0], "  ".join("%s %.6g" % (c, x["mean"] * 1024) for c, x in sorted(cs.items())))
PY
        done; done
    } > "$OUT/uvc_ab.txt" 2>&1
    ( cd tools/ubench && [ -x r06_cvt_check ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w r06_cvt_check.hip -o r06_cvt_check )
    { echo "# what v_cvt_scalef32_pk_f16_fp4 computes (tools/ubench/r06_cvt_check.hip)"; tools/ubench/r06_cvt_check; } >> "$OUT/uvc_ab.txt" 2>&1
    cat "$OUT/uvc_ab.txt"
fi
find "$OUT" -name "*.db" -delete
find "$OUT" -type d -empty -delete
ls "$OUT" | head -80
[ -f "$OUT/kernel_trace.txt" ] && head -40 "$OUT/kernel_trace.txt"
[ -f "$OUT/bench_default.json" ] && python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("bench value %.4g ms/step %.2f frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic")))
for k,v in d.get("summary",{}).items(): print("  ",k, v)
PY
exit 0
