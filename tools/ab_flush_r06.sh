#!/bin/bash
# Round 6 flush A/B on one box, interleaved: fused (tile, run) launch with fp64 atomics (default) / one launch per run with atomics /
# one launch per run with the non-atomic read-modify-write flush of exclusively owned tiles (SNPGPU_FLUSH_RMW=1)
#   tools/ab_flush_r06.sh  -> gpurun_out/r06_flush_ab.txt
out=gpurun_out/r06_flush_ab.txt
mkdir -p gpurun_out
echo "# source stamp $(python bench.py --stamp); configs[2], 8 steps + 2 warm-up per run, three variants interleaved, two rounds" > $out
for rep in 1 2; do
  for v in "fused_atomics:" "per_run_atomics:SNPGPU_RUN_INNER=0" "per_run_rmw:SNPGPU_RUN_INNER=0 SNPGPU_FLUSH_RMW=1"; do
    name=${v%%:*}; envs=${v#*:}
    line=$(env $envs python bench.py --no-sub-results --no-cpu-baseline --no-pmc --no-probe "$@" 2>/dev/null | tail -1)
    echo "$name [$envs] $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print("value %.4g ms_per_step %.2f kernel_ms_per_step %.2f (%d launches) frac %.3f sclk %s MHz power %s W" % (d["value"], d["ms_per_step"], r["ms_per_launch"] * r["launches"] / d["steps"], r["launches"], r["frac"], c.get("sclk_mhz_median"), c.get("power_w_median")))')" | tee -a $out
  done
done
