#!/bin/bash
# Round 6: syrk_uv_kernel (32x32x16) against syrk_uv16_kernel (16x16x32) on one box, interleaved -> gpurun_out/r06_uv16_ab.txt
out=gpurun_out/r06_uv16_ab.txt
mkdir -p gpurun_out
echo "# source stamp $(python bench.py --stamp); configs[2], 8 steps + 2 warm-up per run, interleaved" > $out
for rep in 1 2; do
  for v in 0 1; do
    line=$(env SNPGPU_SYRK_UV16=$v python bench.py --no-sub-results --no-cpu-baseline --no-pmc --no-probe "$@" 2>/dev/null | tail -1)
    echo "SNPGPU_SYRK_UV16=$v $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=d["config"]; print("value %.4g ms_per_step %.2f kernel_ms_per_step %.2f frac %.3f sclk %s MHz power %s W" % (d["value"], d["ms_per_step"], r["ms_per_launch"], r["frac"], c.get("sclk_mhz_median"), c.get("power_w_median")))')" | tee -a $out
  done
done
