#!/bin/bash
# A/B of one environment switch on one box, interleaved (box-to-box noise is +-3 %):
#   tools/ab_env.sh VAR "A_value B_value" [bench.py args...]      -> gpurun_out/ab_VAR.txt
var=$1; vals=$2; shift 2
out=gpurun_out/ab_${var}.txt
: > $out
for rep in 1 2; do
  for v in $vals; do
    line=$(env $var=$v python bench.py --no-sub-results --no-cpu-baseline --no-pmc "$@" 2>/dev/null | tail -1)
    echo "$var=$v args=$* $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("value %.4g ms_per_step %.2f kernel_ms %.2f x %d frac %.3f" % (d["value"], d["ms_per_step"], r["ms_per_launch"], r["launches"], r["frac"]))')" | tee -a $out
  done
done
