#!/bin/bash
python -m pytest tests -m gpu -q -k "2bit_rows or ibs or king or IBS or KING or config4 or shim" 2>&1 | tail -3
bash tools/bench_env.sh "--no-sub-results --workload ibs --steps 50 --warmup 20" "SNPGPU_PREP_TWO_PASS=1" "X=1"
bash tools/bench_env.sh "--no-sub-results --workload king --steps 50 --warmup 20" "SNPGPU_PREP_TWO_PASS=1" "X=1"
