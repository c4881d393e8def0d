#!/bin/bash
# single-product SYRK: super-tile edge and fp32 run / feed block length
bash tools/bench_env.sh "--no-sub-results --steps 8 --warmup 2" "SNPGPU_X1_SUPER=2" "SNPGPU_X1_SUPER=4" "SNPGPU_X1_SUPER=8"
bash tools/bench_env.sh "--no-sub-results --steps 4 --warmup 1 --block 32768" "SNPGPU_H3_PROMOTE=16384" "SNPGPU_H3_PROMOTE=32768"
bash tools/bench_env.sh "--no-sub-results --steps 8 --warmup 2" "SNPGPU_H3_PROMOTE=8192"
