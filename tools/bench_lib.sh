#!/bin/bash
# A/B of two builds of libsnpgpu.so on one box:  tools/bench_lib.sh "<bench args>" libA.so libB.so ...  (each twice, interleaved)
args=$1; shift
for rep in 1 2; do for l in "$@"; do
    SNPGPU_LIB=$PWD/snprelate_amd/$l python bench.py --no-cpu-baseline $args 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$l', '| ms/step %.3f | ms/launch %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"
done; done
