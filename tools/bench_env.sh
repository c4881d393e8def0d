#!/bin/bash
# A/B of environment switches on one box:  tools/bench_env.sh "<bench args>" "VAR=a" "VAR=b" ...   (each env spec run twice, interleaved)
args=$1; shift
for rep in 1 2; do for e in "$@"; do
    env $e python bench.py --no-cpu-baseline $args 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$e', '| ms/step %.3f | ms/launch %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch']))"
done; done
