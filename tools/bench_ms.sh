#!/bin/bash
# ms per step / per dominant-kernel launch of bench.py runs (GPU box):  tools/bench_ms.sh "<bench args>" ["<bench args>" ...]
for a in "$@"; do
    python bench.py --no-cpu-baseline $a 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$a', '| ms/step %.3f | ms/launch %.3f | value %.4g | frac %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['value'], d['roofline']['frac']))"
done
