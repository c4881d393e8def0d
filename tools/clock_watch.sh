#!/bin/bash
# Shader clock and socket power while each bench workload runs (GPU box):  tools/clock_watch.sh <tag>
#   -> gpurun_out/<tag>/clocks_<workload>.txt  (one rocm-smi sample per line) + a min/median/max summary
set -u
TAG=${1:-clk}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
for w in grm ibs king idle; do
    case $w in
        grm) a="--steps 24 --warmup 2";;
        ibs) a="--steps 4000 --warmup 20";;
        king) a="--steps 2500 --warmup 20";;
        idle) a="";;
    esac
    if [ $w != idle ]; then
        python bench.py --no-cpu-baseline --no-sub-results --workload $w $a > "$OUT/bench_$w.json" 2> /dev/null &
        pid=$!
    else
        sleep 3 & pid=$!
    fi
    : > "$OUT/clocks_$w.txt"
    while kill -0 $pid 2> /dev/null; do
        rocm-smi --showclocks --showpower --csv 2> /dev/null | tr '\n' ' ' >> "$OUT/clocks_$w.txt"
        echo >> "$OUT/clocks_$w.txt"
        sleep 0.2
    done
    wait $pid
done
python - "$OUT" <<'PY'
import re, sys, glob, statistics as st
for f in sorted(glob.glob(sys.argv[1] + "/clocks_*.txt")):
    sclk, pw = [], []
    for line in open(f):
        m = re.findall(r"\((\d+)Mhz\)", line)
        nums = re.findall(r"[-+]?\d+\.\d+", line)
        hdr = line.split(" ")[0].split(",") if line.strip() else []
        if hdr and len(line.split(" ")) > 1:
            vals = line.split(" ")[1].split(",")
            d = dict(zip(hdr, vals))
            for k, v in d.items():
                if "sclk" in k.lower():
                    mm = re.search(r"(\d+)Mhz", v)
                    if mm: sclk.append(int(mm.group(1)))
                if "power" in k.lower():
                    try: pw.append(float(v))
                    except ValueError: pass
    def s(x): return "n=%d min %.0f med %.0f max %.0f" % (len(x), min(x), st.median(x), max(x)) if x else "none"
    print(f.split("/")[-1], "sclk MHz:", s(sclk), "| power W:", s(pw))
PY
head -c 600 "$OUT/clocks_grm.txt"
