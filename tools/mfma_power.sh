#!/bin/bash
# Sustained MFMA rate, shader clock and socket power of a register-only MFMA stream (tools/ubench/mfma_power_ubench)
# for f16 / i8 with realistic and with zero operands:  tools/mfma_power.sh <tag>  -> gpurun_out/<tag>/mfma_power.txt
set -u
TAG=${1:-pw}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
[ -x tools/ubench/mfma_power_ubench ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w -o tools/ubench/mfma_power_ubench tools/ubench/mfma_power_ubench.hip
for dt in f16 i8; do for data in real exact zero; do
    tools/ubench/mfma_power_ubench $dt $data 5 > "$OUT/run.txt" &
    pid=$!
    : > "$OUT/smi.txt"
    while kill -0 $pid 2> /dev/null; do
        rocm-smi --showclocks --showpower --csv 2> /dev/null | grep card0 >> "$OUT/smi.txt"
        sleep 0.2
    done
    wait $pid
    python - "$OUT" <<'PY' >> "$OUT/mfma_power.txt"
import re, sys, statistics as st
sclk, pw = [], []
for line in open(sys.argv[1] + "/smi.txt"):
    v = line.strip().split(",")
    m = re.search(r"(\d+)Mhz", v[5])
    if m and int(m.group(1)) > 1000:
        sclk.append(int(m.group(1))); pw.append(float(v[9]))
sclk, pw = sclk[3:-1], pw[3:-1]       # drop ramp-up / tail samples
print(open(sys.argv[1] + "/run.txt").read().strip(), "| sclk MHz median %.0f (min %.0f max %.0f) | power W median %.0f max %.0f"
      % (st.median(sclk), min(sclk), max(sclk), st.median(pw), max(pw)))
PY
done; done
rm -f "$OUT/run.txt" "$OUT/smi.txt"
cat "$OUT/mfma_power.txt"
