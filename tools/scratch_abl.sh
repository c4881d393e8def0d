run() { SNPGPU_PAIR_FP4_16=$4 SNPGPU_LIB=$PWD/snprelate_amd/$1 python bench.py --workload $2 $3 --no-cpu-baseline --no-sub-results --no-pmc --no-probe --steps 150 --warmup 40 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$1 $2 $3 fp4_16=$4', '| value %.4g | ms/step %.3f | kernel ms %.3f | sclk %s power %s' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], c.get('sclk_mhz_median'), c.get('power_w_median')))"; }
for rep in 1 2; do
for l in libsnpgpu_head.so libsnpgpu.so; do
run $l ibs "" 0; run $l ibs "--missing 0.02" 0; run $l king "" 0; run $l king_homo "" 0
done; done
run libsnpgpu.so ibs "" 1
