run() { SNPGPU_SYRK_UV16=$1 SNPGPU_LIB=$PWD/snprelate_amd/$2 python bench.py --no-cpu-baseline --no-sub-results --no-pmc --no-probe --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('uv16=$1 $2', '| ms/step %.2f | kernel ms %.2f | sclk %s power %s' % (d['ms_per_step'], d['roofline']['ms_per_launch'], c.get('sclk_mhz_median'), c.get('power_w_median')))"; }
run 0 libsnpgpu.so
for v in p3 p6 p9 p10 p11 p3; do run 1 libsnpgpu_$v.so; done
