import sys, time
sys.path.insert(0, "/root/repo")
from snprelate_amd import api
f = api.snpgdsOpen("/root/repo/tests/golden/hapmap_geno.gds")
for rep in range(4):
    t0 = time.perf_counter(); r = api.snpgdsGRM(f, method="GCTA", verbose=False); t1 = time.perf_counter()
    t2 = time.perf_counter(); i = api.snpgdsIBS(f, verbose=False); t3 = time.perf_counter()
    t4 = time.perf_counter(); p = api.snpgdsPCA(f, verbose=False); t5 = time.perf_counter()
    t6 = time.perf_counter(); k = api.snpgdsIBDKING(f, verbose=False); t7 = time.perf_counter()
    print("rep %d  snpgdsGRM %.1f ms  snpgdsIBS %.1f ms  snpgdsPCA %.1f ms  snpgdsIBDKING %.1f ms  (279 x %d SNPs)" % (rep, (t1-t0)*1e3, (t3-t2)*1e3, (t5-t4)*1e3, (t7-t6)*1e3, len(r["snp_id"])))
