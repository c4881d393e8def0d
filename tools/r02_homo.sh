#!/bin/bash
python -m pytest tests -m gpu -q -k "homo or KING or king or ibs or IBS or beta or Beta" 2>&1 | tail -3
python - <<'PY'
import time, torch, numpy as np
from snprelate_amd import _lib
n, B = 10000, 16384
t = torch.empty((B, n // 4), dtype=torch.uint8, device="cuda")
_lib.synth_block(t.data_ptr(), n, 0, B, 1, missing=0.05)
for kind, name in ((_lib.KING_HOMO, "king_homo"), (_lib.INDIV_BETA, "beta")):
    a = _lib.Accumulator(kind, n, max_block_snps=B)
    for _ in range(5): a.feed_device(t.data_ptr(), B)
    a.sync(); t0 = time.perf_counter()
    for _ in range(20): a.feed_device(t.data_ptr(), B)
    a.sync(); dt = (time.perf_counter() - t0) / 20
    print(name, "ms/block %.3f  pair-genotypes/s %.3g" % (dt * 1e3, n * n / 2 * B / dt))
    a.close()
PY
