import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from snprelate_amd import _lib
from snprelate_amd.eigen import PanelOperator, _orth
N, b = 50000, 40
g = np.random.default_rng(0).integers(0, 3, size=(4096, N), dtype=np.uint8)
acc = _lib.Accumulator(_lib.PCA_COV, N, max_block_snps=4096)
acc.feed(g)
dev = torch.device("cuda", 0)
op = PanelOperator([acc], N, dev)
q = _orth(torch.randn(b, N, dtype=torch.float64, device=dev))
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("matmul (b=%d): %.2f ms" % (b, timed(lambda: op.matmul(q))))
print("_orth (n x b QR): %.2f ms" % timed(lambda: _orth(q)))
basis = torch.cat([q] * 6, 0)
print("project out basis (240 x n) twice: %.2f ms" % timed(lambda: (q - (q @ basis.T) @ basis)))
print("eigh 480: %.2f ms" % timed(lambda: torch.linalg.eigh(torch.randn(480, 480, dtype=torch.float64, device=dev))))
