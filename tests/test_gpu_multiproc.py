"""Two ranks (one process each) through the multi-GPU drivers.  A single test GPU is shared by
both ranks, so the collectives run on the gloo backend here; on an 8-GPU node the same code
runs with backend "nccl" (RCCL) -- bench.py does."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd import multigpu
dist.init_process_group("gloo")
rank = dist.get_rank()
n, L = 1300, 2100
g = synth_geno(n, L, missing=0.03, seed=17)
blocks = lambda: (g[i:i + 1024] for i in range(0, L, 1024))
grm = multigpu.grm_distributed(blocks(), n, method="GCTA", max_block_snps=1024)
ibs0, kin = multigpu.king_distributed(blocks(), n, max_block_snps=1024)
pca = multigpu.pca_distributed(blocks(), n, eigen_cnt=8, max_block_snps=1024)
# configs[4] shape of run: the counters do not fit at once -> two passes over the SNP stream, two panels per rank each
ibs0_2, kin_2 = multigpu.king_distributed(blocks, n, max_block_snps=1024, passes=2, panels_per_rank=2)
# configs[3] shape of run: nobody may hold the triangle -> slabs go to a sink (files), or stay on their ranks
sink = multigpu.FileSlabSink(%(sink)r, rank=rank)
multigpu.grm_distributed(blocks(), n, method="Eigenstrat", max_block_snps=1024, panels_per_rank=2, sink=sink)
mine = multigpu.grm_distributed(blocks(), n, method="GCTA", max_block_snps=1024, gather=False)
# per-SNP statistics computed once per node (each rank its share of a block's SNPs, all-gathered) instead of once per rank:
# integer statistics, so the result must equal the per-rank form bit for bit
grm_shared = multigpu.grm_distributed(blocks(), n, method="GCTA", max_block_snps=1024, shared_stats=True)
# ... and a data set so small that rank 0's panel is EMPTY (256-row boundaries): it has no context to scan with, and still has to
# join every all-gather of the statistics (ADVICE r05: it used to skip the collective and rank 1 hung)
from snprelate_amd.dist import panel_rows as _pr
assert _pr(300, 2)[1] == 0
g_small = synth_geno(300, 1500, missing=0.03, seed=19)
grm_small = multigpu.grm_distributed((g_small[i:i + 1024] for i in range(0, 1500, 1024)), 300, method="GCTA", max_block_snps=1024, shared_stats=True)
# device-resident 2-bit blocks through the shared-statistics path (what bench.py feeds)
from snprelate_amd.gds import pack_2bit_rows
pk = [torch.from_numpy(pack_2bit_rows(g[i:i + 1024])).cuda() for i in range(0, L, 1024)]
grm_shared_dev = multigpu.grm_distributed(((t.data_ptr(), t.shape[0]) for t in pk), n, method="GCTA", max_block_snps=1024, shared_stats=True)
sink2 = multigpu.FileSlabSink(%(sink)r + "_king", rank=rank)
multigpu.king_distributed(blocks, n, max_block_snps=1024, mem_budget=20 * 700 * 1300, sink=sink2)
dist.barrier()
if rank == 0:
    ref = orc.grm_gcta(g)
    err = np.nanmax(np.abs(grm.cpu().numpy() - ref) / (np.abs(ref) + np.median(np.abs(ref))))
    assert err < 1e-5, err
    r0, rk = orc.king_robust_final(orc.king_robust_count(g), n)
    assert np.array_equal(ibs0.cpu().numpy(), r0, equal_nan=True)
    assert np.array_equal(kin.cpu().numpy(), rk, equal_nan=True)
    c = orc.pca_cov(g); tr = orc.trace_normalize(c, n)
    w = np.linalg.eigvalsh(orc.tri_to_full(c, n))[::-1][:8]
    np.testing.assert_allclose(pca["eigenval"].cpu().numpy(), w, rtol=2e-5)
    assert abs(pca["TraceXTX"] - tr) / tr < 1e-6
    assert np.array_equal(ibs0_2.cpu().numpy(), r0, equal_nan=True) and np.array_equal(kin_2.cpu().numpy(), rk, equal_nan=True)
    from snprelate_amd.dist import panel_rows, slab_range
    got = multigpu.read_file_slabs(%(sink)r, "grm", n)
    assert np.nanmax(np.abs(got - c) / (np.abs(c) + np.median(np.abs(c)))) < 1e-5 and not np.isnan(got).any()
    b = panel_rows(n, 2)
    for p_, slab in mine.items():
        lo, hi = slab_range(n, b[p_], b[p_ + 1])
        assert np.array_equal(slab.cpu().numpy(), grm.cpu().numpy()[lo:hi], equal_nan=True)
    assert np.array_equal(multigpu.read_file_slabs(%(sink)r + "_king", "kinship", n), rk, equal_nan=True)
    assert np.array_equal(multigpu.read_file_slabs(%(sink)r + "_king", "IBS0", n), r0, equal_nan=True)
    assert np.array_equal(grm_shared.cpu().numpy(), grm.cpu().numpy(), equal_nan=True)
    assert np.array_equal(grm_shared_dev.cpu().numpy(), grm.cpu().numpy(), equal_nan=True)
    ref_s = orc.grm_gcta(g_small)
    assert np.nanmax(np.abs(grm_small.cpu().numpy() - ref_s) / (np.abs(ref_s) + np.median(np.abs(ref_s)))) < 1e-5
    print("MULTI_OK", err)
dist.destroy_process_group()
"""


def test_two_rank_drivers(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT, "sink": str(tmp_path / "slabs")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert "MULTI_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_two_ranks_contract():
    """bench.py under torch.distributed.run with 2 ranks (gloo + one shared GPU here; the driver uses
    nccl with one GPU per rank): one JSON line from rank 0 with the contract's fields."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SNPGPU_BENCH_BACKEND="gloo", SNPGPU_BENCH_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "grm", "--samples", "6000",
                        "--block", "2048"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_bench_self_launches_two_ranks():
    """plain `python bench.py --gpus 2` -- no torch.distributed.run wrapper, no WORLD_SIZE -- must launch its own two
    ranks and say what the collective library saw (gloo + one shared GPU here; `nccl` = RCCL with one GPU per rank on a node)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SNPGPU_BENCH_BACKEND="gloo", SNPGPU_BENCH_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "grm",
                        "--samples", "6000", "--block", "2048"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["rccl_ranks"] == 2 and len(c["rank_devices"]) == 2 and c["self_launched"] is True
    assert c["distinct_devices"] == 1                      # both ranks on the one test GPU, and the record says so
    assert all(x.startswith("0@") and "gfx950" in x for x in c["rank_devices"])
    assert isinstance(c["gather_ms"], float) and c["gather_ms"] > 0          # the final gather is on by default for N > 1
    assert "gloo" in c["collective_backend"]
    assert "sclk_mhz_median" in c and "power_w_median" in c and len(c.get("rank_sclk_mhz_median", [])) == 2


def test_bench_single_rank_contract():
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--workload", "ibs", "--samples", "4096", "--block", "2048"], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "cpu_baseline" in d and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline"]["launches"] == 2
    # round 6: shader clock / socket power sampled in the timed region (null + a reason when the SMI library gave nothing)
    c = d["config"]
    assert "sclk_mhz_median" in c and "power_w_median" in c and ("telemetry_error" in c)


def test_bench_probe_and_telemetry_in_the_record():
    """roofline.sustained_peak_measured is a measurement of THIS box in THIS run (register-only MFMA stream through
    snpgpu_diag_mfma_rate), and the record carries shader clock and socket power of the timed region."""
    import json
    env = dict(os.environ, SNPGPU_BENCH_PROBE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "grm", "--samples", "16384",
                        "--block", "16384", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    pr = d["config"]["sustained_probe"]
    for m in ("f16_uv_16x16x32", "f16_uv", "f16_exact_row", "fp4", "f16_zero"):
        assert isinstance(pr[m], list) and pr[m][0] > 0, pr
    assert 1000 < pr["f16_uv"][0] <= 2600 and pr["f16_zero"][0] >= pr["f16_uv"][0] * 0.98 and 3000 < pr["fp4"][0] <= 10500, pr
    assert 1000 < pr["f16_zero"][1] < 2500, pr                     # implied shader clock, MHz
    ro = d["roofline"]
    # the headline kernel runs on v_mfma_f32_16x16x32_f16 (round 6): its ceiling is that shape's sustained rate, above the 32 x 32 x 16 one
    assert ro["kernel"] == "syrk_uv16c_kernel" and ro["sustained_peak_measured"] == pr["f16_uv_16x16x32"][0]
    assert ro["sustained_peak_source"].startswith("measured in this run") and pr["f16_uv_16x16x32"][0] > pr["f16_uv"][0]
    c = d["config"]
    if c["telemetry_samples"]:                                     # an SMI source answered: the numbers must be plausible
        assert 90 <= c["sclk_mhz_median"] <= 2600 and 50 <= c["power_w_median"] <= 1600, c


def test_diag_mfma_rate_direct():
    from snprelate_amd import _lib
    r, mhz = _lib.diag_mfma_rate(_lib.DIAG_F16_ZERO, 0.5)
    assert 1500 < r < 2600 and 1400 < mhz < 2500, (r, mhz)
    assert ":" in _lib.device_pci(0)
    with pytest.raises(_lib.SnpGpuError):
        _lib.diag_mfma_rate(9, 0.5)


def test_bench_rccl_path_with_one_rank():
    """bench.py under torch.distributed.run with backend "nccl" (= RCCL) and ONE rank: all a one-GPU box can execute of the
    driver's multi-GPU launch -- communicator set-up on the device, barriers, the max-over-ranks all-reduce of the timings and
    the final slab gather go through RCCL."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SNPGPU_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29551", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "grm", "--samples", "6000",
                        "--block", "2048", "--gather", "--no-sub-results", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and isinstance(d["config"]["gather_ms"], float)


def test_bench_eight_ranks_on_one_gpu():
    """The driver's 8-GPU launch line at a size one GPU holds eight panels of: `bench.py --gpus 8` under torch.distributed.run with
    8 ranks sharing device 0 over gloo (RCCL needs one device per rank).  The contract line comes from rank 0, the eight panels
    cover the triangle exactly once, every rank reports a pair-kernel time, and the final gather of the slabs completes."""
    import json
    n = 20000
    # round 6: the plain command (bench.py launches its own ranks), no --gather flag (on by default)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SNPGPU_BENCH_BACKEND="gloo", SNPGPU_BENCH_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", "grm", "--samples", str(n),
                        "--block", "4096"], capture_output=True, text=True, timeout=1500, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    c = d["config"]
    assert c["rccl_ranks"] == 8 and len(c["rank_devices"]) == 8 and c["self_launched"] is True
    assert c["shared_stats"] is True           # from 4 ranks: per-SNP statistics once per node (one all-gather per block, no host sync)
    assert len(c["rank_pairs"]) == 8 and sum(c["rank_pairs"]) == n * (n + 1) / 2 and min(c["rank_pairs"]) > 0
    assert len(c["rank_kernel_ms_per_step"]) == 8 and min(c["rank_kernel_ms_per_step"]) > 0
    assert isinstance(c["gather_ms"], float) and np.isfinite(c["gather_ms"]) and c["gather_ms"] > 0
    from snprelate_amd.dist import panel_cost, panel_rows
    b = panel_rows(n, 8)
    cost = [panel_cost(n, b[i], b[i + 1], alpha=min(512.0, n / 32.0)) for i in range(8)]
    assert max(cost) / (sum(cost) / 8) < 1.15          # 256-row boundaries at this small n
