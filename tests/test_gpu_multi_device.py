"""snpgpu_multi: one host process driving several GPUs through the C ABI (what the R shim binds for configs[3] / [4] and the
north_star job).  The test box has ONE GPU: a device ordinal listed several times gives several "devices" on it -- the
plan, the forwarding copies, the gathers and the eigen solver's broadcast / reduce run exactly as on distinct devices
(peer copies; the RCCL form is exercised with a one-device communicator)."""
import numpy as np
import pytest

import oracle as orc
from conftest import synth_geno

pytestmark = pytest.mark.gpu


def _feed_all(m, g, blk, packed=False):
    from snprelate_amd.gds import pack_2bit_rows
    for i in range(0, g.shape[0], blk):
        m.feed(pack_2bit_rows(g[i:i + blk]) if packed else g[i:i + blk])


def _offdiag(got, ref):
    return np.nanmax(np.abs(got - ref) / (np.abs(ref) + np.median(np.abs(ref))))


@pytest.mark.parametrize("devices,ppd", [((0, 0), 2), ((0, 0, 0), 1), ((0,), 3)])
def test_multi_counters_bit_exact_and_grm(devices, ppd):
    from snprelate_amd import _lib
    n, L, blk = 1300, 2500, 1024
    g = synth_geno(n, L, missing=0.03, seed=41)
    with _lib.MultiAccumulator(_lib.IBS, n, devices=devices, panels_per_device=ppd, max_block_snps=blk) as m:
        assert 1 <= m.info()["n_panels"] <= len(devices) * ppd        # empty panels (256-row boundaries) are not created
        rows = sorted(m.panels())
        assert rows[0][0] == 0 and rows[-1][1] == n and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        _feed_all(m, g, blk, packed=True)
        i0, i1, i2 = m.ibs_num()
    ref = orc.ibs_count(g)
    assert np.array_equal(np.stack([i0, i1, i2], 1).astype(np.uint32), ref)
    with _lib.MultiAccumulator(_lib.KING_ROBUST, n, devices=devices, panels_per_device=ppd, max_block_snps=blk) as m:
        _feed_all(m, g, 1000)                                          # uint8 blocks, ragged
        assert np.array_equal(m.king_robust_counts(), orc.king_robust_count(g))
        r0, rk = orc.king_robust_final(orc.king_robust_count(g), n)
        a, b = m.king_robust()
        assert np.array_equal(a, r0, equal_nan=True) and np.array_equal(b, rk, equal_nan=True)
    with _lib.MultiAccumulator(_lib.GRM_GCTA, n, devices=devices, panels_per_device=ppd, max_block_snps=blk) as m:
        _feed_all(m, g, blk)
        assert m.counts()[0] == L
        got = m.grm_gcta()
    assert _offdiag(got, orc.grm_gcta(g)) < 1e-5


def test_multi_king_two_passes_fill_one_triangle():
    """configs[4]'s shape of run: the counters of all panels do not fit at once -- two passes over the SNP stream, each with
    its own resident panels, both gathering into the same packed outputs."""
    from snprelate_amd import _lib
    n, L, blk = 1100, 2000, 1024
    g = synth_geno(n, L, missing=0.05, seed=43)
    ibs0 = np.full(n * (n + 1) // 2, np.nan)
    kin = np.full(n * (n + 1) // 2, np.nan)
    seen = []
    for q in range(2):
        with _lib.MultiAccumulator(_lib.KING_ROBUST, n, devices=(0, 0), panels_per_device=2, n_passes=2, pass_index=q,
                                   max_block_snps=blk) as m:
            seen += m.panels()
            _feed_all(m, g, blk)
            m.king_robust(out=(ibs0, kin))
    seen.sort()
    assert 2 < len(seen) <= 8 and seen[0][0] == 0 and seen[-1][1] == n and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    r0, rk = orc.king_robust_final(orc.king_robust_count(g), n)
    assert np.array_equal(ibs0, r0, equal_nan=True) and np.array_equal(kin, rk, equal_nan=True)


@pytest.mark.parametrize("comm", ["peer", "rccl"])
def test_multi_pca_and_gcta_eigen(comm, monkeypatch):
    """snpgdsPCA's covariance + top-k eigenvectors and "GCTA GRM + eigenvectors from one accumulation" over several
    contexts through snpgpu_multi_topk_eigen.  comm = rccl: a one-device communicator (all this box offers) carries the
    broadcast / reduce through librccl."""
    import torch
    from snprelate_amd import _lib
    from test_gpu_api_golden import _structured_geno
    monkeypatch.setenv("SNPGPU_MULTI_COMM", comm)
    devices = (0,) if comm == "rccl" else (0, 0, 0)
    n, L, k, blk = 1200, 2400, 8, 1024
    g = _structured_geno(n, L, seed=9)
    cov = orc.pca_cov(g)
    tr_ref = orc.trace_normalize(cov, n)
    w_ref, v_ref = np.linalg.eigh(orc.tri_to_full(cov, n))
    w_ref, v_ref = w_ref[::-1][:k], v_ref[:, ::-1][:, :k]
    with _lib.MultiAccumulator(_lib.PCA_COV, n, devices=devices, panels_per_device=2, max_block_snps=blk) as m:
        assert m.info()["uses_rccl"] == (comm == "rccl")
        # the exchange path in use carries a known pattern: broadcast, per-device scaling, sum-reduction (loud on a wrong sum)
        st = m.status()
        assert st["selftest_comm"] == st["selftest_feed"] == st["selftest_gather"] == -1 and st["panels_per_device"] == 2
        assert m.comm_selftest() == (comm == "rccl")
        # round 6: the same call pushed a known 2-bit block through the feed-forward star and a known slab from every listed device
        # through the gather path, each verified on the receiving device
        st = m.status()
        assert st["selftest_comm"] == st["selftest_feed"] == st["selftest_gather"] == 1, st
        assert st["n_devices"] == len(devices) and st["n_distinct_devices"] == 1 and st["uses_rccl"] == int(comm == "rccl")
        assert st["peer_pairs"] == 0 and st["peer_pairs_enabled"] == 0          # one physical GPU: no pair of distinct devices
        # blocks resident on the first device, fed asynchronously
        from snprelate_amd.gds import pack_2bit_rows
        pk = torch.from_numpy(pack_2bit_rows(g)).cuda()
        for i in range(0, L, blk):
            m.feed_device(pk[i:i + blk].data_ptr(), min(blk, L - i))
        m.sync()
        got, tr = m.pca_cov()
        assert abs(tr - tr_ref) < 1e-6 * tr_ref and _offdiag(got, cov) < 1e-5
        w, v, info = m.topk_eigen(k)
    np.testing.assert_allclose(w, w_ref, rtol=2e-5)
    assert np.all(np.abs(np.sum(v[:, :2] * v_ref[:, :2], axis=0)) > 1 - 1e-6) and info["max_rel_residual"] < 1e-8
    ref = orc.grm_gcta(g)
    wg = np.linalg.eigvalsh(orc.tri_to_full(ref, n))[::-1][:k]
    with _lib.MultiAccumulator(_lib.GRM_GCTA, n, devices=devices, panels_per_device=2, max_block_snps=blk) as m:
        _feed_all(m, g, blk)
        m.finalize_inplace()
        out = torch.empty(n * (n + 1) // 2, dtype=torch.float64, device="cuda")
        m.grm_gcta(out_ptr=out.data_ptr())                             # gathered into device memory of the first device
        torch.cuda.synchronize()
        assert _offdiag(out.cpu().numpy(), ref) < 1e-5
        w, v, info = m.topk_eigen(k, scale=1.0)
    np.testing.assert_allclose(w, wg, rtol=2e-5)
    assert info["max_rel_residual"] < 1e-8


@pytest.mark.parametrize("n,nd,ppd,passes", [(5000, 2, 2, 1), (9000, 3, 1, 2), (20000, 4, 2, 2)])
def test_multi_plan_is_the_python_plan(n, nd, ppd, passes):
    """The C++ panel plan of snpgpu_multi (plan_rows / plan_owners) is the twin of snprelate_amd/dist.py (panel_rows / pass_plan):
    same row boundaries, same owner of every panel in every pass -- the one-process and the one-process-per-GPU deployments cut
    the triangle identically."""
    from snprelate_amd import _lib
    from snprelate_amd.dist import pass_plan
    bounds, owned, _ = pass_plan(n, nd, ppd, passes, 8.0)
    for q in range(passes):
        with _lib.MultiAccumulator(_lib.IBS, n, devices=(0,) * nd, panels_per_device=ppd, n_passes=passes, pass_index=q,
                                   max_block_snps=256) as m:
            got = sorted((r0, r1) for r0, r1, _ in m.panels())
        want = sorted((bounds[p], bounds[p + 1]) for d in range(nd) for p in owned[q][d] if bounds[p + 1] > bounds[p])
        assert got == want


def test_multi_north_star_topology_eight_devices_two_panels_each():
    """The north_star job's shape at a size the oracle reaches: 8 "devices" x 2 panels (the plan of an 8-GPU node, all on the one
    GPU of the box), GCTA GRM with missing calls gathered + top-8 eigenvectors from the same accumulation."""
    from snprelate_amd import _lib
    from test_gpu_api_golden import _structured_geno
    n, L, k, blk = 4500, 3000, 8, 1024
    g = _structured_geno(n, L, seed=29)
    ref = orc.grm_gcta(g)
    wr = np.linalg.eigvalsh(orc.tri_to_full(ref, n))[::-1][:k]
    with _lib.MultiAccumulator(_lib.GRM_GCTA, n, devices=(0,) * 8, panels_per_device=2, max_block_snps=blk) as m:
        rows = sorted((r0, r1) for r0, r1, _ in m.panels())
        assert len(rows) >= 12 and rows[0][0] == 0 and rows[-1][1] == n and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        _feed_all(m, g, blk, packed=True)
        m.finalize_inplace()
        got = m.grm_gcta()
        w, v, info = m.topk_eigen(k, scale=1.0)
    assert _offdiag(got, ref) < 1e-5
    np.testing.assert_allclose(w, wr, rtol=2e-5)
    full = orc.tri_to_full(got, n)
    res = np.linalg.norm(full @ v - v * w, axis=0) / np.abs(w)
    assert res.max() < 1e-8 and info["max_rel_residual"] < 1e-8


@pytest.mark.parametrize("mem", ["host", "device"])
def test_multi_feed_packs_byte_genotypes_before_forwarding(mem):
    """snpgpu_multi_feed with byte genotypes (the format the kept GDS reader delivers) and more than one device: the first device
    packs the block to 2-bit rows before the star forwards it (a quarter of the xGMI bytes) and every panel is fed PACKED2 --
    values above 2 are missing calls, as CGenoReadBySNP clamps them; counters bit for bit, ragged last byte (n % 4 != 0)."""
    import torch
    from snprelate_amd import _lib
    n, L, blk = 1303, 2100, 1000
    g = synth_geno(n, L, missing=0.04, seed=5)
    g[17, 5] = 200                                    # > 3: a missing call
    ref = orc.ibs_count(np.minimum(g, 3))
    with _lib.MultiAccumulator(_lib.IBS, n, devices=(0, 0, 0), panels_per_device=1, max_block_snps=1024) as m:
        if mem == "host":
            for i in range(0, L, blk):
                m.feed(g[i:i + blk])
        else:
            gd = torch.from_numpy(g).cuda()
            for i in range(0, L, blk):
                m.feed_device(gd[i:i + blk].data_ptr(), min(blk, L - i), fmt=_lib.GENO_U8)
        m.sync()
        i0, i1, i2 = m.ibs_num()
    assert np.array_equal(np.stack([i0, i1, i2], 1).astype(np.uint32), ref)


def test_northstar_one_command_on_eight_listed_devices():
    """The north_star record as ONE command (tools/northstar_rehearsal.py --mode whole): eight listed devices (the one test GPU, eight
    times), panels per device chosen by the library, exchange self-test first, every block, GRM finalised in place, top-32 eigenpairs,
    gather, and SURVEY 8(d)'s sampled-tile parity (64 x 64 pairs x all SNPs in fp64) in the same run -- at N = 20 000 x 131 072 SNPs."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "northstar_rehearsal.py"), "--mode", "whole", "--n", "20000",
                        "--snps", "131072", "--block", "65536", "--devices", "0,0,0,0,0,0,0,0", "--missing", "0.01"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["devices"] == [0] * 8 and d["panels_per_device"] >= 1 and len(d["panels"]) >= 8
    rows = sorted(d["panels"])
    assert rows[0][0] == 0 and rows[-1][1] == 20000 and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    assert d["parity"]["pairs"] > 2000 and d["parity"]["max_rel_1e-5_contract"] < 1e-5 and d["parity"]["max_offdiag_figure"] < 1e-5, d["parity"]
    assert d["eigen_info"]["max_rel_residual"] < 1e-6 and d["gather_s"] is not None and d["accumulate_s"] > 0
    # round 6: the record says what the object found out about its devices before anything was accumulated
    assert d["selftest_comm"] == d["selftest_feed"] == d["selftest_gather"] == 1 and d["peer_access"] == {"pairs": 0, "enabled": 0}
    assert d["distinct_devices"] == 1 and len(d["device_pci"]) == 1


def test_gather_concurrent_per_device_equals_serial(monkeypatch):
    """Round 6: the gathers finalise and ship every device's panels concurrently (one host thread per listed device, asynchronous
    copies on the device's copy stream).  Same bytes as the serial loop, to host memory and to memory of the first device, for
    one-output and three-output kinds."""
    import torch
    from snprelate_amd import _lib
    n, L, blk = 2100, 2048, 1024
    g = synth_geno(n, L, missing=0.02, seed=53)
    devices = (0, 0, 0, 0)

    def run(kind):
        with _lib.MultiAccumulator(kind, n, devices=devices, panels_per_device=2, max_block_snps=blk) as m:
            _feed_all(m, g, blk, packed=True)
            if kind == _lib.IBS:
                host = np.stack(m.ibs_num(), 1)
                return host, None
            host = m.grm_gcta()
            dev = torch.empty(_lib.tri_size(n), dtype=torch.float64, device="cuda")
            m.grm_gcta(out_ptr=dev.data_ptr())
            return host, dev.cpu().numpy()

    for kind in (_lib.GRM_GCTA, _lib.IBS):
        monkeypatch.delenv("SNPGPU_MULTI_GATHER_SERIAL", raising=False)
        a_host, a_dev = run(kind)
        monkeypatch.setenv("SNPGPU_MULTI_GATHER_SERIAL", "1")
        b_host, b_dev = run(kind)
        assert np.array_equal(a_host, b_host, equal_nan=True)
        if a_dev is not None:
            assert np.array_equal(a_dev, b_dev, equal_nan=True) and np.array_equal(a_dev, a_host, equal_nan=True)
    assert np.array_equal(a_host.astype(np.uint32), orc.ibs_count(g))


def test_panels_per_device_chosen_by_the_library_and_sampled_entries():
    """panels_per_device = -1: the fewest panels per device whose accumulators and per-panel scratch fit the free memory (one, at
    this size); snpgpu_panel_entries returns the finalised entries wherever their panels live."""
    from snprelate_amd import _lib
    n, L, blk = 1500, 3000, 1024
    g = synth_geno(n, L, missing=0.02, seed=47)
    with _lib.MultiAccumulator(_lib.GRM_GCTA, n, devices=(0, 0, 0), panels_per_device=-1, max_block_snps=blk) as m:
        assert m.info()["n_panels"] <= 3
        _feed_all(m, g, blk)
        m.finalize_inplace()
        rng = np.random.default_rng(5)
        i = rng.integers(0, n, 500)
        j = rng.integers(0, n, 500)
        rows, cols = np.minimum(i, j), np.maximum(i, j)
        got = m.entries(rows, cols)
        full = m.grm_gcta()
    assert np.array_equal(got, full[cols + rows * (2 * n - rows - 1) // 2])
    with pytest.raises(_lib.SnpGpuError, match="upper trapezoid"):
        with _lib.Accumulator(_lib.PCA_COV, n, max_block_snps=blk) as a:
            a.feed(g[:blk])
            a.panel_entries([5], [3])
