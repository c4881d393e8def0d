"""CPU-side tests: the C ABI library loads and exports every symbol the header declares, the
product path fails loudly without a GPU, sharding logic, and the multi-rank gather (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    from snprelate_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "snpgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(snpgpu_\w+)\s*\(", hdr))
    declared -= {"snpgpu_ctx", "snpgpu_opts", "snpgpu_reduce_fn"}
    assert len(declared) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == declared
    assert _lib.lib().snpgpu_abi_version() == 2


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from snprelate_amd import _lib, api
    from snprelate_amd.gds import GenoFile
    with pytest.raises(_lib.SnpGpuError):
        _lib.Accumulator(_lib.IBS, 16)
    f = GenoFile(genotype=np.zeros((8, 4), np.uint8))
    with pytest.raises(_lib.SnpGpuError):
        api.snpgdsIBS(f, verbose=False)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "snprelate_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "snp_oracle" not in txt, fn


def test_pack_roundtrip_and_gds_subset(hapmap):
    from snprelate_amd.gds import pack_2bit_rows, unpack_2bit_rows
    rng = np.random.default_rng(0)
    for n in (1, 3, 4, 5, 279):
        g = rng.integers(0, 4, size=(17, n), dtype=np.uint8)
        assert np.array_equal(unpack_2bit_rows(pack_2bit_rows(g), n), g)
    g = hapmap.read_genotype(snp_sel=np.arange(10, 20), samp_sel=np.arange(5, 50))
    assert g.shape == (10, 45)
    assert np.array_equal(g, hapmap.read_genotype()[10:20, 5:50])


def test_panel_rows_properties():
    from snprelate_amd.dist import panel_rows, slab_range, tri_offset
    for n in (10, 1000, 4096, 100000, 500000):
        for world in (1, 2, 4, 8):
            b = panel_rows(n, world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == n
            assert all(b[i] <= b[i + 1] for i in range(world))
            assert all(x % 256 == 0 for x in b[1:-1])
            sizes = [slab_range(n, b[i], b[i + 1])[1] - slab_range(n, b[i], b[i + 1])[0] for i in range(world)]
            assert sum(sizes) == n * (n + 1) // 2 == tri_offset(n, n)
            if n >= 100000:
                # equal TIME: pairs + alpha * columns of the pre-pass (dist.panel_cost); the first panel is the smallest
                from snprelate_amd.dist import panel_cost
                cost = [panel_cost(n, b[i], b[i + 1], alpha=min(512.0, n / (4.0 * world))) for i in range(world)]
                assert max(cost) / (sum(cost) / world) < 1.02
                assert max(sizes) / (sum(sizes) / world) < (1.10 if n == 100000 else 1.03)
                # SNPGPU_PLAN_ALPHA=0: the equal-area plan
                b0 = panel_rows(n, world, alpha=0.0)
                s0 = [slab_range(n, b0[i], b0[i + 1])[1] - slab_range(n, b0[i], b0[i + 1])[0] for i in range(world)]
                assert max(s0) / (sum(s0) / world) < 1.02


def test_panel_plan_balances_memory():
    """Several panels per rank, each given largest-first to the least loaded rank, even out the rectangular
    accumulator storage: with one panel per rank the last GPU (a square holding a triangle) needs about
    twice the memory of the others."""
    from snprelate_amd.dist import panel_plan, panel_storage
    n, world = 500000, 8
    for k in (1, 2, 4):
        bounds, owned = panel_plan(n, world, k)
        assert sorted(p for o in owned for p in o) == list(range(world * k))
        per_rank = [sum(panel_storage(n, bounds[p], bounds[p + 1]) for p in o) for o in owned]
        ratio = max(per_rank) / (sum(per_rank) / world)
        assert ratio < {1: 1.75, 2: 1.40, 4: 1.20}[k], (k, ratio)
        # fp64 accumulators of the worst rank at N = 500 000 on 8 GPUs, GiB
        assert max(per_rank) * 8 / 2**30 < {1: 240, 2: 180, 4: 150}[k]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import oracle as orc
from oracle.synth import synth_geno
from snprelate_amd.dist import panel_rows, slab_range, gather_slabs, panel_plan, gather_plan
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 700
g = synth_geno(n, 200, missing=0.03, seed=4)
full = orc.grm_gcta(g)                       # each rank's "device result" is stood in by the oracle
b = panel_rows(n, world)
lo, hi = slab_range(n, b[rank], b[rank + 1])
out = gather_slabs(torch.from_numpy(full[lo:hi].copy()), n, b, rank, world)
bounds, owned = panel_plan(n, world, 2)       # two panels per rank
slabs = [torch.from_numpy(full[slice(*slab_range(n, bounds[p], bounds[p + 1]))].copy()) for p in owned[rank]]
out2 = gather_plan(slabs, n, bounds, owned, rank, world)
# shared per-SNP statistics: every rank fills its share of the block's SNPs, one all-gather makes the arrays whole on all ranks
from snprelate_amd.dist import snp_share, allgather_block_stats
for L in (200, 199, 3):
    gg = g[:L]
    valid = gg <= 2
    s_ref = (gg * valid).sum(1).astype(np.int32); c_ref = valid.sum(1).astype(np.int32)
    lo_s, hi_s = snp_share(L, rank, world)
    st = torch.zeros((2, L), dtype=torch.int32)
    st[0, lo_s:hi_s] = torch.from_numpy(s_ref[lo_s:hi_s]); st[1, lo_s:hi_s] = torch.from_numpy(c_ref[lo_s:hi_s])
    allgather_block_stats(st[0], st[1], L, rank, world)
    assert np.array_equal(st[0].numpy(), s_ref) and np.array_equal(st[1].numpy(), c_ref), (rank, L)
# a rank without panels (n small against the plan) joins the collective with an empty share: only rank 1 scans, both end up whole
from snprelate_amd.dist import stats_ranks
assert stats_ranks([0, 0, 256], [[0], [1]], 2) == [1] and stats_ranks([0, 256, 512], [[0], [1]], 2) == [0, 1]
for L in (200, 7):
    gg = g[:L]
    valid = gg <= 2
    s_ref = (gg * valid).sum(1).astype(np.int32); c_ref = valid.sum(1).astype(np.int32)
    st = torch.zeros((2, L), dtype=torch.int32)
    if rank == 1:
        st[0] = torch.from_numpy(s_ref); st[1] = torch.from_numpy(c_ref)
    allgather_block_stats(st[0], st[1], L, rank, world, active=[1])
    assert np.array_equal(st[0].numpy(), s_ref) and np.array_equal(st[1].numpy(), c_ref), (rank, L)
shares = [snp_share(201, r, 4) for r in range(4)]
assert shares[0][0] == 0 and shares[-1][1] == 201 and all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
if rank == 0:
    assert np.array_equal(out.numpy(), full, equal_nan=True)
    assert np.array_equal(out2.numpy(), full, equal_nan=True)
    print("GATHER_OK")
dist.destroy_process_group()
"""


def test_gather_slabs_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert "GATHER_OK" in r.stdout, r.stdout + r.stderr


def test_merge_grm_validation_is_host_side(tmp_path):
    """snpgdsMergeGRM (R/IBD.R:624-741): file checks, weights and the merged snp.id list are host logic and raise
    before any device work; the arithmetic itself needs the GPU (fails loudly here)."""
    from snprelate_amd import api, gds, _lib
    n = 5
    def mk(name, snps, cmd=("snpgdsGRM", ":method = GCTA"), fmt=True):
        fn = str(tmp_path / name)
        nodes = {"command": np.array(cmd), "sample.id": np.arange(n), "snp.id": np.asarray(snps), "grm": np.eye(n)}
        if fmt:
            gds.write_output(fn, nodes)
        else:
            with open(fn, "wb") as f:
                np.savez(f, **nodes)
        return fn
    a, b = mk("a.gds", [1, 2, 3]), mk("b.gds", [4, 5])
    back = gds.read_output(a)
    assert str(back["FileFormat"]) == "SNPRELATE_OUTPUT" and list(back["snp.id"]) == [1, 2, 3]
    with pytest.raises(ValueError, match="is not valid"):
        api.snpgdsMergeGRM([a, mk("c.gds", [6], fmt=False)], verbose=False)
    with pytest.raises(ValueError, match="different command"):
        api.snpgdsMergeGRM([a, mk("d.gds", [6], cmd=("snpgdsGRM", ":method = IndivBeta"))], verbose=False)
    with pytest.raises(ValueError, match="created by snpgdsGRM"):
        api.snpgdsMergeGRM([mk("e.gds", [6], cmd=("other", "x"))], verbose=False)
    with pytest.raises(ValueError, match="length\\(weight\\)"):
        api.snpgdsMergeGRM([a, b], weight=[1.0], verbose=False)
    with pytest.raises(ValueError, match="non-empty"):
        api.snpgdsMergeGRM([], verbose=False)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(_lib.SnpGpuError):
            api.snpgdsMergeGRM([a, b], verbose=False)


def test_synth_hash_generator_known_answers():
    """The counter-based generator's numpy twin (oracle/synth.py) is pinned by value: the GPU kernel
    (snpgpu_synth_block) is compared with it bit for bit in tests/test_gpu_parity.py."""
    from oracle.synth import synth_hash_block_packed, synth_hash_geno, synth_hash_threshold
    g = synth_hash_geno([0, 1, 2, 3, 99999, 499999], 5, 3, 20240601, 0.05, 0, False)
    assert g.tolist() == [[1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 2, 2], [1, 0, 3, 1, 0, 0]]
    assert synth_hash_threshold([0, 1, 999999], 20240601, 0).tolist() == [41283, 60245, 30500]
    assert synth_hash_threshold([0, 1, 999999], 20240601, 1).tolist() == [8767, 29525, 3221]
    assert synth_hash_threshold([0, 1, 999999], 20240601, 2).tolist() == [21347, 31671, 15476]
    assert synth_hash_block_packed(10, 0, 2, 7, 0.1, 0, True).tolist() == [[98, 24, 252], [102, 106, 249]]
    # any sub-range / sample subset reproduces the same cells
    full = synth_hash_geno(np.arange(50), 100, 40, 3, 0.02, 2, True)
    assert np.array_equal(synth_hash_geno([7, 31], 110, 5, 3, 0.02, 2, True), full[10:15][:, [7, 31]])


def test_pass_plan_covers_every_panel_once_and_fits_budget():
    """KING-robust at N = 500 000 (configs[4]): 20 B of counters per element of the panel rectangles; the plan must
    cover every panel exactly once over its passes and respect the memory budget it was asked for."""
    from snprelate_amd.dist import pass_plan, passes_needed, panel_storage
    n, world = 500000, 8
    q = passes_needed(n, world, 20.0, 0.7 * 288e9)
    assert q == 4
    for ppr in (1, 2):
        bounds, owned, mx = pass_plan(n, world, ppr, q, 20.0)
        seen = sorted(p for ps in owned for r in ps for p in r)
        assert seen == list(range(world * ppr * q))
        assert bounds[0] == 0 and bounds[-1] == n and all(b % 256 == 0 for b in bounds[:-1])
        worst = max(sum(panel_storage(n, bounds[p], bounds[p + 1]) for p in r) for ps in owned for r in ps) * 20.0
        assert worst == mx and mx <= 0.7 * 288e9
    # a tiny problem: empty panels are legal and every rank still gets its slots
    bounds, owned, _ = pass_plan(1300, 2, 1, 2, 20.0)
    assert sorted(p for ps in owned for r in ps for p in r) == [0, 1, 2, 3]


def test_file_slab_sink_roundtrip(tmp_path):
    """FileSlabSink / read_file_slabs: slabs written per panel (by two "ranks") concatenate to the packed triangle."""
    import torch
    from snprelate_amd.dist import panel_rows, slab_range
    from snprelate_amd.multigpu import FileSlabSink, read_file_slabs
    n = 700
    tri = np.arange(n * (n + 1) // 2, dtype=np.float64) * 0.5
    b = panel_rows(n, 3)
    for rank, panels in ((0, [0, 2]), (1, [1])):
        sink = FileSlabSink(str(tmp_path), rank=rank, chunk_elems=1000)       # several chunks per slab
        for p in panels:
            lo, hi = slab_range(n, b[p], b[p + 1])
            sink.put("grm", p, b[p], b[p + 1], torch.from_numpy(tri[lo:hi].copy()))
        ent = sink.close()
        assert [e["panel"] for e in ent] == panels
    assert np.array_equal(read_file_slabs(str(tmp_path), "grm", n), tri)
    assert np.isnan(read_file_slabs(str(tmp_path), "other", n)).all()


def test_single_product_syrk_arithmetic_model():
    """numpy model of the arithmetic behind syrk_uv_kernel (HISTORY.md 4.2c), independent of the GPU: (1) the two operands
    (g - c_a) u and (g - c_b) v are exact fp16 numbers and their products exact fp32 numbers; (2) product - row term -
    column term + constant equals u v (g_i - avg)(g_j - avg) identically; (3) the weight u v found by the 1024-mantissa
    search is within 5e-6 of y^2 = 1 / (p (1 - p)); (4) the centre rule keeps the running mean of the products within a
    few units over every 64-SNP chunk while using a far centre only where it costs <= 6x in product variance."""
    rng = np.random.default_rng(41)
    n, L = 96, 1024
    p = rng.uniform(0.02, 0.98, L)
    g = ((rng.random((L, n)) < p[:, None]).astype(np.float64) + (rng.random((L, n)) < p[:, None]))
    s = g.sum(1)
    keep = (s > 0) & (s < 2 * n)
    g, s = g[keep], s[keep]
    L = g.shape[0]
    avg = s / n
    y2 = 1.0 / (0.5 * avg * (1 - 0.5 * avg))
    f16 = lambda x: np.float16(x).astype(np.float64)
    mant = 1.0 + np.arange(1024) / 1024.0
    u = np.empty(L); v = np.empty(L)
    for k in range(L):
        cand = mant * 2.0 ** np.floor(np.log2(np.sqrt(y2[k])))
        vv = f16(np.float32(y2[k]) / cand.astype(np.float32))
        best = np.argmin(np.abs(cand * vv - y2[k]))
        u[k], v[k] = cand[best], vv[best]
    w = u * v
    assert np.max(np.abs(w / y2 - 1)) < 5e-6
    ca = np.empty(L); cb = np.empty(L); worst = 0.0
    for c0 in range(0, L, 64):
        cum = 0.0
        for k in range(c0, min(L, c0 + 64)):
            a = avg[k]; near = np.rint(a); far = near + (1.0 if a > near else -1.0)
            if far < 0 or far > 2:
                far = near
            dn, df, var = a - near, a - far, 0.5 * a * (2 - a)
            mnn, mnf = dn * dn * w[k], dn * df * w[k]
            ca[k] = cb[k] = near
            if far != near and (var + dn * dn) * (var + df * df) <= 6 * var * var and abs(cum + mnf) < abs(cum + mnn):
                cb[k] = far; cum += mnf
            else:
                cum += mnn
            worst = max(worst, abs(cum))
    assert worst < 8.0
    A = (g - ca[:, None]) * u[:, None]
    B = (g - cb[:, None]) * v[:, None]
    assert np.array_equal(f16(A), A) and np.array_equal(f16(B), B)                 # (1) exact fp16 operands
    prod = A[:, :, None] * B[:, None, :]
    assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)        #     exact fp32 products
    da, db = avg - ca, avg - cb
    R = ((db * w)[:, None] * (g - ca[:, None])).sum(0)
    Q = ((da * w)[:, None] * (g - cb[:, None])).sum(0)
    K = (da * db * w).sum()
    lhs = prod.sum(0) - R[:, None] - Q[None, :] + K
    z = (g - avg[:, None]) * np.sqrt(w)[:, None]
    rhs = z.T @ z
    assert np.max(np.abs(lhs - rhs)) < 1e-9 * np.max(np.abs(rhs))                  # (2) the identity


def test_converted_operand_form_of_the_single_product_syrk():
    """numpy restatement of syrk_uv16c_kernel's operand path (SNPGPU_SYRK_UV16=2 / 3), independent of the GPU:
    (1) transpose8_kernel's nibble bytes c0 | c1 << 4 come from its pair nibbles c0 + 4 c1 by the mask / shift it uses, and
        uvcorr_kernel's decode gets the codes back;
    (2) an e2m1 nibble holding the code c has the value c / 2 (what v_cvt_scalef32_pk_f16_fp4 makes of it at scale 1);
    (3) fma(c / 2, 2 u, -c_a u) evaluated in fp16 IS (c - c_a) u -- the table value of syrk_uv16_kernel -- for every code, centre
        and every fp16 mantissa of u over the exponents the weights take (one rounding of an exactly representable result);
    (4) uv_tables_kernel's factor layout: dword ((side * 2 + kind) * 4 + quarter) * 4 + d of a 32-slot group holds the pair of
        slots 8 quarter + 2 d, + 1 -- the K elements 8 quarter .. + 7 that lane quarter `quarter` of a 16 x 16 x 32 MFMA carries."""
    # (1)
    codes = np.arange(16, dtype=np.uint32)                      # c0 + 4 c1, c in 0..3
    v = codes | (codes << 8) | (codes << 16) | (codes << 24)
    nib = (v & 0x03030303) | ((v & 0x0C0C0C0C) << 2)
    for n in range(16):
        by = int(nib[n]) & 0xFF
        assert by == (n & 3) | ((n >> 2) << 4) and (by & 3, by >> 4) == (n & 3, n >> 2)
    # (2) e2m1: sign, two exponent bits, one mantissa bit -> 0, 0.5, 1, 1.5, 2, 3, 4, 6
    e2m1 = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]
    assert [e2m1[c] for c in range(3)] == [0.0, 0.5, 1.0]
    # (3)
    mant = (1.0 + np.arange(1024) / 1024.0)
    for e in range(-3, 9):                                      # u from 0.125 (EIGMIX-like weights) to 511 (rare variants' y)
        u = np.float16(mant * 2.0 ** e)
        assert np.array_equal(u.astype(np.float64), mant * 2.0 ** e)
        for ca in range(3):
            f1, f0 = np.float16(2.0) * u, np.float16(-float(ca)) * u
            assert np.array_equal(f1.astype(np.float64), 2.0 * u.astype(np.float64))
            for c in range(3):
                x = np.float16(e2m1[c])
                got = np.float16(x.astype(np.float64) * f1.astype(np.float64) + f0.astype(np.float64))      # a fused multiply-add: one rounding
                want = (c - ca) * u.astype(np.float64)
                assert np.array_equal(got.astype(np.float64), want)
    # (4)
    seen = set()
    for slot in range(0, 32, 2):                                # the even slot of a pair writes the dword
        pp = slot >> 1
        kq, d = pp >> 2, pp & 3
        assert 8 * kq + 2 * d == slot
        for e in range(4):
            seen.add((e * 4 + kq) * 4 + d)
    assert seen == set(range(64))                               # 64 dwords = 256 bytes per group, every one written once


def test_gds_stream_reader_hapmap_blocks_into_caller_buffers(hapmap):
    """Streaming block reader (snpRead + CGenoReadBySNP minus the byte inflation, src/dGenGWAS.cpp:677-733, :1218-1397):
    the HapMap fixture block by block -- ragged block sizes, 279 samples (69.75 bytes per SNP: the bit carry across SNP
    boundaries), rows written into two caller-owned buffers in turn -- equals the whole-file reader."""
    from snprelate_amd.gds import open_gds_stream
    gs = open_gds_stream(os.path.join(GOLDEN, "hapmap_geno.gds"))
    assert (gs.n_snp, gs.n_samp) == (9088, 279) and gs.sample_order
    assert np.array_equal(gs.snp_id, hapmap.snp_id) and np.array_equal(gs.snp_chromosome, hapmap.snp_chromosome)
    for blk in (1, 3, 1000, 4096, 9088, 20000):
        bufs = [np.empty(min(blk, 9088) * 70, np.uint8) for _ in range(2)]
        seen = 0
        for lo, m, rows in gs.blocks(blk, buffers=bufs):
            assert lo == seen and rows.shape == (m, 70) and rows.base is not None
            assert np.array_equal(rows, hapmap.packed[lo:lo + m])
            seen += m
            if blk == 1 and seen >= 40:
                break
        assert seen == (9088 if blk > 1 else 40)
    assert np.array_equal(gs.read_packed(777), hapmap.packed)
    # a sub-range of the SNPs
    got = np.concatenate([r.copy() for _, _, r in gs.blocks(500, snp_begin=1234, snp_end=3001)], 0)
    assert np.array_equal(got, hapmap.packed[1234:3001])


@pytest.mark.parametrize("n_samp", [1, 4, 7, 10, 279, 1001])
@pytest.mark.parametrize("zipped", [False, True])
def test_gds_stream_reader_compressed_and_chained_extents(n_samp, zipped, tmp_path):
    """The byte-stream layer under the block reader: a genotype bit stream stored in several chained file extents, raw or
    zlib-compressed, is read (and inflated) incrementally and realigned to byte-aligned rows."""
    import zlib
    from snprelate_amd.gds import _ByteStream, pack_2bit_rows, realign_bit2_rows
    rng = np.random.default_rng(n_samp)
    L = 531
    g = rng.integers(0, 4, size=(L, n_samp), dtype=np.uint8)
    flat = g.reshape(-1)
    pad = (-flat.size) % 4
    q = np.concatenate([flat, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    stream = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8).tobytes()   # continuous bit2 stream
    payload = zlib.compress(stream, 6) if zipped else stream
    cuts = sorted(set([0, len(payload) // 3, len(payload) // 3 + 1, 2 * len(payload) // 3, len(payload)]))
    fn, extents = tmp_path / "s.bin", []
    with open(fn, "wb") as f:
        for a, b in zip(cuts, cuts[1:]):
            f.write(b"JUNK" * 5)
            extents.append((f.tell(), b - a))
            f.write(payload[a:b])
    want = pack_2bit_rows(g)
    with open(fn, "rb") as f:
        st = _ByteStream(f, extents, len(payload), zipped=zipped, chunk=97)
        for lo in range(0, L, 100):
            hi = min(L, lo + 100)
            b0, b1 = (2 * n_samp * lo) >> 3, (2 * n_samp * hi + 7) >> 3
            rows = realign_bit2_rows(st.read(b0, b1), 8 * b0, n_samp, lo, hi)
            assert np.array_equal(rows, want[lo:hi]), (lo, hi)


@pytest.mark.parametrize("layout", ["raw", "zip_chained"])
def test_gds_stream_reader_snp_order_nodes(layout, tmp_path):
    """A snp.order genotype node (dims [n_samp][n_snp], SNPs fastest: src/dGenGWAS.cpp:576-589) streamed as SNP blocks: raw and
    contiguous through a memory map; zlib-compressed and / or chained over several file extents through ONE inflation into host
    memory (round 4: such nodes were refused)."""
    import zlib
    from snprelate_amd.gds import GenoStream, pack_2bit_rows
    rng = np.random.default_rng(12)
    n, L = 37, 523
    g = rng.integers(0, 4, size=(L, n), dtype=np.uint8)          # [snp][sample]
    flat = np.ascontiguousarray(g.T).reshape(-1)                 # sample-major: sample i's L codes, then sample i + 1's
    q = np.concatenate([flat, np.zeros((-flat.size) % 4, np.uint8)]).reshape(-1, 4)
    stream = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8).tobytes()
    payload = stream if layout == "raw" else zlib.compress(stream, 6)
    cuts = [0, len(payload)] if layout == "raw" else [0, len(payload) // 2, len(payload) // 2 + 3, len(payload)]
    fn, extents = tmp_path / "g.bin", []
    with open(fn, "wb") as f:
        for a, b in zip(cuts, cuts[1:]):
            f.write(b"PAD!" * 3)
            extents.append((f.tell(), b - a))
            f.write(payload[a:b])
    gs = GenoStream(str(fn), np.arange(n).astype(str), np.arange(L), np.ones(L, np.int32), [n, L], extents, len(payload),
                    layout != "raw", False)
    assert (gs.n_snp, gs.n_samp) == (L, n)
    want = pack_2bit_rows(g)
    got = np.concatenate([r.copy() for _, _, r in gs.blocks(100)], 0)
    assert np.array_equal(got, want)
    part = np.concatenate([r.copy() for _, _, r in gs.blocks(64, snp_begin=77, snp_end=300)], 0)
    assert np.array_equal(part, want[77:300])
    if layout != "raw":
        os.environ["SNPGPU_GDS_INFLATE_MAX"] = "16"
        try:
            gs2 = GenoStream(str(fn), np.arange(n).astype(str), np.arange(L), np.ones(L, np.int32), [n, L], extents, len(payload), True, False)
            with pytest.raises(ValueError, match="SNPGPU_GDS_INFLATE_MAX"):
                next(iter(gs2.blocks(100)))
        finally:
            del os.environ["SNPGPU_GDS_INFLATE_MAX"]


def test_gds_node_coder_tags():
    """The compression coder of an array node is read from its descriptor by name: anything but "" and "ZIP" is refused up front
    (a node compressed with LZ4 / LZMA or a random-access container was once streamed as raw 2-bit data)."""
    from snprelate_amd.gds import _node_coder, _node_info

    def coder_prop(text):                      # gdsfmt's tagged, length-prefixed coder property (as in the HapMap file's snp.id node)
        return b"\xc4\x46\x6d\x10" + bytes([len(text)]) + text

    assert _node_coder(b"\x00dBit2\x00") == "" and _node_coder(b"c\x00" + coder_prop(b"ZIP") + b"\x02") == "ZIP"
    # a chance "LZ4" / "ZIP" among the other descriptor bytes of an uncompressed node names no coder
    assert _node_coder(b"\x05LZ4xx\x00ZIP\x07") == ""
    # a level suffix after a dot is not part of the coder's name: "ZIP.max" is a plain zlib stream, "ZIP_RA.max:256K" is not
    assert _node_coder(b"x" + coder_prop(b"ZIP.max")) == "ZIP" and _node_coder(b"x" + coder_prop(b"zip.fast:1M")) == "ZIP"
    assert _node_coder(b"x" + coder_prop(b"ZIP_RA.max:256K")) == "ZIP_RA"
    for tag in ("ZIP_RA", "LZ4", "LZ4_RA", "LZMA", "LZMA_RA"):
        assert _node_coder(b"x" + coder_prop(tag.encode() + b":256K")) == tag
        desc = b"hdr" + coder_prop(tag.encode()) + b"\x00" + b"\xc3\x43\x61" + bytes([8]) + (5).to_bytes(4, "little") + (7).to_bytes(4, "little") + \
               b"\xc4\xc3\x7c\x0c" + (3).to_bytes(4, "little")
        with pytest.raises(ValueError, match=tag):
            _node_info(desc)
    desc = b"hdr\x00" + b"\xc3\x43\x61" + bytes([8]) + (5).to_bytes(4, "little") + (7).to_bytes(4, "little") + b"\xc4\xc3\x7c\x0c" + \
           (3).to_bytes(4, "little") + b"attr"
    assert _node_info(desc) == ([5, 7], 3, False, b"attr")


def test_r_shim_compiles_against_mock():
    """r_shim/gpu_shim.cpp cannot be built here (no R, no gdsfmt, no SNPRelate sources in the build image); a mock header that
    only declares the names it uses (tests/mock_r/dGenGWAS.h) lets g++ check its syntax and its use of include/snpgpu.h --
    argument counts and types of every libsnpgpu call in the seven `.Call` bodies."""
    import subprocess
    r = subprocess.run(["g++", "-std=gnu++14", "-fsyntax-only", "-Wall", "-I" + os.path.join(ROOT, "tests", "mock_r"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "r_shim", "gpu_shim.cpp")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_fp4_operand_algebra_of_the_counter_kernels():
    """The MX-fp4 counter kernels (HISTORY.md 4.1d) build their e2m1 operands from the 2-bit codes by bit logic.  Restated here in
    Python integers, exactly as kernels_pair.hip writes it (Fp4Scheme<>::types, Fp4NomissPipe::decode): every nibble, decoded as
    e2m1 and scaled by the block scale 2, must be the value type it stands for -- including the free sign bit of a zero -- and the
    products summed over SNPs must give the oracle's counters."""
    E2M1 = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]

    def val(nib):                                  # e2m1 nibble x block scale 2
        return (-1.0 if nib & 8 else 1.0) * E2M1[nib & 7] * 2.0

    M, M8 = 0x11111111, 0x88888888
    F = 0xFFFFFFFF

    def types(x, t, x2, x3):                       # {v, s, y, x, h} of one half-word (eight SNPs)
        v = M & ~(x & t) & F
        y = M & ~x & F
        return {"v": v, "s": v | (x3 & M8), "y": y, "x": y | (x2 & M8), "h": M & x & ~t & F}

    rng = np.random.default_rng(5)
    codes = rng.integers(0, 4, size=(2, 4096)).astype(np.int64)          # two samples, codes 0 1 2 3 (3 = missing)
    want = {"v": lambda c: float(c != 3), "s": lambda c: {0: 1.0, 1: -1.0, 2: 1.0, 3: 0.0}[c], "y": lambda c: float(c in (0, 2)),
            "x": lambda c: {0: 1.0, 1: 0.0, 2: -1.0, 3: 0.0}[c], "h": lambda c: float(c == 1)}
    dec = {k: np.zeros((2, codes.shape[1])) for k in want}
    g_half = np.zeros((2, codes.shape[1]))
    for smp in range(2):
        for w0 in range(0, codes.shape[1], 16):
            w = 0
            for k in range(16):
                w |= int(codes[smp, w0 + k]) << (2 * k)
            halves = ((w, (w >> 1), (w << 2) & F, (w << 3) & F, 0), ((w >> 2), (w >> 3), w, (w << 1) & F, 1))
            for x, t, x2, x3, odd in halves:
                ty = types(x, t, x2, x3)
                for k in range(8):
                    snp = w0 + 2 * k + odd
                    for name in want:
                        dec[name][smp, snp] = val((ty[name] >> (4 * k)) & 15)
                    g_half[smp, snp] = val((x & 0x33333333) >> (4 * k) & 15)     # the two-product kernel's g operand: the code itself
    for name, f in want.items():
        exp = np.array([[f(int(c)) for c in row] for row in codes])
        assert np.array_equal(dec[name], exp), name                      # (-0.0 == 0.0: the free sign bit)
    assert np.array_equal(g_half, codes.astype(float))                   # nibble 0b00c1c0 = g / 2, scale 2 (3 only as padding)
    # the counters from the products, against the oracle on the same two samples (packed triangle: pair (0, 1) is entry 1)
    import oracle as orc
    g = np.ascontiguousarray(codes.T.astype(np.uint8))                   # [L, 2]
    ibs = orc.ibs_count(g)[1]                                            # {ibs0, ibs1, ibs2}
    a = {k: float(dec[k][0] @ dec[k][1]) for k in ("v", "s", "y", "x", "h")}
    nvalid, ibs1, ibs0x2 = a["v"], (a["v"] - a["s"]) / 2, a["y"] - a["x"]
    assert (ibs0x2 / 2, ibs1, nvalid - ibs1 - ibs0x2 / 2) == tuple(float(v) for v in ibs)
    # KING-robust's basis {y.y', x.x', y.h', h.y', h.h'}: both called, exactly one het
    yh, hy = float(dec["y"][0] @ dec["h"][1]), float(dec["h"][0] @ dec["y"][1])
    assert a["y"] + yh + hy + a["h"] == nvalid and yh + hy == ibs1


def test_synth_generator_c_twin_matches_the_numpy_twin():
    """oracle.synth_hash_geno_c (C + OpenMP, what the fp64 anchors of the full-size checks use) is bit-identical to the numpy twin of
    snpgpu_synth_block for every spectrum it restates, with and without missing calls and the planted edge-case SNPs."""
    import oracle
    from oracle.synth import synth_hash_geno
    samp = np.r_[np.arange(96), 499999 - 7 * np.arange(40)]
    for spectrum in (0, 1, 2, 3):               # 3: falls back to the numpy form
        for missing in (0.0, 0.05):
            for special in (False, True):
                a = synth_hash_geno(samp, 987000, 2100, 20240601, missing, spectrum, special)
                b = oracle.synth_hash_geno_c(samp, 987000, 2100, 20240601, missing, spectrum, special)
                assert a.dtype == b.dtype == np.uint8 and np.array_equal(a, b), (spectrum, missing, special)


def test_fp64_anchor_equals_the_oracle_on_a_small_set():
    """tests/fp64_anchor.py (the fp64 recomputation that anchors the whole-panel accuracy figures) against the C oracle's full GCTA
    matrix on a set small enough for both: same entries to 1e-12."""
    import oracle as orc
    from oracle.synth import synth_hash_geno
    from fp64_anchor import Fp64Anchor, tri_index
    n, L = 700, 3000
    g = synth_hash_geno(np.arange(n), 0, L, 20240601, missing=0.03, spectrum=1)
    ref = orc.grm_gcta(g)
    a = Fp64Anchor(n, 256, 512, 40, 40, "GRM_GCTA", 20240601, 0.03, 1)
    valid = g <= 2
    for lo in range(0, L, 1024):
        m = min(1024, L - lo)
        a.add(lo, m, (g[lo:lo + m] * valid[lo:lo + m]).sum(1), valid[lo:lo + m].sum(1))
    idx, val = a.finish()
    base = tri_index(n, 256, 256)
    assert idx.size > 400 and np.allclose(val, ref[idx + base], rtol=1e-12, atol=1e-14)


def test_bench_refuses_a_world_that_is_not_gpus():
    """`bench.py --gpus 2` inside a launch environment of ONE rank must not print an n_gpus = 1 line: it
    refuses, before any device or rendezvous is touched."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 2 and "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_needs_one_device_per_rank():
    """plain `bench.py --gpus 2` with fewer than two HIP devices visible (none here): an error and exit code 2, never a silent
    one-rank run"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SNPGPU_BENCH_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 2 and "HIP device" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
