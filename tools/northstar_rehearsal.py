#!/usr/bin/env python3
"""The north_star job -- GCTA-method GRM + top-32 eigenvectors of 500 000 samples x 1 000 000 SNPs on the 8 GPUs of one node --
end to end through the C ABI (no torch algebra: torch only holds the block buffer), as ONE command:

    python tools/northstar_rehearsal.py --mode whole --n 500000            # on an 8 x MI355X node: THE record (all visible devices)
    python tools/northstar_rehearsal.py --mode whole --devices 0,0,0,0,0,0,0,0 --n 20000     # the same code path on one GPU

  --mode whole  (default N = 150 000 on one device; --n 500000 on a node)   snpgpu_multi over --devices (default: every visible
                device; an ordinal may repeat), panels per device chosen by the library so that accumulators, the GCTA both-missing
                plane, the eigen solver's fp32 copy and every panel's scratch fit the devices' free memory (--panels-per-device -1),
                the exchange path's self-test FIRST (snpgpu_multi_comm_selftest: RCCL on distinct devices), every block of the
                1 000 000-SNP data set, snpgpu_multi_finalize_inplace (the GCTA numerator becomes the GRM in place),
                snpgpu_multi_topk_eigen (block Krylov; vector block broadcast to / partial products reduced over the devices),
                the final gather of the packed triangle where a host can hold it (N <= --gather-max), and SURVEY 8(d)'s sampled-tile
                parity in the same run: 64 x 64 sample pairs x ALL SNPs recomputed in fp64 on the host (tests/fp64_anchor.py) against
                the finalised entries read back from the panels (snpgpu_panel_entries).  One JSON line: accumulate / finalise /
                eigen / gather seconds, the parity figures, eigen residual.
  --mode check  (default N = 12 000)   the same pipeline at a size LAPACK reaches in a minute: eigenvalues, residuals and
                the subspace of the top-k eigenvectors against numpy's eigh (the reference's route: LAPACK on the host) of the
                gathered matrix.
  --mode share  (default N = 500 000)   rank `--rank` of the 8-rank plan at the job's real size: its panel(s) take ALL
                blocks, are finalised in place, and the Krylov solver runs two restart cycles on the rank's PART of the
                matrix (a symmetric matrix in its own right): the per-product cost of the solver at N = 500 000 -- panel
                product + tall-skinny algebra -- on one rank's memory footprint.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="whole", choices=["whole", "check", "share"])
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--snps", type=int, default=1000000)
    ap.add_argument("--block", type=int, default=65536)
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--panels-per-device", type=int, default=-1, help="-1: chosen by the library from the devices' free memory")
    ap.add_argument("--devices", default="", help="comma-separated HIP ordinals (may repeat); default: every visible device (check mode: 0,0)")
    ap.add_argument("--gather-max", type=int, default=120000, help="gather the packed triangle on the host up to this many samples")
    ap.add_argument("--parity-samples", type=int, default=64, help="K: K x K sampled pairs x all SNPs in fp64 (0: off)")
    ap.add_argument("--kind", default="GRM_GCTA", choices=["GRM_GCTA", "PCA_COV"])
    ap.add_argument("--depth", type=int, default=0, help="Krylov blocks per restart cycle (0 = the solver's default)")
    ap.add_argument("--eig-block", type=int, default=0, help="vectors per Krylov block (0 = k + 8 rounded up to 16)")
    ap.add_argument("--fp32-until", type=float, default=0.0, help="snpgpu_eig_opts.fp32_until (0 = default, < 0 = fp64 only)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import numpy as np
    import torch
    from snprelate_amd import _lib
    n = a.n or {"whole": 150000, "check": 12000, "share": 500000}[a.mode]
    B, kind = a.block, getattr(_lib, a.kind)
    res = {"mode": a.mode, "n": n, "snps": a.snps, "block_snps": B, "missing": a.missing, "k": a.k, "kind": a.kind}
    buf = [torch.empty((B, (n + 3) // 4), dtype=torch.uint8, device="cuda") for _ in range(2)]

    def stream(feed, sync):
        t0 = time.perf_counter()
        for i, lo in enumerate(range(0, a.snps, B)):
            m = min(B, a.snps - lo)
            if i >= 2:
                sync()                          # the buffer about to be rewritten was fed two blocks ago
            _lib.synth_block(buf[i % 2].data_ptr(), n, lo, m, 20240601, missing=a.missing)
            feed(buf[i % 2].data_ptr(), m)
        sync()
        return time.perf_counter() - t0

    if a.mode in ("whole", "check"):
        if a.devices:
            devs = tuple(int(x) for x in a.devices.split(","))
        else:
            devs = tuple(range(_lib.device_count())) if a.mode == "whole" else (0, 0)
        m = _lib.MultiAccumulator(kind, n, devices=devs, panels_per_device=a.panels_per_device, max_block_snps=B)
        res["devices"] = list(devs)
        res["panels"] = m.panels()
        res["panels_per_device"] = m.status()["panels_per_device"]           # what the library chose when --panels-per-device is -1
        t0 = time.perf_counter()
        res["comm_selftest_uses_rccl"] = m.comm_selftest()          # raises on a wrong sum: nothing is accumulated on a broken exchange path
        res["comm_selftest_s"] = time.perf_counter() - t0
        # round 6: what the object found out about its devices -- peer access per ordered pair of distinct devices, and the outcomes of
        # the eigen-exchange, feed-forward and gather self-tests (1 = passed; the call above raises on a failure)
        st = m.status()
        res["peer_access"] = {"pairs": st["peer_pairs"], "enabled": st["peer_pairs_enabled"]}
        res["selftest_comm"], res["selftest_feed"], res["selftest_gather"] = st["selftest_comm"], st["selftest_feed"], st["selftest_gather"]
        res["distinct_devices"] = st["n_distinct_devices"]
        res["device_pci"] = [_lib.device_pci(d) for d in sorted(set(devs))]
        anchor = None
        if a.parity_samples > 0 and a.kind == "GRM_GCTA":
            # sampled tiles recomputed by the CPU in fp64 -- K rows of the panel that holds sample n / 2, K columns
            # spread from there to the last sample; the block's per-SNP statistics come from the device copy of the block
            from fp64_anchor import Fp64Anchor, block_stats_torch
            mid = [p for p in res["panels"] if p[0] <= n // 2 < p[1]][0]
            anchor = Fp64Anchor(n, mid[0], mid[1], a.parity_samples, a.parity_samples, "GRM_GCTA", 20240601, a.missing, 0)
        t_par = [0.0]

        def feed(ptr, m_snp, _state={"lo": 0}):
            m.feed_device(ptr, m_snp)
            if anchor is not None:
                t1 = time.perf_counter()
                i = (_state["lo"] // B) % 2
                anchor.add(_state["lo"], m_snp, *block_stats_torch(buf[i][:m_snp]))
                t_par[0] += time.perf_counter() - t1
            _state["lo"] += m_snp

        res["accumulate_s"] = stream(feed, m.sync)
        res["parity_host_s"] = t_par[0]            # host work of the parity check, done UNDER the asynchronous kernels of the same block
        res["pair_genotypes_per_s"] = n * n / 2 * a.snps / res["accumulate_s"]
        t0 = time.perf_counter()
        m.finalize_inplace()
        m.sync()
        res["finalize_inplace_s"] = time.perf_counter() - t0
        free, total = torch.cuda.mem_get_info()
        res["hbm_in_use_gib"] = (total - free) / 2 ** 30
        t0 = time.perf_counter()
        w, v, info = m.topk_eigen(a.k, scale=1.0 if a.kind == "GRM_GCTA" else 0.0, depth=a.depth, block=a.eig_block,
                                     fp32_until=a.fp32_until)
        res["eigen_s"] = time.perf_counter() - t0
        res["eigen_info"] = info
        res["eigenvalues_head"] = [float(x) for x in w[:6]]
        res["total_s"] = res["accumulate_s"] + res["finalize_inplace_s"] + res["eigen_s"]
        if anchor is not None:
            idx, f64 = anchor.finish()
            keep = anchor.cols[None, :] >= anchor.rows[:, None]
            rr = np.broadcast_to(anchor.rows[:, None], keep.shape)[keep]
            cc = np.broadcast_to(anchor.cols[None, :], keep.shape)[keep]
            got = m.entries(rr, cc)
            off = rr != cc
            med_off = float(np.median(np.abs(f64[off]))) if off.any() else 1.0
            med_diag = float(np.median(f64[~off])) if (~off).any() else 1.0
            d = np.abs(got - f64)
            res["parity"] = {"pairs": int(idx.size), "what": "finalised GRM entries vs fp64 on the host over all SNPs (tests/fp64_anchor.py)",
                             "max_rel_1e-5_contract": float(np.max(d / (np.abs(f64) + med_diag))),
                             "max_offdiag_figure": float(np.max(d / (np.abs(f64) + med_off))), "median_abs_offdiag": med_off}
        if a.mode == "whole" and n <= a.gather_max and a.kind == "GRM_GCTA":
            t0 = time.perf_counter()
            tri = m.grm_gcta()
            res["gather_s"] = time.perf_counter() - t0
            res["gather_what"] = "packed triangle (%.1f GB) from the panels into host memory" % (tri.nbytes / 1e9)
            del tri
        else:
            res["gather_s"] = None
        if a.mode == "check":
            # the reference's own route: LAPACK on the host (numpy) on the gathered matrix
            tri = m.grm_gcta() if a.kind == "GRM_GCTA" else m.pca_cov()[0]
            m.close()
            full = np.zeros((n, n))
            full[np.triu_indices(n)] = tri
            full = full + np.triu(full, 1).T
            del tri
            t0 = time.perf_counter()
            wd, vd = np.linalg.eigh(full)
            res["lapack_eigh_s"] = time.perf_counter() - t0
            wd, vd = wd[::-1][:a.k + 1], vd[:, ::-1][:, :a.k]
            res["eigenvalue_max_rel_diff"] = float(np.max(np.abs(w - wd[:a.k]) / np.abs(wd[:a.k])))
            # residuals of the returned pairs against the gathered matrix, principal angles between the two top-k subspaces,
            # per-vector cosines where the eigenvalue is separated from both neighbours
            res["max_rel_residual_vs_gathered_matrix"] = float(np.max(np.linalg.norm(full @ v - v * w, axis=0) / np.abs(w)))
            sv = np.linalg.svd(v.T @ vd, compute_uv=False)
            res["subspace_min_cosine"] = float(sv.min())
            d = np.abs(np.diff(wd)) / wd[0]
            sep = np.minimum(np.r_[np.inf, d[:-1]], d)[:a.k] > 1e-4
            cos = np.abs(np.sum(v * vd, axis=0))
            res["separated_vectors"] = int(sep.sum())
            res["separated_vectors_min_cosine"] = float(np.min(cos[sep])) if sep.any() else None
        else:
            m.close()
    else:
        from snprelate_amd.dist import panel_plan
        bounds, owned = panel_plan(n, a.world, 1)
        p = owned[a.rank][0]
        r0, r1 = bounds[p], bounds[p + 1]
        acc = _lib.Accumulator(kind, n, row_begin=r0, row_end=r1 if (r1 < n or r0 > 0) else 0, max_block_snps=B)
        res["panel_rows"] = [r0, r1]
        res["accumulate_s"] = stream(acc.feed_device, acc.sync)
        res["pair_genotypes_per_s_this_rank"] = (acc.slab_size() * a.snps) / res["accumulate_s"]
        t0 = time.perf_counter()
        acc.finalize_inplace()
        acc.sync()
        res["finalize_inplace_s"] = time.perf_counter() - t0
        free, total = torch.cuda.mem_get_info()
        res["hbm_in_use_gib"] = (total - free) / 2 ** 30
        import ctypes
        opts = _lib.EigOpts(tol=1e-30, block=0, depth=0, max_restarts=2, seed=1, y_buf=None, reduce=_lib.REDUCE_FN(), user=None,
                            fp32_until=a.fp32_until)
        handles = (ctypes.c_void_p * 1)(acc._h)
        info = _lib.EigInfo()
        w = np.empty(a.k)
        t0 = time.perf_counter()
        scale = 1.0
        if a.kind == "PCA_COV":
            scale = (n - 1) / acc.pca_panel_trace()
        # two cycles against an unreachable tolerance: a timing run -- the solver reports (correctly, since round 4) that the pairs did
        # not converge; its counters are filled in either way
        rc = _lib.lib().snpgpu_panels_topk_eigen(handles, 1, scale, a.k, ctypes.byref(opts), _lib._ptr(w), None, _lib.HOST, ctypes.byref(info))
        if rc != 0 and "not converged" not in _lib.lib().snpgpu_last_error().decode("utf-8", "replace"):
            _lib.check(rc)
        dt = time.perf_counter() - t0
        res["residual_after_two_cycles"] = info.max_rel_residual
        res["krylov_two_cycles_s"] = dt
        res["krylov_products"] = info.matmuls
        res["krylov_products_fp32"] = info.matmuls_fp32
        res["s_per_product_incl_algebra"] = dt / max(info.matmuls, 1)
        acc.close()
    line = json.dumps(res, sort_keys=True)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
