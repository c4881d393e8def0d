#!/usr/bin/env python3
"""bench.py -- throughput of the pairwise hot path on MI355X.

Metric (BASELINE.json): SNP-pair-genotypes/sec = N^2 * L / 2 / t.

A "step" is one pass of the hot path over one feed block of B SNPs for ALL sample pairs (pre-pass + pair
kernel + accumulation into the resident N x N panel).  The timed region is K steps PLUS one finalise of the
panel into the caller's packed-triangle buffer (t = accumulate + finalise), with the 2-bit
packed synthetic genotypes resident in HBM before it starts (blocks come from the counter-based generator
snpgpu_synth_block; oracle/synth.py is its CPU twin).

Default workload = BASELINE.json configs[2]: snpgdsGRM method="GCTA", synthetic N = 100 000 samples
(x 1 000 000 SNPs = 15 steps of 65 536 SNPs + a remainder; every step is identical work; round 4: 65 536-SNP feed blocks, the upper
clamp of the reference's own block size, src/genIBS.cpp:286-289 -- a block then runs as SIX fp32 runs with six weight targets: the same
flush rate as three runs per 32 768 SNPs at a smaller weight error, DESIGN.md 4.2 / HISTORY.md 4.2d).  Other workloads:
  --workload ibs    configs[1]  snpgdsIBSNum   N = 10 000
  --workload king   snpgdsIBDKING robust       N = 10 000, 5 % missing
  --workload pca    snpgdsPCA covariance       N = 100 000
At N = 1 the JSON line also carries short runs of the other paths, one row each in `summary` -- the LAST key of the line, so that
it survives the driver's 8 KB tail -- [value, ms per step, roofline fraction of the dominant kernel, kernel, kernel ms per step]: ibs,
ibs_missing_0.02, king, king_missing_0, king_homo, the real-data path grm_missing_0.02, grm_feed_pinned_2bit (SURVEY 8(d)'s "end-to-end
incl. feed": the same steps with every block coming from pinned host memory), grm_exact_row, grm_run8192, grm_fast and the north_star
fp32-MFMA tile grm_f32; --details FILE writes their long form.  The CPU baseline of SURVEY 8(d) rides in `cpu_baseline`.
The headline workload has NO missing calls (imputed data); data with missing calls takes the `grm_missing_0.02` path.
Multi-GPU (--gpus N): the output triangle is cut into row panels of equal time, one per rank; the total problem is fixed =>
"strong" scaling.  Launched as the driver does (python -m torch.distributed.run ... bench.py --gpus N) or as a plain command --
without a torch.distributed environment `python bench.py --gpus N` launches its own N ranks (one per device), and in every case a rank
whose WORLD_SIZE is not --gpus refuses to run.  The line then says what the collective library saw (config.rccl_ranks, collective_backend,
rank_devices = "ordinal@PCI address architecture" per rank, distinct_devices) and times the final RCCL gather of the slabs on rank 0
AFTER the timed region (config.gather_ms; never part of `value`; --no-gather skips it).  From four ranks the per-SNP statistics of a
block are computed once per node (--shared-stats).
Round 6: config.sclk_mhz_median / power_w_median = shader clock and socket power sampled during the timed region (amdsmi), and
roofline.sustained_peak_measured = what a register-only stream of the kernel's MFMA instruction and operand class sustains on THIS box in
THIS run (snpgpu_diag_mfma_rate; config.sustained_probe holds every class) -- the kernels run against the socket power cap.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20240601
WORKLOADS = {
    #            kind           N        B      missing  metric kernel (0 pair counters / 1 SYRK)
    "grm":  dict(kind="GRM_GCTA", n=100000, b=65536, missing=0.0, which=1,
                 name="snpgdsGRM method=GCTA, synthetic 100000 x 1000000 (configs[2]), fed in blocks of 65536 SNPs"),
    "pca":  dict(kind="PCA_COV", n=100000, b=65536, missing=0.0, which=1,
                 name="snpgdsPCA covariance, synthetic 100000 samples, blocks of 65536 SNPs"),
    # counter kernels: 65536-SNP feed blocks (the upper clamp of the reference's own block size, src/genIBS.cpp:286-289):
    # one HBM counter update per block
    "ibs":  dict(kind="IBS", n=10000, b=65536, missing=0.0, which=0,
                 name="snpgdsIBSNum, synthetic 10000 x 500000 (configs[1]), fed in blocks of 65536 SNPs"),
    "king": dict(kind="KING_ROBUST", n=10000, b=65536, missing=0.05, which=0,
                 name="snpgdsIBDKING KING-robust, synthetic 10000 samples, 5% missing, blocks of 65536 SNPs"),
    "king_homo": dict(kind="KING_HOMO", n=10000, b=65536, missing=0.05, which=0,
                      name="snpgdsIBDKING KING-homo, synthetic 10000 samples, 5% missing, blocks of 65536 SNPs"),
}
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2516.6         # dense fp16 MFMA: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz (guide: ~2.5 PF)
PEAK_VALU_TLANEOPS = 78.6             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
POP_OPS = {"IBS": 8, "KING_ROBUST": 11}   # VALU bit-ops per 32 SNP pairs (popcount backend, kernels_pair.hip)
# Measured on this part (tools/mfma_power.sh -> profiles/r01_mfma_power.txt): a register-only MFMA stream is held
# back by the socket power limit as soon as the operands are not zeros (zeros: 2470 TFLOP/s / 4940 TOP/s at 2.39 GHz).
SUSTAINED_F16_TFLOPS = {1: 1840.0, 2: 1840.0, 3: 1689.0}   # row operand in {-1,0,1} (1.85 GHz) / both operands real-valued (1.71 GHz)
SUSTAINED_I8_TOPS = {False: 4129.0, True: 4486.0}  # operands in {-1,0,1}: 2.06 GHz / blocks without missing calls, one binary and
                                                   # one {-1,0,1} product: harmonic mean of 4911 (binary, 2.39 GHz) and 4129
PEAK_I8_MFMA_TOPS = 5033.0            # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz (= 2x the dense bf16 peak)
PEAK_FP4_MFMA_TFLOPS = 10066.4        # MX-fp4 (v_mfma_scale_f32_32x32x64_f8f6f4): 4x the dense bf16 peak (guide: ~10 PF dense)
SUSTAINED_FP4_TFLOPS = 9099.0         # the guide's register-only measurement of that instruction (MI355X_MICROARCH.md)
I8_SLOTS = {"IBS": 4, "KING_ROBUST": 5, "KING_HOMO": 4}   # int8 dot products per pair-genotype (I8Scheme<> in kernels_pair.hip)
TRAFFIC_FILE = "profiles/r06_pmc_hbm_traffic.json"


def source_stamp():
    """sha256 (first 16 hex digits) over the library's sources and this file, in sorted path order: recomputable from a git
    checkout (`python bench.py --stamp`), carried by the bench line and by every file under profiles/ of the same tree."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "snprelate_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "snprelate_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "snprelate_amd", "*.py")) + [os.path.abspath(__file__)])
    for fn in files:
        h.update(os.path.relpath(fn, ROOT).encode() + b"\0")
        with open(fn, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_ranks):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: this process becomes the launcher -- the same
    command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU
    (LOCAL_RANK = HIP device), the ranks' stdout / stderr inherited so that rank 0's JSON line is this command's line.  Returns the
    launcher's exit code."""
    import subprocess
    if "SNPGPU_BENCH_FORCE_DEVICE" not in os.environ:
        import torch
        have = torch.cuda.device_count()
        if have < n_ranks:
            print("bench.py: --gpus %d but only %d HIP device(s) visible; one rank per GPU is the only layout this bench runs "
                  "(no oversubscription outside the tests' SNPGPU_BENCH_FORCE_DEVICE hook)" % (n_ranks, have), file=sys.stderr)
            return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SNPGPU_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n_ranks, " ".join(cmd[1:8])), file=sys.stderr)
    return subprocess.call(cmd, env=env)


class Telemetry:
    """Shader clock and socket power of ONE device sampled on a host thread while the timed region runs (amdsmi's Python binding;
    `rocm-smi --csv` polled when that fails) -> medians in the bench line, so that a slow box and slow code can be told apart from
    the record.  Sampling never touches the HIP streams; a failure of either source leaves the fields null with the reason."""

    def __init__(self, device_index, period=0.05):
        import threading
        self.period, self.samples, self.source, self.error = period, [], None, None
        self._stop = threading.Event()
        self._thread = None
        self._smi = None
        self._handle = None
        self._bdf = None
        try:
            from snprelate_amd import _lib
            self._bdf = _lib.device_pci(device_index).lower()
        except Exception:
            pass
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            pick = None
            for h in hs:
                try:
                    if self._bdf and amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == self._bdf:
                        pick = h
                except Exception:
                    pass
            if pick is None and len(hs) > device_index:
                pick = hs[device_index]
            if pick is None:
                raise RuntimeError("no amdsmi handle for device %d" % device_index)
            self._smi, self._handle, self.source = amdsmi, pick, "amdsmi"
            self._read()                                   # fail here, not on the thread
        except Exception as e:
            self._smi = None
            self.error = "amdsmi: %s" % str(e).strip().replace("\n", " ")[:120]
            import shutil
            if shutil.which("rocm-smi"):
                self.source, self.period = "rocm-smi", max(period, 0.25)
                self._card = device_index
            else:
                self.source = None

    def _read(self):
        """(sclk MHz, socket power W) or None"""
        if self._smi is not None:
            a = self._smi
            clk = pw = None
            try:
                m = a.amdsmi_get_gpu_metrics_info(self._handle)
                c = m.get("current_gfxclks") or m.get("current_gfxclk")
                if isinstance(c, (list, tuple)):
                    c = [x for x in c if isinstance(x, (int, float)) and 0 < x < 60000]
                    clk = sum(c) / len(c) if c else None
                elif isinstance(c, (int, float)) and 0 < c < 60000:
                    clk = float(c)
                w = m.get("current_socket_power")
                if not isinstance(w, (int, float)) or not (0 < w < 60000):
                    w = m.get("average_socket_power")
                if isinstance(w, (int, float)) and 0 < w < 60000:
                    pw = float(w)
            except Exception:
                pass
            if clk is None:
                ci = a.amdsmi_get_clock_info(self._handle, a.AmdSmiClkType.GFX)
                clk = float(ci.get("clk", ci.get("cur_clk")))
            if pw is None:
                pi = a.amdsmi_get_power_info(self._handle)
                w = pi.get("current_socket_power")
                if not isinstance(w, (int, float)) or w <= 0:
                    w = pi.get("socket_power", pi.get("average_socket_power"))
                pw = float(w)
            return clk, pw
        if self.source == "rocm-smi":
            import re
            import subprocess
            out = subprocess.run(["rocm-smi", "-d", str(self._card), "--showclocks", "--showpower", "--csv"], capture_output=True, text=True,
                                 timeout=10).stdout
            rows = [l for l in out.splitlines() if l.strip()]
            hdr = [l for l in rows if l.lower().startswith("device")]
            val = [l for l in rows if l.lower().startswith("card")]
            if not hdr or not val:
                return None
            d = dict(zip(hdr[0].split(","), val[0].split(",")))
            clk = pw = None
            for k, v in d.items():
                if "sclk" in k.lower():
                    mm = re.search(r"(\d+)\s*mhz", v, re.I)
                    if mm:
                        clk = float(mm.group(1))
                elif "power" in k.lower():
                    try:
                        pw = float(v)
                    except ValueError:
                        pass
            return clk, pw
        return None

    def _run(self):
        while not self._stop.is_set():
            try:
                r = self._read()
                if r:
                    self.samples.append((time.perf_counter(),) + tuple(r))
            except Exception as e:
                self.error = str(e)[:120]
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.source is None:
            return self
        self.samples = []
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self, t0=None, t1=None):
        """medians over the samples taken in [t0, t1] (perf_counter values; all samples when none given)"""
        import statistics as st
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=5)
            self._thread = None
        ss = [x for x in self.samples if (t0 is None or x[0] >= t0) and (t1 is None or x[0] <= t1)]
        clk = [x[1] for x in ss if x[1]]
        pw = [x[2] for x in ss if x[2]]
        return {"sclk_mhz_median": round(st.median(clk), 1) if clk else None, "sclk_mhz_min": round(min(clk), 1) if clk else None,
                "power_w_median": round(st.median(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                "telemetry_samples": len(ss), "telemetry_source": self.source,
                "telemetry_error": None if ss else (self.error or "no samples in the timed region")}


def sustained_probe(device_index, seconds_each, modes=("f16_uv_16x16x32", "f16_uv", "f16_exact_row", "fp4", "f16_zero")):
    """Register-only MFMA streams on THIS device, now (snpgpu_diag_mfma_rate): what the matrix pipe sustains under the socket power
    cap with operands shaped like the kernels' -> {mode: [TFLOP/s, implied shader MHz]}."""
    from snprelate_amd import _lib
    ids = {"f16_zero": _lib.DIAG_F16_ZERO, "f16_exact_row": _lib.DIAG_F16_EXACT_ROW, "f16_uv": _lib.DIAG_F16_UV, "fp4": _lib.DIAG_FP4,
           "f16_uv_16x16x32": _lib.DIAG_F16_UV_16X16X32, "fp4_16x16x128": _lib.DIAG_FP4_16X16X128,
           "f16_exact_row_16x16x32": _lib.DIAG_F16_EXACT_ROW_16X16X32}
    out = {}
    for m in modes:
        try:
            r, mhz = _lib.diag_mfma_rate(ids[m], seconds_each, device_index)
            out[m] = [round(r, 1), round(mhz, 0)]
        except Exception as e:
            out[m] = "failed: %s" % str(e)[:100]
    return out


def pmc_traffic(key):
    """HBM bytes per step of the dominant kernel QUOTED from the committed rocprofv3 PMC passes of this same command
    (fallback when rocprofv3 is not on PATH or the measuring child runs fail); None when no measurement for this exact
    workload/size is committed."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            tab = json.load(f)
        return tab[key] if key in tab else None
    except Exception:
        return None


def algorithmic_bytes(wl, B, n_total_snps=1000000):
    """the bytes one step must move -- the block's 2-bit genotypes once (N B / 4) plus this step's share of the
    result written once per job (8 bytes per pair for the fp64 GRM / covariance triangle, 12 for the three IBS counters,
    16 for KING's two doubles, over the ~L / B steps of the configs' 1 000 000 (GRM) / 500 000 (IBS) SNPs)."""
    n = wl["n"]
    per_pair = {"GRM_GCTA": 8, "PCA_COV": 8, "IBS": 12, "KING_ROBUST": 16, "KING_HOMO": 16}[wl["kind"]]
    L = 500000 if wl["kind"] == "IBS" else n_total_snps
    return n * B / 4.0 + per_pair * (n * (n + 1) / 2.0) / max(1.0, L / float(B))


def measure_traffic(args, kernel):
    """HBM counters of the dominant kernel measured for THIS command: two child runs of bench.py (2 steps + 1 warm-up) under
    rocprofv3 with one PMC counter each (MI355X_MICROARCH.md: separate passes, --kernel-trace only), per-dispatch sums from the
    result database, expressed per step.  `traffic` applies the guide's gfx950 correction (FETCH_SIZE reports half the bytes of
    coalesced reads: x 2; calibrated here on streaming kernels of known size, HISTORY.md 4.2; WRITE_SIZE is exact); the raw
    counter bytes ride along."""
    import sqlite3
    import subprocess
    import tempfile
    feeds = 3
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="snpgpu_pmc_")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--workload", args.workload, "--steps", "2", "--warmup", "1", "--no-sub-results", "--no-cpu-baseline", "--no-pmc", "--no-probe", "--no-telemetry"]
        if args.n:
            cmd += ["--samples", str(args.n)]
        if args.block:
            cmd += ["--block", str(args.block)]
        if args.missing is not None:
            cmd += ["--missing", str(args.missing)]
        subprocess.run(cmd, cwd=d, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True,
                       timeout=600)
        db = None
        for dp, _, files in os.walk(d):
            for fn in files:
                if fn.endswith("_results.db"):
                    db = os.path.join(dp, fn)
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        tot = 0.0
        for k, v in c.execute("select %s, sum(value) from counters_collection where counter_name = ? group by %s" % (kcol, kcol), (counter,)):
            if kernel.split("<")[0] in k:
                tot += v
        got[counter] = tot * 1024.0 / feeds           # KiB counters -> bytes per feed block
    return {"traffic": 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], "traffic_raw": got["FETCH_SIZE"] + got["WRITE_SIZE"],
            "traffic_source": "measured in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes; 2 x FETCH + WRITE, bytes per step)",
            "traffic_fetch_raw": got["FETCH_SIZE"], "traffic_write": got["WRITE_SIZE"]}


def synth_blocks(n, b, missing, count, device_index):
    """`count` consecutive 2-bit packed blocks [b][ceil(n/4)] of the seeded data set, generated on the device."""
    import torch
    from snprelate_amd import _lib
    out = []
    for i in range(count):
        t = torch.empty((b, (n + 3) // 4), dtype=torch.uint8, device=torch.device("cuda", device_index))
        _lib.synth_block(t.data_ptr(), n, i * b, b, SEED, missing=missing, device=device_index)
        out.append(t)
    return out


def _time_oracle(fn, g):
    t0 = time.perf_counter()
    fn(g)
    return time.perf_counter() - t0


def physical_cores():
    """physical cores of this host (distinct (package, core id) pairs of /proc/cpuinfo; half the hardware threads if that
    cannot be read)"""
    try:
        seen, pkg = set(), 0
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = int(line.split(":")[1])
            elif line.startswith("core id"):
                seen.add((pkg, int(line.split(":")[1])))
        if seen:
            return len(seen)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(kind):
    """the CPU oracle (C + OpenMP restatement of the reference's algorithms, kind "port") timed on this
    host's cores in the same run -- all FOUR paths (GCTA GRM, PCA covariance, IBS counts, KING-robust counters), each at
    1 thread and at the host's PHYSICAL core count with the threads bound to cores (OMP_PROC_BIND / OMP_PLACES, set before the
    OpenMP runtime starts), on bounded slices (~3 s each) of the synthetic set with 2 % missing calls -- N = 4000 samples for
    the 1-thread runs, N = 16 000 for the all-core runs (enough pairs per block to occupy 128 cores) --, plus
    configs[0] (HapMap, 279 samples x 8039 SNPs after the default filters, 1 thread).  Top-level fields = this workload's
    path at the physical core count.  The restatement tiles the pair loops over samples (cache-resident tiles, one task per
    tile) and shares GCTA's denominator walk -- serial in the reference, src/genPCA.cpp:1209-1219 -- out by rows; counts and
    sums are unchanged (tests/test_oracle_golden.py)."""
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import oracle as orc
    from oracle.synth import synth_hash_geno
    fns = {"GRM_GCTA": orc.grm_gcta, "PCA_COV": orc.pca_cov, "IBS": orc.ibs_count, "KING_ROBUST": orc.king_robust_count}
    n, L = 4000, 20000
    cores = physical_cores()
    # the all-core leg needs enough pairs per 256-SNP block to keep `cores` threads busy between the block barriers (at
    # N = 4000 a block is 0.1 ms of work for 128 cores): N = 16000 there when the host has more than 16 cores
    n_big = 16000 if cores > 16 else n
    gs = {n: synth_hash_geno(np.arange(n), 0, L, SEED, missing=0.02)}
    if n_big != n:
        gs[n_big] = synth_hash_geno(np.arange(n_big), 0, 4096, SEED, missing=0.02)
    before = orc.num_threads()
    runs, mine = [], None
    try:
        for name, fn in fns.items():
            for threads, nn in ((cores, n_big), (1, n)):
                g = gs[nn]
                orc.set_num_threads(threads)
                fn(g[:256])                                     # warm the library / thread team
                cal = _time_oracle(fn, g[:512])
                budget = 3.0
                La = int(min(g.shape[0], max(512, 512 * round(budget / max(cal, 1e-3)))))
                dt = min(_time_oracle(fn, g[:La]) for _ in range(2))
                r = dict(path=name, threads=threads, n=nn, L=La, seconds=dt, value=nn * nn * La / 2 / dt,
                         sample="synthetic %d samples x the first %d SNPs of the seeded set, 2%% missing" % (nn, La))
                runs.append(r)
                if name == kind and threads == cores:
                    mine = r
        for name in fns:
            a = [r for r in runs if r["path"] == name]
            a[0]["speedup_vs_1_thread"] = a[0]["value"] / a[1]["value"]
        # configs[0]: the bundled HapMap file through the default snpgdsGRM filters (fixture committed under tests/golden)
        try:
            from snprelate_amd.gds import open_gds, unpack_2bit_rows
            orc.set_num_threads(1)
            f = open_gds(os.path.join(ROOT, "tests", "golden", "hapmap_geno.gds"))
            chrom = f.snp_chromosome
            gh = unpack_2bit_rows(f.packed, f.n_samp)[(chrom >= 1) & (chrom <= 22)]
            valid = gh <= 2
            s, c = (gh * valid).sum(1), valid.sum(1)
            keep = (c > 0) & (s > 0) & (s < 2 * c) & ((gh.shape[1] - c) / gh.shape[1] <= 0.01)
            gh = np.ascontiguousarray(gh[keep])
            orc.grm_gcta(gh[:64])
            dth = min(_time_oracle(orc.grm_gcta, gh) for _ in range(3))
            runs.append(dict(path="GRM_GCTA", threads=1, n=gh.shape[1], L=gh.shape[0], seconds=dth,
                             value=gh.shape[1] ** 2 * gh.shape[0] / 2 / dth,
                             sample="configs[0]: snpgdsGRM GCTA on HapMap, %d samples x %d SNPs" % (gh.shape[1], gh.shape[0])))
        except Exception as e:       # the fixture is optional for the bench
            runs.append(dict(sample="configs[0] HapMap run failed: %s" % e))
    finally:
        orc.set_num_threads(before)
    # SURVEY 8(d)'s own sample as well: 4000 samples x 20000 SNPs (2 % missing) at ALL cores, this workload's path
    survey = None
    try:
        orc.set_num_threads(cores)
        fns[kind](gs[n][:256])
        dts = _time_oracle(fns[kind], gs[n])             # one pass (~10 s): the default run stays within minutes
        survey = [n, L, cores, round(dts, 3), float("%.4g" % (n * n * L / 2 / dts))]
    except Exception:
        pass
    finally:
        orc.set_num_threads(before)
    rows = [[r.get("path", "?"), r.get("threads"), r.get("n"), r.get("L"), round(r.get("seconds", 0.0), 3), float("%.4g" % r.get("value", 0.0))]
            for r in runs if "path" in r]
    return {"value": mine["value"], "unit": "SNP-pair-genotypes/s", "cores": cores, "kind": "port",
            "sample": "oracle %s (C + OpenMP restatement of the reference) on synthetic %d x %d, 2%% missing: %.1f s on %d threads (one per "
                      "physical core; host has %d hardware threads)" % (fns[kind].__name__, mine["n"], mine["L"], mine["seconds"], cores,
                                                                         os.cpu_count() or 0),
            "speedup_vs_1_thread": mine.get("speedup_vs_1_thread"),
            "survey_4000x20000_all_cores": survey,          # [n, L, threads, seconds, pair-genotypes/s]
            "runs_cols": ["path", "threads", "n", "L", "seconds", "value"], "runs": rows}       # last row: configs[0] (HapMap, 1 thread)


PROBE = {}           # filled by main(): sustained_probe() of this run (mode -> [TFLOP/s, MHz]); empty = constants of round 1
PROBE_NOTE = "constant (profiles/r01_mfma_power.txt), not measured in this run"


def sustained(mode, fallback):
    """(TFLOP/s, source) the matrix pipe sustains with operands of class `mode`: this run's probe, else the round-1 constant"""
    v = PROBE.get(mode)
    if isinstance(v, list) and v[0] > 0:
        return float(v[0]), "measured in this run: register-only MFMA stream, operand class %s (snpgpu_diag_mfma_rate)" % mode
    return float(fallback), PROBE_NOTE


def roofline(wl, world, my_pairs, B, per_launch_ms, klaunch, env, syrk_ms_per_step=None):
    """Roofline object of the dominant kernel on this rank's panel."""
    syrk = env.get("SNPGPU_SYRK", "")
    if wl["which"] == 1:
        flops = 2.0 * my_pairs * B                       # 2 flop per pair-genotype
        achieved = flops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        if syrk == "f32":
            peak, kname, extra, tkey = PEAK_F32_MFMA_TFLOPS, "syrk_mfma_kernel", {}, "grm_f32"
        else:
            # split-fp16 SYRK: exact row operand (g - c) x (hi + lo) -> 2 executed MFMA flops per algorithmic flop, for
            # blocks with and without missing calls; SNPGPU_SYRK=h3 / SNPGPU_SYRK_MISS3: hi hi' + hi lo' + lo hi' -> 3
            three = syrk == "h3" or (wl["missing"] > 0 and env.get("SNPGPU_SYRK_MISS3"))
            x1 = not three and env.get("SNPGPU_SYRK_X1", "1") != "0"      # one wave per SIMD, 256 x 256 tiles (default)
            # blocks WITHOUT missing calls: single-product kernel (SNP weight = u v in fp16, integer centres): 1 executed
            # MFMA flop per algorithmic flop
            uv = x1 and wl["missing"] == 0 and env.get("SNPGPU_SYRK_UV", "1") != "0"
            execd = 3 if three else 1 if uv else 2
            # round 6: the single-product kernel on v_mfma_f32_16x16x32_f16 unless SNPGPU_SYRK_UV16=0: syrk_uv16c_kernel (default 3: operands converted
            # from nibble words, a tile's fp32 runs walked by one work item, half its sums carried in LDS, pace-maker fetches; 2: one launch of
            # (tile, run) items, no carry), 1: syrk_uv16_kernel (operands from LDS tables)
            uv16 = uv and env.get("SNPGPU_SYRK_UV16", "3") != "0"
            uv16k = "syrk_uv16_kernel" if env.get("SNPGPU_SYRK_UV16", "3") == "1" else "syrk_uv16c_kernel"
            peak, kname = PEAK_F16_MFMA_TFLOPS, ("syrk_h3_kernel<3, false>" if three else uv16k if uv16 else "syrk_uv_kernel" if uv else
                                                 "syrk_x1_kernel" if x1 else "syrk_h3_kernel<2, true>")
            sus, sus_src = sustained("f16_uv_16x16x32" if uv16 else "f16_uv" if uv else "f16_exact_row", SUSTAINED_F16_TFLOPS[1 if uv else execd])
            extra = {"executed_per_algorithmic": execd, "executed_frac": execd * achieved / peak,
                     "sustained_peak_measured": sus, "sustained_peak_source": sus_src, "executed_frac_of_sustained": execd * achieved / sus,
                     "algorithmic_vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS}
            tkey = "grm" if wl["missing"] == 0 else "grm_missing"
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "kernel": kname, "ms_per_launch": per_launch_ms, "launches": klaunch}
        roof.update(extra)
    elif env.get("SNPGPU_PAIR_BACKEND", "") == "popcount":
        ops = POP_OPS[wl["kind"]] * my_pairs * B / 32.0  # VALU lane-ops per launch
        achieved = ops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        roof = {"bound": "valu", "achieved": achieved, "peak": PEAK_VALU_TLANEOPS, "unit": "Tlane-op/s",
                "frac": achieved / PEAK_VALU_TLANEOPS, "kernel": "pair_popcount_kernel",
                "ms_per_launch": per_launch_ms, "launches": klaunch}
        tkey = None
    else:
        slots = I8_SLOTS[wl["kind"]]
        if wl["missing"] == 0.0 and "SNPGPU_I8_NO_NOMISS" not in env:
            slots = 2                                    # blocks without missing calls: h.h' and x.x' (I8Scheme<PM_IBS_NOMISS>)
        ops = 2.0 * slots * my_pairs * B                 # multiply-adds x 2 per launch
        achieved = ops / (per_launch_ms * 1e-3) / 1e12 if per_launch_ms > 0 else 0.0
        fp4 = env.get("SNPGPU_PAIR_FP4", "1") != "0" and (slots == 2 or env.get("SNPGPU_PAIR_FP4_GENERAL", "1") != "0")
        if fp4:
            # the products on the MX-fp4 MFMA (e2m1 operands {0, +-1/2, 1, 3/2} x 2: exact); blocks without missing calls: two
            # products (pair_mfma_fp4_nomiss_kernel), blocks with missing calls: four (IBS) / five (KING-robust)
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP4_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP4_MFMA_TFLOPS,
                    "kernel": "pair_mfma_fp4_nomiss_kernel" if slots == 2 else "pair_mfma_fp4_kernel",
                    "ms_per_launch": per_launch_ms, "launches": klaunch, "products_per_pair_genotype": slots}
            sus, sus_src = sustained("fp4", SUSTAINED_FP4_TFLOPS)
            roof.update({"sustained_peak_measured": sus, "sustained_peak_source": sus_src, "frac_of_sustained": achieved / sus})
        else:
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_I8_MFMA_TOPS, "unit": "TOP/s",
                    "frac": achieved / PEAK_I8_MFMA_TOPS, "kernel": "pair_mfma_i8_kernel",
                    "ms_per_launch": per_launch_ms, "launches": klaunch, "products_per_pair_genotype": slots,
                    "sustained_peak_measured": SUSTAINED_I8_TOPS[slots == 2],
                    "frac_of_sustained": achieved / SUSTAINED_I8_TOPS[slots == 2]}
        if wl["kind"] == "KING_HOMO":
            # two kernels per step: the counters (four fp4 products) and, in blocks with missing calls, the both-missing weight sums
            # (one fp16 product per weight, syrk_uv_kernel).  The row names the one that takes longer; `step_min_ms` = the time
            # both would take at their peaks (8 fp4 flops + 4 fp16 flops per pair-genotype) -> run_workload's step_frac
            roof["kernel"] += "<PM_KING_HOMO>"
            uvk = "syrk_uv16_kernel" if env.get("SNPGPU_SYRK_UV16", "3") != "0" else "syrk_uv_kernel"       # (binary tables: always a lookup form)
            roof["kernels_ms_per_step"] = {roof["kernel"]: per_launch_ms, uvk: syrk_ms_per_step or 0.0}
            roof["step_min_ms"] = my_pairs * B * (8.0 / (PEAK_FP4_MFMA_TFLOPS * 1e12) + (4.0 / (PEAK_F16_MFMA_TFLOPS * 1e12) if syrk_ms_per_step else 0.0)) * 1e3
            if (syrk_ms_per_step or 0.0) > per_launch_ms:
                roof.update({"kernel": uvk, "ms_per_launch": syrk_ms_per_step, "peak": PEAK_F16_MFMA_TFLOPS,
                             "achieved": 4.0 * my_pairs * B / (syrk_ms_per_step * 1e-3) / 1e12,
                             "products_per_pair_genotype": 2})
                roof["frac"] = roof["achieved"] / roof["peak"]
                sus, sus_src = sustained("f16_uv_16x16x32" if uvk == "syrk_uv16_kernel" else "f16_uv", SUSTAINED_F16_TFLOPS[1])
                roof.update({"sustained_peak_measured": sus, "sustained_peak_source": sus_src, "frac_of_sustained": roof["achieved"] / sus})
        tkey = wl["kind"].lower().replace("_robust", "")
    key = "%s_n%d_b%d" % (tkey, wl["n"], B) if tkey else None
    # the main line's figure is MEASURED after the timed run (measure_traffic: rocprofv3 child passes of this command); what is set
    # here is the fallback, QUOTED from the committed PMC passes of the same command
    t = pmc_traffic(key) if (world == 1 and key) else None
    roof["traffic"] = (2.0 * t["fetch_size_raw"] + t["write_size"]) if t else None
    roof["traffic_raw"] = t["hbm_bytes_per_launch_raw"] if t else None
    roof["traffic_source"] = ("quoted from %s[%s], not measured in this run" % (TRAFFIC_FILE, key)) if t else None
    roof["algorithmic_bytes"] = algorithmic_bytes(wl, B) * (my_pairs / (wl["n"] * (wl["n"] + 1) / 2.0))
    return roof


def run_workload(wl, steps, warmup, rank, world, local, feed="device", env_over=None, dist=None, gather=False, telemetry=None,
                 shared_stats=False):
    """W warm-up steps, then K timed steps + one finalise (packed slab into a preallocated device buffer).
    Returns (result dict for rank 0, or None)."""
    import torch
    from snprelate_amd import _lib
    from snprelate_amd.dist import panel_rows, slab_range
    env_over = env_over or {}
    saved = {k: os.environ.get(k) for k in env_over}
    os.environ.update(env_over)                      # the library reads its switches when a context is created
    try:
        n, B = wl["n"], wl["b"]
        device = torch.device("cuda", local)
        bounds = panel_rows(n, world)
        r0, r1 = bounds[rank], bounds[rank + 1]
        kind = getattr(_lib, wl["kind"])
        # per-SNP statistics once per NODE (multigpu.SharedStats: every rank scans its share of a block's SNPs, one all-gather of
        # 8 bytes per SNP on the contexts' stream, no host synchronisation) instead of once per rank -- the GRM / PCA kinds only
        # (the counter kinds' pre-pass needs no statistics)
        shared = None
        if shared_stats and dist is not None and world > 1 and wl["which"] == 1 and feed == "device":
            from snprelate_amd.multigpu import SharedStats
            shared = SharedStats(rank, world, None, device, bounds, [[r] for r in range(world)])
        acc = _lib.Accumulator(kind, n, device=local, row_begin=r0, row_end=(r1 if r1 != n or r0 != 0 else 0),
                               max_block_snps=B, stream=shared.stream.cuda_stream if shared else None) if r1 > r0 else None
        blocks = synth_blocks(n, B, wl["missing"], max(1, min(steps + warmup, 3)), local)
        lo, hi = slab_range(n, r0, r1)
        n_out = {"IBS": 3, "KING_ROBUST": 2, "KING_HOMO": 2}.get(wl["kind"], 1)
        out_dtype = torch.int32 if wl["kind"] == "IBS" else torch.float64
        outs = [torch.empty(max(hi - lo, 1), dtype=out_dtype, device=device) for _ in range(n_out)]
        torch.cuda.synchronize()

        pinned = []
        if feed != "device":
            from snprelate_amd.gds import unpack_2bit_rows
            for blk in blocks[:2]:
                h = blk.cpu().numpy()
                if feed == "pinned_u8":
                    h = unpack_2bit_rows(h, n)
                pb = _lib.PinnedBuffer(h.shape)
                pb.array[:] = h
                pinned.append(pb)

        def step(i):
            if shared is not None:          # (a rank without a panel still joins the collective)
                blk = blocks[i % len(blocks)]
                st = shared.block([acc] if acc is not None else [], blk.data_ptr(), B, _lib.GENO_PACKED2, blk.shape[1])
                if acc is not None:
                    acc.feed_device_stats(blk.data_ptr(), B, st[0].data_ptr(), st[1].data_ptr())
                return
            if acc is None:
                return
            if pinned:
                pb = pinned[i % len(pinned)]
                acc.host_wait(pb)
                acc.feed_pinned(pb, B, _lib.GENO_U8 if feed == "pinned_u8" else _lib.GENO_PACKED2)
            else:
                acc.feed_device(blocks[i % len(blocks)].data_ptr(), B)

        def finalise():
            if acc is None:
                return
            if wl["kind"] == "IBS":
                acc.ibs_num(packed=True, out_ptrs=[o.data_ptr() for o in outs])
            elif wl["kind"] == "KING_ROBUST":
                acc.king_robust(packed=True, out_ptrs=(outs[0].data_ptr(), outs[1].data_ptr()))
            elif wl["kind"] == "KING_HOMO":
                acc.king_homo(packed=True, out_ptrs=(outs[0].data_ptr(), outs[1].data_ptr()))
            elif wl["kind"] == "GRM_GCTA":
                acc.grm_gcta(packed=True, out_ptr=outs[0].data_ptr())
            else:   # a panel of a sharded covariance is normalised with the all-reduced trace: raw sums here
                acc.pca_cov(packed=True, normalize=False, out_ptr=outs[0].data_ptr())

        def fence():
            if acc is not None:
                acc.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for i in range(warmup):
            step(i)
        fence()
        if acc is not None:
            acc.set_timing(True)
        if telemetry is not None:
            telemetry.start()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        if acc is not None:
            acc.sync()
        t_steps = time.perf_counter() - t0
        finalise()
        fence()
        dt = time.perf_counter() - t0
        tele = telemetry.stop(t0, t0 + dt) if telemetry is not None else None
        kms, klaunch = acc.get_timing(wl["which"]) if acc is not None else (0.0, 0)
        syrk_ms_per_step = None
        if acc is not None and wl["kind"] == "KING_HOMO":
            syrk_ms_per_step = acc.get_timing(1)[0] / max(steps, 1)
        if acc is not None:
            acc.set_timing(False)
        rank_pairs = rank_kernel_ms = None
        if dist is not None:
            t = torch.tensor([dt, t_steps], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, t_steps = float(t[0].item()), float(t[1].item())
            # every rank's share of the triangle and its pair-kernel time per step (how well the time-balanced plan held)
            mine = torch.zeros(4 * world, device=device, dtype=torch.float64)
            mine[4 * rank] = (r1 - r0) * n - (r0 + r1 - 1) * (r1 - r0) / 2.0
            mine[4 * rank + 1] = kms / max(klaunch, 1)
            mine[4 * rank + 2] = (tele or {}).get("sclk_mhz_median") or 0.0
            mine[4 * rank + 3] = (tele or {}).get("power_w_median") or 0.0
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            rank_pairs = [float(x) for x in mine[0::4].tolist()]
            rank_kernel_ms = [float(x) for x in mine[1::4].tolist()]
            if tele is not None:
                tele["rank_sclk_mhz_median"] = [float(x) for x in mine[2::4].tolist()]
                tele["rank_power_w_median"] = [float(x) for x in mine[3::4].tolist()]

        gather_ms = None
        if dist is not None and gather:
            # the north_star's final exchange: RCCL gather of the finished slabs on rank 0 (outside `value`)
            try:
                from snprelate_amd.dist import gather_slabs
                torch.cuda.synchronize(); dist.barrier()
                t1 = time.perf_counter()
                full = gather_slabs(outs[0][: hi - lo], n, bounds, rank, world)
                torch.cuda.synchronize(); dist.barrier()
                gather_ms = (time.perf_counter() - t1) * 1e3
                del full
            except Exception as e:           # never let the optional leg take the bench line down
                gather_ms = "failed: %s" % str(e)[:200]

        res = None
        if rank == 0:
            my_pairs = (r1 - r0) * n - (r0 + r1 - 1) * (r1 - r0) / 2.0
            value = (n * n / 2.0) * B * steps / dt
            res = {"value": value, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                   "finalize_ms": (dt - t_steps) * 1e3, "steps_only_ms_per_step": t_steps / steps * 1e3,
                   "gather_ms": gather_ms, "rank_pairs": rank_pairs, "rank_kernel_ms": rank_kernel_ms, "telemetry": tele,
                   "shared_stats": shared is not None,
                   "roofline": roofline(wl, world, my_pairs, B, kms / max(klaunch, 1), klaunch, os.environ, syrk_ms_per_step)}
            if "step_min_ms" in res["roofline"]:
                res["roofline"]["step_frac"] = res["roofline"]["step_min_ms"] / res["steps_only_ms_per_step"]
        if acc is not None:
            acc.close()
        del outs, blocks
        torch.cuda.empty_cache()
        return res
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def dtype_of(wl, env):
    if wl["which"] == 1:
        if env.get("SNPGPU_SYRK", "") == "f32":
            return "f32 (fp32 MFMA, fp64 panel sums)"
        if (wl["missing"] == 0 and env.get("SNPGPU_SYRK", "") != "h3" and env.get("SNPGPU_SYRK_UV", "1") != "0"
                and env.get("SNPGPU_SYRK_X1", "1") != "0"):
            runs = "runs of 8192" if env.get("SNPGPU_H3_PROMOTE") == "8192" else "runs of 32768" if env.get("SNPGPU_SYRK_FAST", "0") not in ("", "0") \
                else "runs of <= 11264"
            return "f16 (exact operands: integer-centred genotype x fp16 factor of the SNP weight; fp32 MFMA accumulate in %s slots, fp64 panel sums)" % runs
        return "f16 (hi/lo split column operand = 22 bits, exact row operand; fp32 MFMA accumulate, fp64 panel sums)"
    if env.get("SNPGPU_PAIR_BACKEND", "") == "popcount":
        return "u32 (wavefront bit-ops)"
    if env.get("SNPGPU_PAIR_FP4", "1") != "0" and ((wl["missing"] == 0 and "SNPGPU_I8_NO_NOMISS" not in env)
                                                    or env.get("SNPGPU_PAIR_FP4_GENERAL", "1") != "0"):
        return "fp4 e2m1 (MX-fp4 MFMA with power-of-two scales, operands {0, +-1/2, 1, 3/2} x 2; fp32 accumulate of integers < 2^24: exact)"
    return "i8 (int8 MFMA, int32 accumulate: exact)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 8 (grm, pca) / 50 (ibs, king)")
    ap.add_argument("--warmup", type=int, default=None, help="default 2 (grm, pca) / 20 (ibs, king: ms-scale "
                    "steps, the first ~10 ms on an idle GPU run at a lower clock)")
    ap.add_argument("--workload", default="grm", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", "--n", dest="n", type=int, default=0, help="override the number of samples (not the named config)")
    ap.add_argument("--block", type=int, default=0, help="override SNPs per step")
    ap.add_argument("--missing", type=float, default=None, help="override the missing-call rate of the synthetic data")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-results", action="store_true", help="skip the short ibs / king / grm_f32 / grm_missing runs")
    ap.add_argument("--gather", action="store_true", help="(the default for --gpus N > 1 since round 6; kept for old command lines)")
    ap.add_argument("--no-gather", action="store_true", help="multi-GPU: skip the final RCCL gather of the slabs on rank 0 (north_star's "
                    "\"final RCCL gather over xGMI\": timed AFTER the timed region, reported as config.gather_ms, never part of `value`; "
                    "skipped by itself when the packed triangle would not fit next to rank 0's panel, N > 100 000)")
    ap.add_argument("--shared-stats", default="auto", choices=["auto", "on", "off"], help="multi-GPU GRM / PCA: per-SNP statistics of a block "
                    "computed once per node (each rank its share of the SNPs + one all-gather of 8 bytes per SNP on the contexts' stream) "
                    "instead of once per rank; bit-identical results.  auto = on from 4 ranks")
    ap.add_argument("--no-probe", action="store_true", help="skip the sustained-MFMA-rate probe (about 5 s) that fills "
                    "roofline.sustained_peak_measured and config.sustained_probe")
    ap.add_argument("--no-telemetry", action="store_true", help="do not sample shader clock / socket power during the timed region")
    ap.add_argument("--pmc", action="store_true", help="(default when rocprofv3 is on PATH) MEASURE roofline.traffic: re-runs this "
                    "workload twice under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, no other trace "
                    "domain) after the timed run and sums the dominant kernel's counters per step (adds about a minute)")
    ap.add_argument("--no-pmc", action="store_true", help="quote roofline.traffic from the committed profile instead of measuring it")
    ap.add_argument("--stamp", action="store_true", help="print the source stamp of this tree and exit")
    ap.add_argument("--details", default=None, metavar="FILE", help="write the long form (every sub-run with its whole roofline object) to FILE; "
                    "the printed line carries one `summary` row per sub-run")
    ap.add_argument("--feed", default="device", choices=["device", "pinned_u8", "pinned_2bit"],
                    help="device: blocks resident in HBM (the metric). pinned_*: blocks come from page-locked host "
                         "memory through snpgpu_feed(SNPGPU_HOST_PINNED) -- the PCIe-inclusive rate of the R reader path")
    args = ap.parse_args()
    if args.stamp:
        print(source_stamp())
        return 0
    # --gpus N > 1 as a plain command (no torch.distributed environment): become the launcher
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args.gpus)
    quick = args.workload in ("ibs", "king", "king_homo")
    if args.steps is None:
        args.steps = 50 if quick else 8
    if args.warmup is None:
        args.warmup = 20 if quick else 2

    import torch

    wl = dict(WORKLOADS[args.workload])
    overridden = bool(args.n or args.block or args.missing is not None)
    if args.missing is not None:
        wl["missing"] = float(args.missing)
    if args.n:
        wl["n"] = args.n
        wl["name"] += " [OVERRIDE n=%d]" % args.n
    if args.block:
        wl["b"] = args.block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_multiproc.py runs 2 ranks on the single test GPU over gloo)
    backend = os.environ.get("SNPGPU_BENCH_BACKEND", "nccl")
    if "SNPGPU_BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["SNPGPU_BENCH_FORCE_DEVICE"])
    if world != args.gpus:
        # a record whose n_gpus is not what the command asked for is worse than no record: refuse before any rendezvous
        print("bench.py: --gpus %d but the launch environment has WORLD_SIZE=%d (rank %d): refusing to run" % (args.gpus, world, rank),
              file=sys.stderr)
        return 2
    dist = None
    # SNPGPU_BENCH_FORCE_DIST=1 (tests): initialise the process group under torch.distributed.run even with ONE rank, so that the
    # RCCL code path -- communicator set-up, barrier, all-reduce of the timings, the slab gather -- executes on a one-GPU box
    if world > 1 or (os.environ.get("SNPGPU_BENCH_FORCE_DIST") and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(torch.device("cuda", local))

    # what the collective library really saw: the number of ranks that joined an all-reduce of ones on the
    # backend in use, and every rank's device (HIP ordinal, PCI address, name) -- eight ranks on eight DISTINCT devices, or not
    comm = None
    if dist is not None:
        ones = torch.ones(1, device=torch.device("cuda", local), dtype=torch.float32)
        dist.all_reduce(ones)
        from snprelate_amd import _lib as _l
        pr = torch.cuda.get_device_properties(local)
        me = "%d@%s %s" % (local, _l.device_pci(local), getattr(pr, "gcnArchName", pr.name).split(":")[0])
        devs = [None] * world
        dist.all_gather_object(devs, me)
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                pass
        comm = {"rccl_ranks": int(round(float(ones.item()))), "collective_backend": "nccl (= RCCL%s)" % (" " + ver if ver else "") if backend == "nccl"
                else backend + " (test hook SNPGPU_BENCH_BACKEND; the driver's runs use nccl = RCCL)",
                "rank_devices": devs, "distinct_devices": len(set(d.split(" ")[0].split("@")[1] for d in devs)),
                "self_launched": bool(os.environ.get("SNPGPU_BENCH_SELF_LAUNCHED"))}

    if not args.no_probe and not overridden or os.environ.get("SNPGPU_BENCH_PROBE"):
        # every rank at once (the node's power budget is shared); N = 1: 1.5 s per operand class, N > 1: the headline's class only
        PROBE.update(sustained_probe(local, 1.5 if world == 1 else 1.0, ("f16_uv_16x16x32", "f16_uv", "f16_exact_row", "fp4", "f16_zero") if world == 1 else ("f16_uv_16x16x32",)))
    tele = None if args.no_telemetry else Telemetry(local)
    do_gather = dist is not None and not args.no_gather and wl["n"] <= 100000
    main_res = run_workload(wl, args.steps, args.warmup, rank, world, local, feed=args.feed, dist=dist, gather=do_gather, telemetry=tele,
                            shared_stats=args.shared_stats == "on" or (args.shared_stats == "auto" and world >= 4))
    out = None
    if rank == 0:
        out = {
            "metric": "SNP-pair-genotypes/sec (N^2*L/2/t)", "value": main_res["value"], "unit": "SNP-pair-genotypes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtype_of(wl, os.environ), "data": "synthetic",
            "config": {"workload": wl["name"], "n_samples": wl["n"], "snps_per_step": wl["b"],
                       "missing_rate": wl["missing"], "parallelism": "row-panel x%d" % world, "feed": args.feed,
                       "timed_region": "K steps + one finalise (packed triangle, device buffer)",
                       "finalize_ms": main_res["finalize_ms"],
                       "steps_only_ms_per_step": main_res["steps_only_ms_per_step"],
                       "gather_ms": main_res["gather_ms"], "rank_pairs": main_res["rank_pairs"],
                       "rank_kernel_ms_per_step": main_res["rank_kernel_ms"], "shared_stats": main_res["shared_stats"]},
            "roofline": main_res["roofline"],
        }
        if comm:
            out["config"].update(comm)
        if main_res.get("telemetry"):
            out["config"].update(main_res["telemetry"])
        if PROBE:
            out["config"]["sustained_probe"] = dict(PROBE, cols=["TFLOP/s", "implied_sclk_mhz"])
    # short runs of the other configurations, so that the driver-timed record also covers configs[1], KING, the north_star's
    # fp32 tile, the real-data (missing calls) path and the feed-inclusive rate.  The bench line stays SHORT (< 8 KB: the driver
    # keeps an 8 KB tail): per run one row of `summary` -- the LAST key of the line -- [value, ms_per_step, roofline frac of its
    # dominant kernel, that kernel, its ms per step]; the long form of every run goes to --details FILE.
    details = {}
    summary = {}
    if world == 1 and not args.no_sub_results and not overridden and args.workload == "grm" and args.feed == "device":
        plan = [("ibs", WORKLOADS["ibs"], 40, 20, {}, "device"), ("ibs_missing_0.02", dict(WORKLOADS["ibs"], missing=0.02), 40, 20, {}, "device"),
                ("king", WORKLOADS["king"], 40, 20, {}, "device"),
                ("king_missing_0", dict(WORKLOADS["king"], missing=0.0), 40, 20, {}, "device"),
                ("king_homo", WORKLOADS["king_homo"], 20, 10, {}, "device"),
                ("grm_missing_0.02", dict(WORKLOADS["grm"], missing=0.02), 6, 2, {}, "device"),
                # SURVEY 8(d) "end-to-end incl. feed": the same GRM steps with every block coming from page-locked host memory
                # through snpgpu_feed(SNPGPU_HOST_PINNED) (2-bit rows, two pinned buffers, copies under the previous block's kernels)
                ("grm_feed_pinned_2bit", WORKLOADS["grm"], 6, 2, {}, "pinned_2bit"),
                ("grm_exact_row", WORKLOADS["grm"], 4, 1, {"SNPGPU_SYRK_UV": "0"}, "device"),
                ("grm_run8192", WORKLOADS["grm"], 6, 2, {"SNPGPU_H3_PROMOTE": "8192"}, "device"),
                ("grm_fast", WORKLOADS["grm"], 6, 2, {"SNPGPU_SYRK_FAST": "1"}, "device"),
                ("grm_f32", WORKLOADS["grm"], 2, 1, {"SNPGPU_SYRK": "f32"}, "device")]
        notes = {"grm_exact_row": "SNPGPU_SYRK_UV=0: exact-row kernel for every block", "grm_run8192": "SNPGPU_H3_PROMOTE=8192: eight fp32 runs per block",
                 "grm_fast": "SNPGPU_SYRK_FAST=1: round 2's kernels (one 32768-SNP run, one weight target; 1.6e-5 instead of < 1e-5)",
                 "grm_f32": "SNPGPU_SYRK=f32: north_star's fp32-MFMA tile", "grm_feed_pinned_2bit": "blocks fed from pinned host memory (PCIe inclusive)"}
        for name, w, k, wu, env_over, feed in plan:
            try:
                r = run_workload(dict(w), k, wu, 0, 1, local, env_over=env_over, feed=feed)
                envv = dict(os.environ, **env_over)
                roof = r["roofline"]
                details[name] = {"value": r["value"], "unit": "SNP-pair-genotypes/s", "ms_per_step": r["ms_per_step"],
                                 "steps": k, "warmup": wu, "finalize_ms": r["finalize_ms"], "dtype": dtype_of(w, envv),
                                 "workload": w["name"], "missing_rate": w["missing"], "feed": feed, "note": notes.get(name), "roofline": roof}
                if w["which"] == 1:      # the whole step (pre-pass, both-missing counts, every launch) against the same peak
                    details[name]["step_frac_of_peak"] = (w["n"] ** 2 * w["b"] / (r["ms_per_step"] * 1e-3) / 1e12) / roof["peak"]
                # KING-homo: two kernels per step -> the STEP's fraction (time both would take at their peaks / step time), the kernel
                # that takes longer and its ms
                summary[name] = [float("%.4g" % r["value"]), round(r["ms_per_step"], 3), round(roof.get("step_frac", roof["frac"]), 4),
                                 roof["kernel"].split(" ")[0], round(roof["ms_per_launch"], 3)]
            except Exception as e:
                details[name] = {"error": str(e)[:300]}
                summary[name] = ["error", str(e)[:80]]
        # the second headline: data WITH missing calls (array data, any non-imputed call set) take the exact-row kernel + GCTA's
        # both-missing contraction; its figure rides at the top level as well
        m = details.get("grm_missing_0.02", {})
        if "value" in m:
            out["real_data_path"] = {"workload": "configs[2] with 2% missing calls", "value": m["value"], "unit": m["unit"],
                                     "ms_per_step": m["ms_per_step"], "step_frac_of_peak": m.get("step_frac_of_peak"),
                                     "kernel": m["roofline"]["kernel"]}
        fd = details.get("grm_feed_pinned_2bit", {})
        if "value" in fd:
            out["feed_inclusive"] = {"workload": "configs[2], blocks from pinned host memory via snpgpu_feed(SNPGPU_HOST_PINNED), 2-bit rows",
                                     "value": fd["value"], "unit": fd["unit"], "ms_per_step": fd["ms_per_step"],
                                     "vs_resident": fd["value"] / out["value"]}
    if rank == 0 and world == 1 and not args.no_pmc and args.feed == "device":
        import shutil
        if args.pmc or shutil.which("rocprofv3"):
            try:
                out["roofline"].update(measure_traffic(args, out["roofline"]["kernel"]))
            except Exception as e:          # the quoted figure stays
                out["roofline"]["traffic_measure_error"] = str(e)[:200]
    if rank == 0:
        r = out["roofline"]
        if r.get("traffic") and r.get("algorithmic_bytes"):
            r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes"]
        out["source_stamp"] = source_stamp()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl["kind"])
        if args.details:
            with open(args.details, "w") as f:
                json.dump(dict(out, sub_results=details), f, indent=1)
        if summary:
            out["summary_cols"] = ["value", "ms_per_step", "roofline_frac", "kernel", "kernel_ms_per_step"]
            out["summary"] = summary                     # LAST key: what the driver's 8 KB tail must hold
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
