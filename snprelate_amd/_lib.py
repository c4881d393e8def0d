"""ctypes binding of libsnpgpu.so (include/snpgpu.h).

There is no CPU fallback: if the HIP library is missing, fails to load, or no
MI355X is visible, every compute entry point raises ``SnpGpuError``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNPGPU_LIB: another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("SNPGPU_LIB") or os.path.join(_HERE, "libsnpgpu.so")

# enums of include/snpgpu.h
IBS, KING_ROBUST, KING_HOMO, GRM_GCTA, PCA_COV, EIGMIX, INDIV_BETA = 1, 2, 3, 4, 5, 6, 7
GENO_U8, GENO_PACKED2 = 0, 1
HOST, DEVICE, HOST_PINNED = 0, 1, 2

EXPORTS = [
    "snpgpu_abi_version", "snpgpu_last_error", "snpgpu_device_count",
    "snpgpu_create", "snpgpu_destroy", "snpgpu_feed", "snpgpu_sync", "snpgpu_counts",
    "snpgpu_host_alloc", "snpgpu_host_free", "snpgpu_host_wait",
    "snpgpu_slab_size", "snpgpu_set_timing", "snpgpu_get_timing", "snpgpu_ibs_num", "snpgpu_ibs_ave", "snpgpu_king_robust_counts",
    "snpgpu_king_robust", "snpgpu_king_homo", "snpgpu_grm_gcta", "snpgpu_pca_cov", "snpgpu_panel_entries", "snpgpu_block_stats", "snpgpu_feed_stats",
    "snpgpu_ibd_mom", "snpgpu_eigmix", "snpgpu_indiv_beta", "snpgpu_gnrIBD_PLINK", "snpgpu_gnrIBD_Beta",
    "snpgpu_gnrGRM_avg_val", "snpgpu_gnrEigMix",
    "snpgpu_pca_eigen", "snpgpu_pca_panel_matmul", "snpgpu_pca_panel_matmul_f32", "snpgpu_pca_panel_trace", "snpgpu_ws_set_geno", "snpgpu_ws_sel_snp_base",
    "snpgpu_ws_get_geno_dim", "snpgpu_ws_snp_rate_freq", "snpgpu_ws_clear",
    "snpgpu_gnrIBSNum", "snpgpu_gnrIBSAve", "snpgpu_gnrIBD_KING_Robust",
    "snpgpu_gnrIBD_KING_Homo", "snpgpu_gnrGRM", "snpgpu_gnrPCA",
    "snpgpu_proj_create", "snpgpu_proj_destroy", "snpgpu_proj_sync", "snpgpu_proj_set_eigvec", "snpgpu_proj_snp_corr",
    "snpgpu_proj_snp_loading", "snpgpu_proj_samp_loading_feed", "snpgpu_proj_samp_loading",
    "snpgpu_gnrPCACorr", "snpgpu_gnrPCASNPLoading", "snpgpu_gnrPCASampLoading",
    "snpgpu_proj_samp_loading_reset", "snpgpu_gnrPCA_randomized",
    "snpgpu_proj_snp_loading_ext", "snpgpu_gnrEigMixSNPLoading", "snpgpu_gnrEigMixSampLoading",
    "snpgpu_gnrGRMMerge", "snpgpu_synth_block", "snpgpu_ws_sel_snp_base_ex",
    "snpgpu_finalize_inplace", "snpgpu_panels_topk_eigen",
    "snpgpu_multi_create", "snpgpu_multi_destroy", "snpgpu_multi_info", "snpgpu_multi_comm_selftest", "snpgpu_multi_panel", "snpgpu_multi_feed",
    "snpgpu_multi_host_wait", "snpgpu_multi_sync", "snpgpu_multi_counts", "snpgpu_multi_ibs_num", "snpgpu_multi_ibs_ave",
    "snpgpu_multi_king_robust", "snpgpu_multi_king_robust_counts", "snpgpu_multi_king_homo", "snpgpu_multi_grm_gcta",
    "snpgpu_multi_eigmix", "snpgpu_multi_pca_trace", "snpgpu_multi_pca_cov", "snpgpu_multi_finalize_inplace",
    "snpgpu_multi_topk_eigen", "snpgpu_diag_mfma_rate", "snpgpu_diag_device_pci", "snpgpu_multi_get_status",
]


class SnpGpuError(RuntimeError):
    pass


class Opts(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("bayesian", ctypes.c_int32),
                ("row_begin", ctypes.c_int64), ("row_end", ctypes.c_int64),
                ("max_block_snps", ctypes.c_int64), ("stream", ctypes.c_void_p)]


REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)


class EigOpts(ctypes.Structure):       # snpgpu_eig_opts
    _fields_ = [("tol", ctypes.c_double), ("block", ctypes.c_int32), ("depth", ctypes.c_int32),
                ("max_restarts", ctypes.c_int32), ("seed", ctypes.c_uint32), ("y_buf", ctypes.c_void_p),
                ("reduce", REDUCE_FN), ("user", ctypes.c_void_p), ("fp32_until", ctypes.c_double)]


class MultiOpts(ctypes.Structure):     # snpgpu_multi_opts
    _fields_ = [("devices", ctypes.POINTER(ctypes.c_int32)), ("n_devices", ctypes.c_int32),
                ("panels_per_device", ctypes.c_int32), ("n_passes", ctypes.c_int32), ("pass_", ctypes.c_int32)]


class MultiStatus(ctypes.Structure):   # snpgpu_multi_status
    _fields_ = [(k, ctypes.c_int32) for k in ("n_devices", "n_distinct_devices", "n_panels", "panels_per_device", "uses_rccl", "peer_pairs",
                                              "peer_pairs_enabled", "selftest_comm", "selftest_feed", "selftest_gather")] + \
               [("reserved", ctypes.c_int32 * 6)]


class EigInfo(ctypes.Structure):       # snpgpu_eig_info
    _fields_ = [("restarts", ctypes.c_int32), ("matmuls", ctypes.c_int32), ("block", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("max_rel_residual", ctypes.c_double), ("matmuls_fp32", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


_lib = None


def build(force=False):
    """Compile libsnpgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP/HSA runtime; if it is initialised AFTER the system runtime
    # that libsnpgpu.so links against, torch reports "No HIP GPUs are available".  The other order
    # works, so when torch is installed let it probe the devices first (plumbing only: the library
    # itself has no torch dependency and runs without it, e.g. under R).
    try:
        import torch
        torch.cuda.is_available()
    except Exception:  # pragma: no cover
        pass
    if not os.path.exists(LIB_PATH):
        raise SnpGpuError("libsnpgpu.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "there is no CPU fallback")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SnpGpuError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i64, c_int, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    L.snpgpu_abi_version.restype = c_int
    L.snpgpu_last_error.restype = ctypes.c_char_p
    L.snpgpu_device_count.argtypes = [ctypes.POINTER(c_int)]
    L.snpgpu_diag_mfma_rate.argtypes = [c_int, c_int, dbl, ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    L.snpgpu_diag_device_pci.argtypes = [c_int, ctypes.c_char_p, c_int]
    L.snpgpu_synth_block.argtypes = [vp, i64, i64, i64, ctypes.c_uint32, dbl, c_int, c_int, c_int, vp]
    L.snpgpu_create.argtypes = [c_int, i64, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.snpgpu_destroy.argtypes = [vp]
    L.snpgpu_feed.argtypes = [vp, vp, i64, c_int, c_int]
    L.snpgpu_sync.argtypes = [vp]
    L.snpgpu_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    L.snpgpu_host_free.argtypes = [vp]
    L.snpgpu_host_wait.argtypes = [vp, vp]
    L.snpgpu_block_stats.argtypes = [vp, vp, i64, c_int, vp, vp]
    L.snpgpu_feed_stats.argtypes = [vp, vp, i64, c_int, c_int, vp, vp]
    L.snpgpu_counts.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.snpgpu_set_timing.argtypes = [vp, c_int]
    L.snpgpu_get_timing.argtypes = [vp, c_int, ctypes.POINTER(dbl), ctypes.POINTER(i64)]
    L.snpgpu_slab_size.argtypes = [vp]
    L.snpgpu_slab_size.restype = i64
    L.snpgpu_ibs_num.argtypes = [vp, vp, vp, vp, c_int, c_int]
    L.snpgpu_ibs_ave.argtypes = [vp, vp, c_int, c_int]
    L.snpgpu_king_robust_counts.argtypes = [vp, vp, c_int]
    L.snpgpu_king_robust.argtypes = [vp, vp, vp, vp, c_int, c_int]
    L.snpgpu_king_homo.argtypes = [vp, vp, vp, c_int, c_int]
    L.snpgpu_grm_gcta.argtypes = [vp, vp, c_int, c_int]
    L.snpgpu_pca_cov.argtypes = [vp, vp, c_int, c_int, dbl, ctypes.POINTER(dbl), c_int]
    L.snpgpu_pca_eigen.argtypes = [vp, c_int, vp, vp, c_int]
    L.snpgpu_ibd_mom.argtypes = [vp, vp, c_int, vp, vp, c_int, c_int]
    L.snpgpu_eigmix.argtypes = [vp, c_int, dbl, vp, c_int, c_int]
    L.snpgpu_indiv_beta.argtypes = [vp, c_int, vp, ctypes.POINTER(dbl), c_int, c_int]
    L.snpgpu_gnrIBD_PLINK.argtypes = [c_int, vp, c_int, c_int, c_int, vp, vp, vp]
    L.snpgpu_gnrIBD_Beta.argtypes = [c_int, c_int, c_int, c_int, vp, ctypes.POINTER(dbl)]
    L.snpgpu_gnrGRM_avg_val.argtypes = [ctypes.POINTER(dbl)]
    L.snpgpu_gnrEigMix.argtypes = [c_int, c_int, c_int, c_int, vp, vp, vp, vp]
    L.snpgpu_pca_panel_matmul.argtypes = [vp, dbl, vp, c_int, vp]
    L.snpgpu_pca_panel_matmul_f32.argtypes = [vp, dbl, vp, c_int, vp]
    L.snpgpu_pca_panel_trace.argtypes = [vp, ctypes.POINTER(dbl)]
    L.snpgpu_finalize_inplace.argtypes = [vp, c_int, dbl]
    L.snpgpu_panel_entries.argtypes = [vp, vp, vp, i64, vp]
    L.snpgpu_panels_topk_eigen.argtypes = [ctypes.POINTER(vp), c_int, dbl, c_int, ctypes.POINTER(EigOpts), vp, vp, c_int,
                                           ctypes.POINTER(EigInfo)]
    L.snpgpu_multi_create.argtypes = [c_int, i64, ctypes.POINTER(Opts), ctypes.POINTER(MultiOpts), ctypes.POINTER(vp)]
    L.snpgpu_multi_destroy.argtypes = [vp]
    L.snpgpu_multi_info.argtypes = [vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.snpgpu_multi_comm_selftest.argtypes = [vp, ctypes.POINTER(c_int)]
    L.snpgpu_multi_get_status.argtypes = [vp, ctypes.POINTER(MultiStatus)]
    L.snpgpu_multi_panel.argtypes = [vp, c_int, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(c_int)]
    L.snpgpu_multi_feed.argtypes = [vp, vp, i64, c_int, c_int]
    L.snpgpu_multi_host_wait.argtypes = [vp, vp]
    L.snpgpu_multi_sync.argtypes = [vp]
    L.snpgpu_multi_counts.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.snpgpu_multi_ibs_num.argtypes = [vp, vp, vp, vp, c_int]
    L.snpgpu_multi_ibs_ave.argtypes = [vp, vp, c_int]
    L.snpgpu_multi_king_robust.argtypes = [vp, vp, vp, vp, c_int]
    L.snpgpu_multi_king_robust_counts.argtypes = [vp, vp, c_int]
    L.snpgpu_multi_king_homo.argtypes = [vp, vp, vp, c_int]
    L.snpgpu_multi_grm_gcta.argtypes = [vp, vp, c_int]
    L.snpgpu_multi_eigmix.argtypes = [vp, c_int, dbl, vp, c_int]
    L.snpgpu_multi_pca_trace.argtypes = [vp, ctypes.POINTER(dbl)]
    L.snpgpu_multi_pca_cov.argtypes = [vp, vp, c_int, ctypes.POINTER(dbl), c_int]
    L.snpgpu_multi_finalize_inplace.argtypes = [vp, c_int, dbl]
    L.snpgpu_multi_topk_eigen.argtypes = [vp, dbl, c_int, ctypes.POINTER(EigOpts), vp, vp, c_int, ctypes.POINTER(EigInfo)]
    L.snpgpu_ws_set_geno.argtypes = [vp, i64, i64, c_int, c_int]
    L.snpgpu_ws_sel_snp_base.argtypes = [c_int, dbl, dbl, ctypes.POINTER(ctypes.c_int32), vp]
    L.snpgpu_ws_sel_snp_base_ex.argtypes = [vp, c_int, dbl, dbl, ctypes.POINTER(ctypes.c_int32), vp]
    L.snpgpu_ws_get_geno_dim.argtypes = [ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.snpgpu_ws_snp_rate_freq.argtypes = [vp, vp, vp]
    L.snpgpu_gnrIBSNum.argtypes = [c_int, c_int, vp, vp, vp]
    L.snpgpu_gnrIBSAve.argtypes = [c_int, c_int, c_int, vp]
    L.snpgpu_gnrIBD_KING_Robust.argtypes = [vp, c_int, c_int, c_int, vp, vp]
    L.snpgpu_gnrIBD_KING_Homo.argtypes = [c_int, c_int, c_int, vp, vp]
    L.snpgpu_gnrGRM.argtypes = [c_int, ctypes.c_char_p, c_int, c_int, vp]
    L.snpgpu_gnrGRMMerge.argtypes = [c_int, i64, ctypes.POINTER(vp), ctypes.c_char_p, vp, vp, vp, c_int]
    L.snpgpu_gnrPCA.argtypes = [c_int, c_int, c_int, c_int, ctypes.POINTER(dbl), vp, vp, vp,
                                ctypes.POINTER(dbl)]
    L.snpgpu_proj_create.argtypes = [i64, c_int, ctypes.POINTER(Opts), ctypes.POINTER(vp)]
    L.snpgpu_proj_destroy.argtypes = [vp]
    L.snpgpu_proj_sync.argtypes = [vp]
    L.snpgpu_proj_set_eigvec.argtypes = [vp, vp, c_int]
    L.snpgpu_proj_snp_corr.argtypes = [vp, vp, i64, c_int, c_int, vp, c_int]
    L.snpgpu_proj_snp_loading.argtypes = [vp, vp, i64, c_int, c_int, c_int, vp, vp, vp, c_int]
    L.snpgpu_proj_samp_loading_feed.argtypes = [vp, vp, i64, c_int, c_int, vp, vp, vp, c_int]
    L.snpgpu_proj_samp_loading.argtypes = [vp, vp, c_int]
    L.snpgpu_proj_snp_loading_ext.argtypes = [vp, vp, i64, c_int, c_int, vp, vp, c_int, vp, c_int]
    L.snpgpu_gnrEigMixSNPLoading.argtypes = [vp, vp, c_int, vp, c_int, c_int, vp]
    L.snpgpu_gnrEigMixSampLoading.argtypes = [c_int, vp, vp, c_int, c_int, vp]
    L.snpgpu_proj_samp_loading_reset.argtypes = [vp]
    L.snpgpu_gnrPCA_randomized.argtypes = [c_int, c_int, c_int, vp, c_int, c_int, vp, vp, ctypes.POINTER(dbl)]
    L.snpgpu_gnrPCACorr.argtypes = [c_int, vp, c_int, c_int, vp]
    L.snpgpu_gnrPCASNPLoading.argtypes = [vp, vp, c_int, dbl, c_int, c_int, c_int, vp, vp, vp]
    L.snpgpu_gnrPCASampLoading.argtypes = [c_int, vp, vp, vp, c_int, c_int, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise SnpGpuError(lib().snpgpu_last_error().decode("utf-8", "replace"))


def device_count():
    n = ctypes.c_int(0)
    check(lib().snpgpu_device_count(ctypes.byref(n)))
    return n.value


DIAG_F16_ZERO, DIAG_F16_EXACT_ROW, DIAG_F16_UV, DIAG_FP4, DIAG_F16_UV_16X16X32, DIAG_FP4_16X16X128, DIAG_F16_EXACT_ROW_16X16X32 = 0, 1, 2, 3, 4, 5, 6


def diag_mfma_rate(mode=DIAG_F16_UV, seconds=2.0, device=0):
    """(TFLOP/s, implied shader MHz) a register-only stream of the MFMA instruction of `mode` sustains on `device` right now
    (snpgpu_diag_mfma_rate: the power-capped rate the SYRK / pair-counter kernels run against)."""
    L = lib()
    r, mhz = ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(L.snpgpu_diag_mfma_rate(int(device), int(mode), float(seconds), ctypes.byref(r), ctypes.byref(mhz)))
    return r.value, mhz.value


def device_pci(device=0):
    """PCI address of HIP device `device` ("0000:05:00.0")"""
    buf = ctypes.create_string_buffer(64)
    check(lib().snpgpu_diag_device_pci(int(device), buf, 64))
    return buf.value.decode()


def synth_block(dev_ptr, n_samp, snp_begin, n_snp, seed, missing=0.0, spectrum=0, special=False, device=0, stream=None):
    """Fill DEVICE memory at dev_ptr with SNPs [snp_begin, snp_begin + n_snp) of the seeded synthetic data set as
    2-bit rows [n_snp][ceil(n_samp/4)] (snpgpu_synth_block; counter-based, see oracle/synth.py for the CPU twin)."""
    check(lib().snpgpu_synth_block(ctypes.c_void_p(int(dev_ptr)), int(n_samp), int(snp_begin), int(n_snp),
                                   ctypes.c_uint32(int(seed) & 0xFFFFFFFF), float(missing), int(spectrum),
                                   int(bool(special)), int(device), ctypes.c_void_p(stream) if stream else None))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return a.ctypes.data_as(ctypes.c_void_p)


class PinnedBuffer:
    """Page-locked host block buffer (snpgpu_host_alloc) exposed as a numpy uint8 array."""

    def __init__(self, shape):
        self.shape = tuple(int(x) for x in shape)
        nbytes = int(np.prod(self.shape))
        p = ctypes.c_void_p()
        check(lib().snpgpu_host_alloc(nbytes, ctypes.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).reshape(self.shape)

    @property
    def ptr(self):
        return self._p.value

    def free(self):
        if self._p:
            self.array = None
            lib().snpgpu_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def tri_size(n):
    return n * (n + 1) // 2


class Accumulator:
    """One streaming accumulator context (level 1 of the C ABI)."""

    def __init__(self, kind, n_samp, device=0, bayesian=False, row_begin=0, row_end=0,
                 max_block_snps=0, stream=None):
        self.kind, self.n = kind, int(n_samp)
        o = Opts(int(device), int(bool(bayesian)), int(row_begin), int(row_end), int(max_block_snps),
                 ctypes.c_void_p(stream) if stream else None)
        h = ctypes.c_void_p()
        check(lib().snpgpu_create(int(kind), self.n, ctypes.byref(o), ctypes.byref(h)))
        self._h = h
        self.row_begin = int(row_begin)
        self.row_end = int(row_end) if row_end else self.n
        self.full = (self.row_begin == 0 and self.row_end == self.n)

    def close(self):
        if self._h:
            lib().snpgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- feeding ---------------------------------------------------------
    def feed(self, geno, fmt=None):
        """geno: numpy uint8 [n_snp][n_samp] (U8) or [n_snp][ceil(n/4)] (PACKED2)."""
        g = np.ascontiguousarray(geno, dtype=np.uint8)
        if fmt is None:
            fmt = GENO_U8 if g.shape[1] == self.n else GENO_PACKED2
        exp = self.n if fmt == GENO_U8 else (self.n + 3) // 4
        if g.ndim != 2 or g.shape[1] != exp:
            raise ValueError("genotype block has the wrong shape")
        check(lib().snpgpu_feed(self._h, _ptr(g), g.shape[0], fmt, HOST))

    def feed_pinned(self, buf, n_snp, fmt=GENO_U8):
        """Asynchronous feed out of a PinnedBuffer (call host_wait(buf) before refilling it)."""
        check(lib().snpgpu_feed(self._h, ctypes.c_void_p(buf.ptr), int(n_snp), fmt, HOST_PINNED))

    def host_wait(self, buf):
        check(lib().snpgpu_host_wait(self._h, ctypes.c_void_p(buf.ptr)))

    def feed_device(self, dev_ptr, n_snp, fmt=GENO_PACKED2):
        check(lib().snpgpu_feed(self._h, ctypes.c_void_p(int(dev_ptr)), int(n_snp), fmt, DEVICE))

    def block_stats_device(self, dev_ptr, n_snp, sum_ptr, num_ptr, fmt=GENO_PACKED2):
        """per-SNP (sum, num) of `n_snp` rows at device address dev_ptr into device int32 arrays (snpgpu_block_stats)"""
        check(lib().snpgpu_block_stats(self._h, ctypes.c_void_p(int(dev_ptr)), int(n_snp), fmt, ctypes.c_void_p(int(sum_ptr)),
                                       ctypes.c_void_p(int(num_ptr))))

    def feed_device_stats(self, dev_ptr, n_snp, sum_ptr, num_ptr, fmt=GENO_PACKED2):
        check(lib().snpgpu_feed_stats(self._h, ctypes.c_void_p(int(dev_ptr)), int(n_snp), fmt, DEVICE, ctypes.c_void_p(int(sum_ptr)),
                                      ctypes.c_void_p(int(num_ptr))))

    def sync(self):
        check(lib().snpgpu_sync(self._h))

    def counts(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().snpgpu_counts(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_timing(self, on=True):
        check(lib().snpgpu_set_timing(self._h, int(on)))

    def get_timing(self, which):
        """(summed kernel ms, launches) of the pair kernel: which=0 popcount, 1 SYRK."""
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        check(lib().snpgpu_get_timing(self._h, int(which), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def slab_size(self):
        return lib().snpgpu_slab_size(self._h)

    def _shape(self, packed):
        return (self.slab_size(),) if packed else (self.n, self.n)

    # ---- finalisers --------------------------------------------------------
    def ibs_num(self, packed=False, out_ptrs=None):
        if out_ptrs is not None:
            check(lib().snpgpu_ibs_num(self._h, *[ctypes.c_void_p(int(x)) for x in out_ptrs], int(packed), DEVICE))
            return None
        o = [np.empty(self._shape(packed), np.int32) for _ in range(3)]
        check(lib().snpgpu_ibs_num(self._h, _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), int(packed), HOST))
        return o

    def ibs_ave(self, packed=False):
        o = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_ibs_ave(self._h, _ptr(o), int(packed), HOST))
        return o

    def king_robust_counts(self):
        o = np.empty((self.slab_size(), 5), np.uint32)
        check(lib().snpgpu_king_robust_counts(self._h, _ptr(o), HOST))
        return o

    def king_robust(self, family=None, packed=False, out_ptrs=None):
        fam = None if family is None else np.ascontiguousarray(family, np.int32)
        if out_ptrs is not None:
            check(lib().snpgpu_king_robust(self._h, _ptr(fam), ctypes.c_void_p(int(out_ptrs[0])),
                                           ctypes.c_void_p(int(out_ptrs[1])), int(packed), DEVICE))
            return None
        a = np.empty(self._shape(packed), np.float64)
        b = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_king_robust(self._h, _ptr(fam), _ptr(a), _ptr(b), int(packed), HOST))
        return a, b

    def king_homo(self, packed=False, out_ptrs=None):
        if out_ptrs is not None:
            check(lib().snpgpu_king_homo(self._h, ctypes.c_void_p(int(out_ptrs[0])), ctypes.c_void_p(int(out_ptrs[1])), int(packed), DEVICE))
            return None
        a = np.empty(self._shape(packed), np.float64)
        b = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_king_homo(self._h, _ptr(a), _ptr(b), int(packed), HOST))
        return a, b

    def grm_gcta(self, packed=False, out_ptr=None):
        """out_ptr: optional DEVICE pointer receiving the result (then nothing is returned)."""
        if out_ptr is not None:
            check(lib().snpgpu_grm_gcta(self._h, ctypes.c_void_p(int(out_ptr)), int(packed), DEVICE))
            return None
        o = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_grm_gcta(self._h, _ptr(o), int(packed), HOST))
        return o

    def pca_cov(self, packed=False, normalize=True, trace_in=0.0, want_matrix=True, out_ptr=None):
        tr = ctypes.c_double(0)
        if out_ptr is not None:
            check(lib().snpgpu_pca_cov(self._h, ctypes.c_void_p(int(out_ptr)), int(packed), int(normalize),
                                       float(trace_in), ctypes.byref(tr), DEVICE))
            return None, tr.value
        o = np.empty(self._shape(packed), np.float64) if want_matrix else None
        check(lib().snpgpu_pca_cov(self._h, _ptr(o), int(packed), int(normalize), float(trace_in),
                                   ctypes.byref(tr), HOST))
        return o, tr.value

    def ibd_mom(self, e, constraint=False, packed=False):
        e = np.ascontiguousarray(e, np.float64)
        a = np.empty(self._shape(packed), np.float64)
        b = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_ibd_mom(self._h, _ptr(e), int(bool(constraint)), _ptr(a), _ptr(b), int(packed), HOST))
        return a, b

    def eigmix(self, diagadj=True, scale=1.0, packed=False):
        o = np.empty(self._shape(packed), np.float64)
        check(lib().snpgpu_eigmix(self._h, int(bool(diagadj)), float(scale), _ptr(o), int(packed), HOST))
        return o

    def indiv_beta(self, mode=1, packed=False):
        o = np.empty(self._shape(packed), np.float64)
        avg = ctypes.c_double(0)
        check(lib().snpgpu_indiv_beta(self._h, int(mode), _ptr(o), ctypes.byref(avg), int(packed), HOST))
        return o, avg.value

    def pca_panel_trace(self):
        tr = ctypes.c_double(0)
        check(lib().snpgpu_pca_panel_trace(self._h, ctypes.byref(tr)))
        return tr.value

    def pca_panel_matmul(self, scale, q_ptr, m, y_ptr, fp32=False):
        """Y += scale * (this panel's part of C) Q; q_ptr/y_ptr: device pointers, column-major n x m.
        fp32: the product on fp32 matrix instructions (snpgpu_pca_panel_matmul_f32)."""
        fn = lib().snpgpu_pca_panel_matmul_f32 if fp32 else lib().snpgpu_pca_panel_matmul
        check(fn(self._h, float(scale), ctypes.c_void_p(int(q_ptr)), int(m), ctypes.c_void_p(int(y_ptr))))

    def finalize_inplace(self, diagadj=True, scale=1.0):
        """GRM_GCTA / EIGMIX: the accumulators become the final matrix in place (then usable by the eigen solver)."""
        check(lib().snpgpu_finalize_inplace(self._h, int(bool(diagadj)), float(scale)))

    def panel_entries(self, rows, cols):
        """fp64 result-plane entries (rows[k], cols[k]) of this panel (snpgpu_panel_entries)"""
        return panel_entries(self._h, rows, cols)

    def pca_eigen(self, k):
        w = np.empty(k, np.float64)
        v = np.empty((k, self.n), np.float64)   # column-major n x k
        check(lib().snpgpu_pca_eigen(self._h, int(k), _ptr(w), _ptr(v), HOST))
        return w, v.T


def panel_entries(handle, rows, cols):
    r = np.ascontiguousarray(rows, np.int64)
    c = np.ascontiguousarray(cols, np.int64)
    out = np.empty(r.size, np.float64)
    check(lib().snpgpu_panel_entries(handle, _ptr(r), _ptr(c), r.size, _ptr(out)))
    return out


class MultiAccumulator:
    """snpgpu_multi: ONE host process driving several GPUs (the in-process counterpart of the one-process-per-GPU drivers
    of multigpu.py).  `devices` may repeat an ordinal (several panels' worth of "devices" on one GPU in tests)."""

    def __init__(self, kind, n_samp, devices=(0,), panels_per_device=1, n_passes=1, pass_index=0, bayesian=False,
                 max_block_snps=0):
        self.kind, self.n = kind, int(n_samp)
        self._devs = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
        mo = MultiOpts(self._devs, len(devices), int(panels_per_device), int(n_passes), int(pass_index))
        o = Opts(0, int(bool(bayesian)), 0, 0, int(max_block_snps), None)
        h = ctypes.c_void_p()
        check(lib().snpgpu_multi_create(int(kind), self.n, ctypes.byref(o), ctypes.byref(mo), ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().snpgpu_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        check(lib().snpgpu_multi_info(self._h, ctypes.byref(a), ctypes.byref(b)))
        return {"n_panels": a.value, "uses_rccl": bool(b.value)}

    def status(self):
        """snpgpu_multi_get_status as a dict: devices (listed / distinct), panels, the resolved panels_per_device, whether RCCL carries
        the eigen exchanges, peer access (ordered pairs of distinct devices: all / enabled) and the self-test outcomes (-1 = not run)."""
        st = MultiStatus()
        check(lib().snpgpu_multi_get_status(self._h, ctypes.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in MultiStatus._fields_ if k != "reserved"}

    def comm_selftest(self):
        """One broadcast + sum-reduction of a known pattern over the devices through the exchange path in use, then a known 2-bit block
        through the feed-forward star and a known slab from every device through the gather path; raises on a wrong word.  Returns
        True when the exchange path is RCCL."""
        b = ctypes.c_int(0)
        check(lib().snpgpu_multi_comm_selftest(self._h, ctypes.byref(b)))
        return bool(b.value)

    def panels(self):
        """[(row_begin, row_end, device)] of the resident panels"""
        out = []
        for i in range(self.info()["n_panels"]):
            r0, r1, d = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
            check(lib().snpgpu_multi_panel(self._h, i, None, ctypes.byref(r0), ctypes.byref(r1), ctypes.byref(d)))
            out.append((r0.value, r1.value, d.value))
        return out

    def entries(self, rows, cols):
        """result entries (rows[k] <= cols[k]) wherever their panels live: snpgpu_panel_entries per resident panel"""
        rows = np.asarray(rows, np.int64)
        cols = np.asarray(cols, np.int64)
        out = np.full(rows.size, np.nan)
        for i in range(self.info()["n_panels"]):
            h, r0, r1, d = ctypes.c_void_p(), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
            check(lib().snpgpu_multi_panel(self._h, i, ctypes.byref(h), ctypes.byref(r0), ctypes.byref(r1), ctypes.byref(d)))
            sel = (rows >= r0.value) & (rows < r1.value)
            if sel.any():
                out[sel] = panel_entries(h, rows[sel], cols[sel])
        return out

    def feed(self, geno, fmt=None):
        g = np.ascontiguousarray(geno, dtype=np.uint8)
        if fmt is None:
            fmt = GENO_U8 if g.shape[1] == self.n else GENO_PACKED2
        exp = self.n if fmt == GENO_U8 else (self.n + 3) // 4
        if g.ndim != 2 or g.shape[1] != exp:
            raise ValueError("genotype block has the wrong shape")
        check(lib().snpgpu_multi_feed(self._h, _ptr(g), g.shape[0], fmt, HOST))

    def feed_device(self, dev_ptr, n_snp, fmt=GENO_PACKED2):
        check(lib().snpgpu_multi_feed(self._h, ctypes.c_void_p(int(dev_ptr)), int(n_snp), fmt, DEVICE))

    def sync(self):
        check(lib().snpgpu_multi_sync(self._h))

    def counts(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().snpgpu_multi_counts(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def _tri(self, dtype, cols=None):
        shape = (tri_size(self.n),) if cols is None else (tri_size(self.n), cols)
        return np.full(shape, -1 if np.issubdtype(dtype, np.integer) else np.nan, dtype)

    def ibs_num(self, out=None):
        o = out or [self._tri(np.int32) for _ in range(3)]
        check(lib().snpgpu_multi_ibs_num(self._h, _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), HOST))
        return o

    def king_robust(self, family=None, out=None):
        fam = None if family is None else np.ascontiguousarray(family, np.int32)
        a, b = out or (self._tri(np.float64), self._tri(np.float64))
        check(lib().snpgpu_multi_king_robust(self._h, _ptr(fam), _ptr(a), _ptr(b), HOST))
        return a, b

    def king_robust_counts(self, out=None):
        o = out if out is not None else np.zeros((tri_size(self.n), 5), np.uint32)
        check(lib().snpgpu_multi_king_robust_counts(self._h, _ptr(o), HOST))
        return o

    def grm_gcta(self, out=None, out_ptr=None):
        if out_ptr is not None:
            check(lib().snpgpu_multi_grm_gcta(self._h, ctypes.c_void_p(int(out_ptr)), DEVICE))
            return None
        o = out if out is not None else self._tri(np.float64)
        check(lib().snpgpu_multi_grm_gcta(self._h, _ptr(o), HOST))
        return o

    def pca_cov(self, normalize=True, want_matrix=True):
        tr = ctypes.c_double(0)
        o = self._tri(np.float64) if want_matrix else None
        check(lib().snpgpu_multi_pca_cov(self._h, _ptr(o), int(normalize), ctypes.byref(tr), HOST))
        return o, tr.value

    def finalize_inplace(self, diagadj=True, scale=1.0):
        check(lib().snpgpu_multi_finalize_inplace(self._h, int(bool(diagadj)), float(scale)))

    def topk_eigen(self, k, scale=0.0, tol=1e-9, block=0, depth=0, seed=20240601, fp32_until=0.0):
        """(eigenvalues [k], eigenvectors [n, k], info) on the host"""
        opts = EigOpts(tol=float(tol), block=int(block), depth=int(depth), max_restarts=0, seed=int(seed), y_buf=None,
                       reduce=REDUCE_FN(), user=None, fp32_until=float(fp32_until))
        w = np.empty(k, np.float64)
        v = np.empty((k, self.n), np.float64)
        info = EigInfo()
        check(lib().snpgpu_multi_topk_eigen(self._h, float(scale), int(k), ctypes.byref(opts), _ptr(w), _ptr(v), HOST,
                                            ctypes.byref(info)))
        return w, v.T, {"restarts": info.restarts, "matmuls": info.matmuls, "max_rel_residual": info.max_rel_residual,
                        "block": info.block, "depth": info.depth, "matmuls_fp32": info.matmuls_fp32}


class Projector:
    """RAII wrapper over snpgpu_proj (PCA projections, include/snpgpu.h section 1b).
    Matrices follow R's layouts: eigvec [n_eig][n_samp]; per-block results [n_snp][n_eig]."""

    def __init__(self, n_samp, n_eig, device=0, max_block_snps=16384):
        self._h = ctypes.c_void_p()
        self.n, self.k = int(n_samp), int(n_eig)
        o = Opts(device=device, bayesian=0, row_begin=0, row_end=0, max_block_snps=max_block_snps, stream=None)
        check(lib().snpgpu_proj_create(self.n, self.k, ctypes.byref(o), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().snpgpu_proj_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _block(self, geno):
        g = np.ascontiguousarray(geno, dtype=np.uint8)
        fmt = GENO_U8 if g.shape[1] == self.n else GENO_PACKED2
        exp = self.n if fmt == GENO_U8 else (self.n + 3) // 4
        if g.ndim != 2 or g.shape[1] != exp:
            raise ValueError("genotype block has the wrong shape")
        return g, fmt

    def set_eigvec(self, eigvec):
        e = np.ascontiguousarray(eigvec, dtype=np.float64)
        if e.shape != (self.k, self.n):
            raise ValueError("eigvec must be [n_eig][n_samp]")
        check(lib().snpgpu_proj_set_eigvec(self._h, _ptr(e), HOST))

    def snp_corr(self, geno):
        g, fmt = self._block(geno)
        out = np.empty((g.shape[0], self.k), dtype=np.float64)
        check(lib().snpgpu_proj_snp_corr(self._h, _ptr(g), g.shape[0], fmt, HOST, _ptr(out), HOST))
        return out

    def snp_loading(self, geno, bayesian=False):
        g, fmt = self._block(geno)
        out = np.empty((g.shape[0], self.k), dtype=np.float64)
        af = np.empty(g.shape[0], dtype=np.float64)
        sc = np.empty(g.shape[0], dtype=np.float64)
        check(lib().snpgpu_proj_snp_loading(self._h, _ptr(g), g.shape[0], fmt, HOST, int(bool(bayesian)), _ptr(out),
                                            _ptr(af), _ptr(sc), HOST))
        return out, af, sc

    def samp_loading_feed(self, geno, sload, afreq, scale):
        g, fmt = self._block(geno)
        sl = np.ascontiguousarray(sload, dtype=np.float64)
        af = np.ascontiguousarray(afreq, dtype=np.float64)
        sc = np.ascontiguousarray(scale, dtype=np.float64)
        if sl.shape != (g.shape[0], self.k) or af.shape != (g.shape[0],) or sc.shape != (g.shape[0],):
            raise ValueError("sload / afreq / scale do not match the block")
        check(lib().snpgpu_proj_samp_loading_feed(self._h, _ptr(g), g.shape[0], fmt, HOST, _ptr(sl), _ptr(af), _ptr(sc), HOST))

    def samp_loading(self):
        out = np.empty((self.k, self.n), dtype=np.float64)
        check(lib().snpgpu_proj_samp_loading(self._h, _ptr(out), HOST))
        return out
