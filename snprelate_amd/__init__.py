"""snprelate_amd -- MI355X-native pairwise relatedness hot path behind SNPRelate's snpgds* API.

The compute path is libsnpgpu.so (hand-written HIP for gfx950, C ABI in
include/snpgpu.h); this package is the host-side mirror of the reference's R
interface for that path.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import Accumulator, SnpGpuError  # noqa: F401
