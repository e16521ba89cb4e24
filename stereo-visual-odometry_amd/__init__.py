"""MI355X-native stereo visual-odometry hot path (FAST + pyramidal LK + triangulation + PnP).

The product is `libsvo_hip.so` (hand-written HIP kernels for gfx950 behind the C-ABI of
include/svo_abi.h) plus the C++ mirror of the reference's lzb_vio::System/Tracking API in host/.
This Python package is a thin ctypes binding of that C-ABI, used by tests/, bench.py and
__graft_entry__.py; torch appears only as the owner of device memory and streams.

There is no CPU fallback: importing works anywhere, but creating a Context raises when the HIP
library or a GPU is missing.  The directory name contains '-', so load it through
`__graft_entry__.load_package()` (importlib) under the module name `stereo_visual_odometry_amd`.
"""
from .binding import (Context, Config, StepResult, PnPResult, KP_DTYPE, STEP_DTYPE, SvoError,
                      build_library, library_path, load_library, MEM_HOST, MEM_DEVICE, MODE_LK, MODE_ORB, LK_ACCUM_EXACT, LK_ACCUM_SSE2, LK_ACCUM_SIMD128, LK_ACCUM_SSE2_LEGACY)

__all__ = ["Context", "Config", "StepResult", "PnPResult", "KP_DTYPE", "STEP_DTYPE", "SvoError",
           "build_library", "library_path", "load_library", "MEM_HOST", "MEM_DEVICE", "MODE_LK", "MODE_ORB", "LK_ACCUM_EXACT", "LK_ACCUM_SSE2", "LK_ACCUM_SIMD128", "LK_ACCUM_SSE2_LEGACY"]
