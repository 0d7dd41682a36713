"""lfinterpolator_amd — MI355X (gfx950) light-field view interpolation hot path.

The product is the C-ABI library ``lib/liblfi_hip.so`` (hand-written HIP kernels, ``include/lfi.h``) plus the C++ host
code in ``csrc/host`` (``Interpolator``, ``LfLoader``, the command line).  This Python package is plumbing for tests and
``bench.py``: a ctypes binding of the two shared libraries.  There is no CPU fallback — importing works without a GPU
(so that the build and symbol checks can run anywhere), but creating a context raises unless a gfx950 device and the
built library are present.
"""
from .abi import (LFI_METHOD_STD, LFI_METHOD_TEN_WM, LFI_FLAG_UNIFIED_FOCUS_MAP, LFI_FLAG_TEN_ROUND_PER_BATCH, LFI_FLAG_SINGLE_SWEEP_DIRECTION, LFI_FLAG_STD_ANALYTIC_BAND, LFI_FLAG_STD_MEASURED_BAND, LFI_FLAG_STD_BAND_PROBE_FAIL,
                  LFI_LAYOUT_RGBA, LFI_LAYOUT_PLANAR_RGB,
                  Context, LfiError, load_hip_library, ABI_SYMBOLS)
from .host import HostParams, build_params, load_host_library, load_image, write_png, load_grid
from .build import build_all
from .sharding import view_range, rank_params, broadcast_grid, allgather_grid, image_slice, row_band, input_rows, input_rows_all_focus
from . import build

__all__ = ["LFI_LAYOUT_RGBA", "LFI_LAYOUT_PLANAR_RGB", "LFI_METHOD_STD", "LFI_METHOD_TEN_WM", "LFI_FLAG_UNIFIED_FOCUS_MAP", "LFI_FLAG_TEN_ROUND_PER_BATCH", "LFI_FLAG_SINGLE_SWEEP_DIRECTION", "LFI_FLAG_STD_ANALYTIC_BAND", "LFI_FLAG_STD_MEASURED_BAND", "LFI_FLAG_STD_BAND_PROBE_FAIL",
           "Context", "LfiError", "load_hip_library", "ABI_SYMBOLS", "HostParams", "build_params", "load_host_library", "load_image", "write_png", "load_grid",
           "build_all", "view_range", "rank_params", "broadcast_grid", "allgather_grid", "image_slice", "row_band", "input_rows", "input_rows_all_focus"]
