"""ctypes binding of include/lfi.h (lib/liblfi_hip.so).  Thin: every method is one C-ABI call."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from .build import HIP_LIB

LFI_METHOD_STD = 0
LFI_METHOD_TEN_WM = 1
LFI_FLAG_UNIFIED_FOCUS_MAP = 1
LFI_FLAG_TEN_ROUND_PER_BATCH = 2
LFI_FLAG_SINGLE_SWEEP_DIRECTION = 4
LFI_FLAG_STD_ANALYTIC_BAND = 8
LFI_FLAG_STD_MEASURED_BAND = 16
LFI_FLAG_STD_BAND_PROBE_FAIL = 32
LFI_KERNEL_FOCUS_ESTIMATE = 2
METHODS = {"STD": LFI_METHOD_STD, "TEN_WM": LFI_METHOD_TEN_WM, "FOCUS": LFI_KERNEL_FOCUS_ESTIMATE}

# every symbol include/lfi.h declares
ABI_SYMBOLS = [
    "lfi_create", "lfi_destroy", "lfi_last_error", "lfi_abi_version", "lfi_device_count", "lfi_set_grid", "lfi_set_row_window",
    "lfi_upload_image", "lfi_attach_grid", "lfi_broadcast_grid", "lfi_grid_device_ptr", "lfi_fill_synthetic", "lfi_set_params",
    "lfi_attach_views", "lfi_views_device_ptr", "lfi_focus_map", "lfi_render", "lfi_benchmark", "lfi_timer_start",
    "lfi_timer_stop", "lfi_sync", "lfi_download_view", "lfi_download_map", "lfi_download_quilt", "lfi_download_quilt_tiles", "lfi_release_inputs", "lfi_alloc_pinned", "lfi_free_pinned", "lfi_upload_map", "lfi_set_stream",
    "lfi_set_variant", "lfi_list_variants", "lfi_download_coords", "lfi_download_prequant", "lfi_debug_mfma_f16",
    "lfi_grid_modified", "lfi_prepare", "lfi_memory_info", "lfi_last_kernel_name", "lfi_fill_synthetic_images", "lfi_set_output_layout", "lfi_view_layout", "lfi_fill_synthetic_scene", "lfi_upload_image_async", "lfi_upload_wait", "lfi_render_stream", "lfi_compare_view", "lfi_debug_mfma_f16_chain", "lfi_debug_pk_minmax3_f16", "lfi_std_band_info",
]


class LfiError(RuntimeError):
    pass


class _Params(C.Structure):
    _fields_ = [("views", C.c_int32), ("focused_offsets", C.c_void_p), ("offsets", C.c_void_p),
                ("weights_fp16", C.c_void_p), ("focus_map_ids", C.c_void_p), ("n_focus_ids", C.c_int32),
                ("focus", C.c_float), ("range", C.c_float), ("block_radius", C.c_int32 * 2), ("flags", C.c_uint32)]


class BenchStats(C.Structure):
    _fields_ = [("runs", C.c_int32), ("mean_ms", C.c_float), ("median_ms", C.c_float), ("min_ms", C.c_float),
                ("max_ms", C.c_float), ("back_to_back_ms", C.c_float)]


LFI_LAYOUT_RGBA = 0
LFI_LAYOUT_PLANAR_RGB = 1
LAYOUTS = {"rgba": LFI_LAYOUT_RGBA, "planar": LFI_LAYOUT_PLANAR_RGB}


class ViewLayout(C.Structure):
    _fields_ = [("layout", C.c_int32), ("rows", C.c_int32), ("row_pitch_bytes", C.c_size_t), ("plane_stride_bytes", C.c_size_t),
                ("view_stride_bytes", C.c_size_t)]


class Quality(C.Structure):
    _fields_ = [("mse", C.c_double * 3), ("psnr", C.c_double * 3), ("psnr_all", C.c_double), ("ssim", C.c_double * 3), ("ssim_all", C.c_double)]


class StdBandInfo(C.Structure):
    _fields_ = [("probed", C.c_int32), ("within_budget", C.c_int32), ("analytic_forced", C.c_int32), ("sums", C.c_int32),
                ("worst_fraction", C.c_float), ("probe_ms", C.c_float), ("message", C.c_char * 160)]


class MemoryInfo(C.Structure):
    _fields_ = [("grid_bytes", C.c_size_t), ("derived_bytes", C.c_size_t), ("views_bytes", C.c_size_t), ("maps_bytes", C.c_size_t),
                ("workspace_bytes", C.c_size_t), ("derived_build_ms", C.c_float)]


_lib = None


def load_hip_library() -> C.CDLL:
    """Load lib/liblfi_hip.so; a missing library is an error, never a fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime under the same soname (libamdhip64.so.7) as /opt/rocm's, and the first one a
    # process loads wins.  If ours pulled in /opt/rocm's first, a later `import torch` would find "No HIP GPUs"; so when torch is
    # installed it goes first (bench.py and the distributed path need it anyway).  The C++ host code is unaffected.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(HIP_LIB):
        raise LfiError(f"{HIP_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); lfinterpolator_amd has no CPU fallback")
    lib = C.CDLL(HIP_LIB)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    sig = {
        "lfi_create": (i, [i, C.POINTER(vp)]),
        "lfi_destroy": (i, [vp]),
        "lfi_last_error": (C.c_char_p, [vp]),
        "lfi_abi_version": (i, []),
        "lfi_device_count": (i, []),
        "lfi_set_grid": (i, [vp, i, i, i, i]),
        "lfi_set_row_window": (i, [vp, i, i, i, i]),
        "lfi_upload_image": (i, [vp, i, vp, sz]),
        "lfi_attach_grid": (i, [vp, vp, sz]),
        "lfi_broadcast_grid": (i, [C.POINTER(vp), i, i]),
        "lfi_grid_device_ptr": (i, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "lfi_fill_synthetic": (i, [vp, C.c_uint32]),
        "lfi_set_params": (i, [vp, C.POINTER(_Params)]),
        "lfi_attach_views": (i, [vp, vp, sz]),
        "lfi_views_device_ptr": (i, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "lfi_focus_map": (i, [vp]),
        "lfi_render": (i, [vp, i, i, i, i]),
        "lfi_benchmark": (i, [vp, i, i, i, i, i, i, C.POINTER(BenchStats)]),
        "lfi_timer_start": (i, [vp]),
        "lfi_timer_stop": (i, [vp, C.POINTER(C.c_float)]),
        "lfi_sync": (i, [vp]),
        "lfi_download_view": (i, [vp, i, vp, sz]),
        "lfi_download_map": (i, [vp, i, vp, sz]),
        "lfi_release_inputs": (i, [vp]),
        "lfi_download_quilt": (i, [vp, i, i, i, vp, sz]),
        "lfi_download_quilt_tiles": (i, [vp, i, i, i, i, i, vp, sz]),
        "lfi_alloc_pinned": (i, [sz, C.POINTER(vp)]),
        "lfi_free_pinned": (i, [vp]),
        "lfi_grid_modified": (i, [vp]),
        "lfi_upload_map": (i, [vp, i, vp, sz]),
        "lfi_set_stream": (i, [vp, vp]),
        "lfi_set_variant": (i, [vp, i, C.c_char_p]),
        "lfi_list_variants": (C.c_char_p, [i]),
        "lfi_download_coords": (i, [vp, i, i, i, vp]),
        "lfi_download_prequant": (i, [vp, i, i, i, vp]),
        "lfi_debug_mfma_f16": (i, [vp, vp, vp, vp]),
        "lfi_prepare": (i, [vp, i, i, i, i]),
        "lfi_render_stream": (i, [vp, i, i, vp, i, vp, sz]),
        "lfi_compare_view": (i, [vp, i, vp, sz, C.POINTER(Quality)]),
        "lfi_upload_image_async": (i, [vp, i, vp, sz]),
        "lfi_upload_wait": (i, [vp]),
        "lfi_fill_synthetic_scene": (i, [vp, C.c_uint32]),
        "lfi_debug_mfma_f16_chain": (i, [vp, i, i, vp, vp, vp]),
        "lfi_debug_pk_minmax3_f16": (i, [vp, vp]),
        "lfi_set_output_layout": (i, [vp, i]),
        "lfi_view_layout": (i, [vp, C.POINTER(ViewLayout)]),
        "lfi_fill_synthetic_images": (i, [vp, C.c_uint32, i, i]),
        "lfi_memory_info": (i, [vp, C.POINTER(MemoryInfo)]),
        "lfi_std_band_info": (i, [vp, C.POINTER(StdBandInfo)]),
        "lfi_last_kernel_name": (C.c_char_p, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export what lfi.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def broadcast_grid(contexts, root: int = 0) -> None:
    """lfi_broadcast_grid over a list of Context objects (single process, one context per GPU, RCCL)."""
    lib = load_hip_library()
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib.lfi_broadcast_grid(arr, len(contexts), root)
    if rc != 0:
        raise LfiError(f"lfi_broadcast_grid failed ({rc}): {lib.lfi_last_error(contexts[root]._h).decode()}")


class Context:
    """One GPU context (lfi_ctx).  Mirrors the device-facing half of the reference's Interpolator."""

    def __init__(self, device: int = 0):
        self._lib = load_hip_library()
        handle = C.c_void_p()
        rc = self._lib.lfi_create(device, C.byref(handle))
        if rc != 0:
            raise LfiError(f"lfi_create failed ({rc}): {self._lib.lfi_last_error(None).decode()}")
        self._h = handle
        self.device = device
        self.cols = self.rows = self.width = self.height = self.views = 0
        self._keep = None
        self._pinned = []

    # -- helpers --------------------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise LfiError(f"lfi error {rc}: {self._lib.lfi_last_error(self._h).decode()}")

    def close(self) -> None:
        if getattr(self, "_h", None):
            for p in self._pinned: # arrays from pinned_empty must not be used after this
                self._lib.lfi_free_pinned(C.c_void_p(p))
            self._pinned = []
            self._lib.lfi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def n_images(self) -> int:
        return self.cols * self.rows

    # -- grid -------------------------------------------------------------------------------------------------
    def set_grid(self, cols: int, rows: int, width: int, height: int) -> None:
        self._check(self._lib.lfi_set_grid(self._h, cols, rows, width, height))
        self.cols, self.rows, self.width, self.height = cols, rows, width, height
        self.out_rows = (0, height)

    def set_row_window(self, out_y0: int, out_y1: int, in_y0: int, in_y1: int) -> None:
        """Render rows [out_y0, out_y1) only, holding input rows [in_y0, in_y1) only (row-band sharding)."""
        self._check(self._lib.lfi_set_row_window(self._h, out_y0, out_y1, in_y0, in_y1))
        self.out_rows = (out_y0, out_y1)

    def upload_image(self, g: int, rgba: np.ndarray) -> None:
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert rgba.shape == (self.height, self.width, 4)
        self._check(self._lib.lfi_upload_image(self._h, g, _ptr(rgba), self.width * 4))

    def upload_grid(self, lf: np.ndarray, asynchronous: bool = False) -> None:
        """lf: [N][H][W][4] u8 with g = col*rows + row."""
        assert lf.shape == (self.n_images, self.height, self.width, 4)
        for g in range(self.n_images):
            if asynchronous:
                self.upload_image_async(g, lf[g])
            else:
                self.upload_image(g, lf[g])

    def upload_image_async(self, g: int, rgba: np.ndarray) -> None:
        """Enqueue the copy on the context's copy stream; `rgba` may be pageable (staged, free on return) or from pinned_empty
        (DMA'd in place: keep it alive until upload_wait / sync)."""
        assert rgba.shape == (self.height, self.width, 4) and rgba.dtype == np.uint8 and rgba.flags.c_contiguous
        self._check(self._lib.lfi_upload_image_async(self._h, g, _ptr(rgba), self.width * 4))

    def upload_wait(self) -> None:
        self._check(self._lib.lfi_upload_wait(self._h))

    def attach_grid(self, device_ptr: int, nbytes: int) -> None:
        self._check(self._lib.lfi_attach_grid(self._h, C.c_void_p(device_ptr), nbytes))

    def grid_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.lfi_grid_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def release_inputs(self) -> None:
        """Fixed-focus use only: the planar copy becomes the only copy of the inputs, the RGBA planes are freed."""
        self._check(self._lib.lfi_release_inputs(self._h))

    def grid_modified(self) -> None:
        """The input planes were written behind the library's back (attached buffer, raw device pointer)."""
        self._check(self._lib.lfi_grid_modified(self._h))

    def fill_synthetic(self, seed: int, g0: int | None = None, g1: int | None = None) -> None:
        if g0 is None and g1 is None:
            self._check(self._lib.lfi_fill_synthetic(self._h, seed))
        else:
            self._check(self._lib.lfi_fill_synthetic_images(self._h, seed, g0 or 0, self.n_images if g1 is None else g1))

    def fill_synthetic_scene(self, seed: int) -> None:
        """Structured light field (texture at a piecewise-constant focus) for focus-map timing; call after set_params."""
        self._check(self._lib.lfi_fill_synthetic_scene(self._h, seed))

    # -- parameters --------------------------------------------------------------------------------------------
    def set_params(self, hp, flags: int = 0) -> None:
        """hp: lfinterpolator_amd.host.HostParams (or anything with the same arrays)."""
        p = _Params()
        foc = np.ascontiguousarray(hp.focused_offsets, dtype=np.int32)
        off = np.ascontiguousarray(hp.offsets, dtype=np.float32)
        w = np.ascontiguousarray(hp.weights, dtype=np.uint16)
        ids = np.ascontiguousarray(hp.focus_map_ids, dtype=np.int32)
        assert foc.shape == (self.n_images, 2) and off.shape == (self.n_images, 2) and w.shape[1] == self.n_images
        p.views = w.shape[0]
        p.focused_offsets = foc.ctypes.data
        p.offsets = off.ctypes.data
        p.weights_fp16 = w.ctypes.data
        p.focus_map_ids = ids.ctypes.data if len(ids) else None
        p.n_focus_ids = len(ids)
        p.focus = float(hp.focus)
        p.range = float(hp.range)
        p.block_radius[0] = int(hp.block_radius[0])
        p.block_radius[1] = int(hp.block_radius[1])
        p.flags = flags
        self._check(self._lib.lfi_set_params(self._h, C.byref(p)))
        self.views = w.shape[0]

    def set_output_layout(self, layout) -> None:
        """'rgba' (the reference's planes) or 'planar' (alpha-free byte planes; downloads re-create alpha = 255)."""
        self._check(self._lib.lfi_set_output_layout(self._h, LAYOUTS[layout] if isinstance(layout, str) else layout))

    def view_layout(self) -> ViewLayout:
        vl = ViewLayout()
        self._check(self._lib.lfi_view_layout(self._h, C.byref(vl)))
        return vl

    def attach_views(self, device_ptr: int, nbytes: int) -> None:
        self._check(self._lib.lfi_attach_views(self._h, C.c_void_p(device_ptr), nbytes))

    def views_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.lfi_views_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # -- kernels -------------------------------------------------------------------------------------------------
    def focus_map(self) -> None:
        self._check(self._lib.lfi_focus_map(self._h))

    def render(self, method, all_focus: bool = False, v0: int = 0, v1: int | None = None) -> None:
        m = METHODS[method] if isinstance(method, str) else method
        self._check(self._lib.lfi_render(self._h, m, int(all_focus), v0, self.views if v1 is None else v1))

    def prepare(self, method, all_focus: bool = False, v0: int = 0, v1: int | None = None) -> None:
        m = METHODS[method] if isinstance(method, str) else method
        self._check(self._lib.lfi_prepare(self._h, m, int(all_focus), v0, self.views if v1 is None else v1))

    def memory_info(self) -> MemoryInfo:
        mi = MemoryInfo()
        self._check(self._lib.lfi_memory_info(self._h, C.byref(mi)))
        return mi

    def std_band_info(self) -> StdBandInfo:
        """The band method's self-check on this device (runs it if it has not run yet)."""
        info = StdBandInfo()
        self._check(self._lib.lfi_std_band_info(self._h, C.byref(info)))
        return info

    def last_kernel_name(self) -> str:
        return self._lib.lfi_last_kernel_name(self._h).decode()

    def render_stream(self, method, weights: np.ndarray, out: np.ndarray | None = None, all_focus: bool = False) -> None:
        """weights: [total_views][N] fp16 bits of the whole path; out: None or a [total_views][H][W][4] u8 array (ideally from
        pinned_empty) that receives every view."""
        m = METHODS[method] if isinstance(method, str) else method
        w = np.ascontiguousarray(weights, dtype=np.uint16)
        assert w.ndim == 2 and w.shape[1] == self.n_images
        if out is not None:
            assert out.shape == (w.shape[0], self.height, self.width, 4) and out.dtype == np.uint8 and out.flags.c_contiguous
        self._check(self._lib.lfi_render_stream(self._h, m, int(all_focus), _ptr(w), w.shape[0], _ptr(out) if out is not None else None,
                                                self.width * 4))

    def compare_view(self, v: int, reference: np.ndarray) -> Quality:
        ref = np.ascontiguousarray(reference, dtype=np.uint8)
        assert ref.shape == (self.height, self.width, 4)
        q = Quality()
        self._check(self._lib.lfi_compare_view(self._h, v, _ptr(ref), self.width * 4, C.byref(q)))
        return q

    def benchmark(self, method, all_focus=False, v0=0, v1=None, warmup=3, runs=20) -> BenchStats:
        m = METHODS[method] if isinstance(method, str) else method
        st = BenchStats()
        self._check(self._lib.lfi_benchmark(self._h, m, int(all_focus), v0, self.views if v1 is None else v1, warmup,
                                            runs, C.byref(st)))
        return st

    def timer_start(self) -> None:
        self._check(self._lib.lfi_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self._lib.lfi_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def sync(self) -> None:
        self._check(self._lib.lfi_sync(self._h))

    def set_stream(self, hip_stream: int | None) -> None:
        self._check(self._lib.lfi_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_variant(self, method, name: str) -> None:
        m = METHODS[method] if isinstance(method, str) else method
        self._check(self._lib.lfi_set_variant(self._h, m, name.encode()))

    def list_variants(self, method) -> list[str]:
        m = METHODS[method] if isinstance(method, str) else method
        return self._lib.lfi_list_variants(m).decode().split(",")

    # -- results -------------------------------------------------------------------------------------------------
    def download_view(self, v: int, out: np.ndarray | None = None) -> np.ndarray:
        """Whole-image array; with a row window only rows [out_y0, out_y1) are filled (the rest stays zero).
        `out` may be a page-locked array from `pinned_empty` (a true DMA instead of a staged copy)."""
        if out is None:
            out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        assert out.shape == (self.height, self.width, 4) and out.dtype == np.uint8 and out.flags.c_contiguous
        self._check(self._lib.lfi_download_view(self._h, v, _ptr(out), self.width * 4))
        return out

    def download_views(self, v0: int = 0, v1: int | None = None, out: np.ndarray | None = None) -> np.ndarray:
        v1 = self.views if v1 is None else v1
        if out is None:
            return np.stack([self.download_view(v) for v in range(v0, v1)])
        assert out.shape == (v1 - v0, self.height, self.width, 4)
        for v in range(v0, v1):
            self.download_view(v, out[v - v0])
        return out

    def pinned_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array over page-locked host memory (lfi_alloc_pinned); freed when the context is closed."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.lfi_alloc_pinned(nbytes, C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def download_quilt_tiles(self, out: np.ndarray, tiles_x: int, tiles_y: int, first_tile: int, n: int, v0: int = 0) -> None:
        """views v0 … v0+n-1 into tiles first_tile … of the quilt image `out` ((tiles_y·H, tiles_x·W, 4) uint8, C-contiguous)"""
        assert out.dtype == np.uint8 and out.flags.c_contiguous and out.shape == (tiles_y * self.height, tiles_x * self.width, 4)
        self._check(self._lib.lfi_download_quilt_tiles(self._h, tiles_x, tiles_y, first_tile, n, v0, _ptr(out), tiles_x * self.width * 4))

    def download_quilt(self, tiles_x: int, tiles_y: int, v0: int = 0) -> np.ndarray:
        out = np.empty((tiles_y * self.height, tiles_x * self.width, 4), dtype=np.uint8)
        self._check(self._lib.lfi_download_quilt(self._h, tiles_x, tiles_y, v0, _ptr(out), tiles_x * self.width * 4))
        return out

    def download_map(self, k: int) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._lib.lfi_download_map(self._h, k, _ptr(out), self.width * 4))
        return out

    def upload_map(self, k: int, rgba: np.ndarray) -> None:
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert rgba.shape == (self.height, self.width, 4)
        self._check(self._lib.lfi_upload_map(self._h, k, _ptr(rgba), self.width * 4))

    def download_coords(self, g: int, all_focus: bool = False, map_index: int = 1) -> np.ndarray:
        out = np.empty((self.height, self.width, 2), dtype=np.int32)
        self._check(self._lib.lfi_download_coords(self._h, g, int(all_focus), map_index, _ptr(out)))
        return out

    def download_prequant(self, method, v: int, all_focus: bool = False) -> np.ndarray:
        m = METHODS[method] if isinstance(method, str) else method
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self._lib.lfi_download_prequant(self._h, m, int(all_focus), v, _ptr(out)))
        return out

    def debug_mfma_f16_chain(self, a_bits: np.ndarray, b_bits: np.ndarray, shape: int = 0) -> np.ndarray:
        """C[32][32] = A[32][K] · B[K][32] (fp16 bit patterns) through K/16 (shape 0) or K/32 (shape 1) chained MFMAs."""
        a = np.ascontiguousarray(a_bits, dtype=np.uint16)
        b = np.ascontiguousarray(b_bits, dtype=np.uint16)
        k = a.shape[1]
        assert a.shape == (32, k) and b.shape == (k, 32)
        c = np.empty((32, 32), dtype=np.float32)
        self._check(self._lib.lfi_debug_mfma_f16_chain(self._h, shape, k, _ptr(a), _ptr(b), _ptr(c)))
        return c

    def debug_pk_minmax3_f16(self) -> int:
        """Mismatching halves of v_pk_minimum3_f16 / v_pk_maximum3_f16 against integer min / max over all byte triples."""
        out = C.c_uint32(0xffffffff)
        self._check(self._lib.lfi_debug_pk_minmax3_f16(self._h, C.byref(out)))
        return int(out.value)

    def debug_mfma_f16(self, a_bits: np.ndarray, b_bits: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a_bits, dtype=np.uint16)
        b = np.ascontiguousarray(b_bits, dtype=np.uint16)
        assert a.shape == (32, 16) and b.shape == (16, 32)
        c = np.empty((32, 32), dtype=np.float32)
        self._check(self._lib.lfi_debug_mfma_f16(self._h, _ptr(a), _ptr(b), _ptr(c)))
        return c
