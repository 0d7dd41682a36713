"""ctypes binding of the C++ host parameterisation (lib/liblfi_host.so, csrc/host/params.cpp)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from .build import HOST_LIB

_lib = None


def load_host_library() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run __graft_entry__.build()")
        lib = C.CDLL(HOST_LIB)
        lib.lfi_host_build_params.restype = C.c_int
        lib.lfi_host_build_params.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_float, C.c_float,
                                              C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32 * 2), C.c_char_p,
                                              C.c_size_t]
        lib.lfi_host_float_to_half.restype = C.c_uint16
        lib.lfi_host_float_to_half.argtypes = [C.c_float]
        lib.lfi_host_half_to_float.restype = C.c_float
        lib.lfi_host_half_to_float.argtypes = [C.c_uint16]
        lib.lfi_host_load_image.restype = C.c_int
        lib.lfi_host_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_char_p, C.c_size_t]
        lib.lfi_host_write_png.restype = C.c_int
        lib.lfi_host_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        lib.lfi_host_load_grid.restype = C.c_int
        lib.lfi_host_load_grid.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4 + [C.c_void_p, C.c_char_p, C.c_size_t]
        _lib = lib
    return _lib


@dataclass
class HostParams:
    """The reference's per-launch parameter block (src/kernels.cu:15-17, 63-69 + the weight matrix)."""
    focused_offsets: np.ndarray  # [N][2] int32
    offsets: np.ndarray          # [N][2] float32
    weights: np.ndarray          # [V][N] uint16 (fp16 bits)
    focus_map_ids: np.ndarray    # [≤32] int32
    focus: float
    range: float
    block_radius: np.ndarray     # [2] int32

    def rows(self, v0: int, v1: int) -> "HostParams":
        """The same block restricted to the weight rows [v0, v1) — what one rank of a view-sharded job needs."""
        return HostParams(self.focused_offsets, self.offsets, np.ascontiguousarray(self.weights[v0:v1]),
                          self.focus_map_ids, self.focus, self.range, self.block_radius)


def build_params(cols: int, rows: int, width: int, height: int, trajectory: str, focus: float = 0.0, range: float = 0.0,
                 effect: float = 3.0, aspect: float = 1.0, views: int = 64) -> HostParams:
    """What Interpolator::interpolate prepares before its launches (reference src/interpolator.cu:250-256)."""
    lib = load_host_library()
    n = cols * rows
    foc = np.zeros((n, 2), dtype=np.int32)
    off = np.zeros((n, 2), dtype=np.float32)
    w = np.zeros((max(views, 1), n), dtype=np.uint16)
    ids = np.zeros(32, dtype=np.int32)
    n_ids = C.c_int32()
    radius = (C.c_int32 * 2)()
    err = C.create_string_buffer(512)
    rc = lib.lfi_host_build_params(cols, rows, width, height, trajectory.encode(), focus, range, effect, aspect, views,
                                   foc.ctypes.data, off.ctypes.data, w.ctypes.data, ids.ctypes.data, C.byref(n_ids),
                                   C.byref(radius), err, len(err))
    if rc != 0:
        raise ValueError(err.value.decode())
    return HostParams(foc, off, w, ids[:n_ids.value].copy(), focus, range, np.array([radius[0], radius[1]], np.int32))


def _err_call(fn, *args):
    err = C.create_string_buffer(512)
    rc = fn(*args, err, len(err))
    if rc != 0:
        raise RuntimeError(err.value.decode())


def load_image(path: str) -> np.ndarray:
    """csrc/host/image_io.cpp: PNG / PPM → [H][W][4] u8."""
    lib = load_host_library()
    w, h = C.c_int(), C.c_int()
    _err_call(lib.lfi_host_load_image, path.encode(), C.byref(w), C.byref(h), None)
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    _err_call(lib.lfi_host_load_image, path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p))
    return out


def write_png(path: str, image: np.ndarray) -> None:
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = image.shape
    _err_call(load_host_library().lfi_host_write_png, path.encode(), w, h, c, image.ctypes.data_as(C.c_void_p))


def load_grid(path: str):
    """LfLoader::loadData: returns (cols, rows, [N][H][W][4] u8 with g = col*rows + row)."""
    lib = load_host_library()
    cols, rows, w, h = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _err_call(lib.lfi_host_load_grid, path.encode(), C.byref(cols), C.byref(rows), C.byref(w), C.byref(h), None)
    out = np.empty((cols.value * rows.value, h.value, w.value, 4), dtype=np.uint8)
    _err_call(lib.lfi_host_load_grid, path.encode(), C.byref(cols), C.byref(rows), C.byref(w), C.byref(h),
              out.ctypes.data_as(C.c_void_p))
    return cols.value, rows.value, out
