"""ctypes binding of the C oracle (oracle/lfi_oracle.c → oracle/build/liblfi_oracle.so).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED — see oracle/lfi_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liblfi_oracle.so")

TEN_M16 = 0
TEN_EXACT = 1
ALL_FOCUS = 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc through oracle/Makefile (building the checker is not using it)."""
    src_newer = (not os.path.exists(_SO)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO) for f in ("lfi_oracle.c", "lfi_oracle.h"))
    if force or src_newer:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p


def _declare(l: C.CDLL) -> None:
    l.lfo_f32_to_f16.restype = C.c_uint16
    l.lfo_f32_to_f16.argtypes = [C.c_float]
    l.lfo_f64_to_f16.restype = C.c_uint16
    l.lfo_f64_to_f16.argtypes = [C.c_double]
    l.lfo_f16_to_f32.restype = C.c_float
    l.lfo_f16_to_f32.argtypes = [C.c_uint16]
    l.lfo_f16_to_u8_rz.restype = C.c_uint8
    l.lfo_f16_to_u8_rz.argtypes = [C.c_uint16]
    l.lfo_interpret_trajectory.restype = C.c_int
    l.lfo_interpret_trajectory.argtypes = [C.c_char_p, C.c_int, C.c_int, _f32p]
    l.lfo_trajectory_point.restype = None
    l.lfo_trajectory_point.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
    l.lfo_trajectory_center.restype = None
    l.lfo_trajectory_center.argtypes = [_f32p, _f32p]
    l.lfo_weights_f32.restype = None
    l.lfo_weights_f32.argtypes = [_f32p, C.c_int, C.c_int, C.c_float, _f32p]
    l.lfo_weight_matrix_f16.restype = None
    l.lfo_weight_matrix_f16.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_float, _u16p]
    l.lfo_offsets.restype = None
    l.lfo_offsets.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _vp, _vp]
    l.lfo_focus_map_ids.restype = C.c_int
    l.lfo_focus_map_ids.argtypes = [_f32p, C.c_int, C.c_int, _i32p, C.c_int]
    l.lfo_block_radius.restype = None
    l.lfo_block_radius.argtypes = [C.c_int, C.c_int, _i32p]
    l.lfo_hash32.restype = C.c_uint32
    l.lfo_hash32.argtypes = [C.c_uint32] * 5
    l.lfo_fill_synthetic.restype = None
    l.lfo_fill_synthetic.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_uint32]
    l.lfo_fill_synthetic_plane.restype = None
    l.lfo_fill_synthetic_plane.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_uint32]
    l.lfo_warp_coords.restype = None
    l.lfo_warp_coords.argtypes = [C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_float, C.c_float,
                                  C.c_int, C.c_int, _vp]
    common = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint, _vp,
              C.c_float, C.c_float]
    l.lfo_blend_std.restype = None
    l.lfo_blend_std.argtypes = common + [C.c_int, C.c_int, _vp, _vp]
    l.lfo_blend_ten.restype = None
    l.lfo_blend_ten.argtypes = common + [C.c_int, C.c_int, C.c_int, _vp, _vp]
    l.lfo_blend_f64.restype = None
    l.lfo_blend_f64.argtypes = common + [C.c_int, C.c_int, _vp]
    l.lfo_focus_estimate.restype = None
    l.lfo_focus_estimate.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_float, C.c_float, _i32p,
                                     C.c_int, C.c_int, _vp]
    l.lfo_focus_filter.restype = None
    l.lfo_focus_filter.argtypes = [_vp, C.c_int, C.c_int, _i32p, C.c_int, C.c_int, _vp]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


# ---- host parameterisation ---------------------------------------------------------------------------------

def interpret_trajectory(text: str, cols: int, rows: int) -> np.ndarray:
    out = np.zeros(4, dtype=np.float32)
    n = lib().lfo_interpret_trajectory(text.encode(), cols, rows, out.ctypes.data_as(_f32p))
    if n < 0:
        raise ValueError("bad trajectory " + text)
    return out


def trajectory_point(se, views, i):
    se, sp = _f32(se)
    out = np.zeros(2, dtype=np.float32)
    lib().lfo_trajectory_point(sp, views, i, out.ctypes.data_as(_f32p))
    return out


def weights_f32(view_xy, cols, rows, effect):
    v, vp = _f32(view_xy)
    out = np.zeros(cols * rows, dtype=np.float32)
    lib().lfo_weights_f32(vp, cols, rows, effect, out.ctypes.data_as(_f32p))
    return out


def weight_matrix_f16(se, cols, rows, views, effect):
    se, sp = _f32(se)
    out = np.zeros((views, cols * rows), dtype=np.uint16)
    lib().lfo_weight_matrix_f16(sp, cols, rows, views, effect, out.ctypes.data_as(_u16p))
    return out


def offsets(se, cols, rows, width, height, aspect, focus):
    se, sp = _f32(se)
    off = np.zeros((cols * rows, 2), dtype=np.float32)
    foc = np.zeros((cols * rows, 2), dtype=np.int32)
    lib().lfo_offsets(sp, cols, rows, width, height, aspect, focus, _p(off), _p(foc))
    return off, foc


def focus_map_ids(se, cols, rows, max_ids=32):
    se, sp = _f32(se)
    out = np.zeros(max_ids, dtype=np.int32)
    n = lib().lfo_focus_map_ids(sp, cols, rows, out.ctypes.data_as(_i32p), max_ids)
    return out[:n].copy()


def block_radius(width, height):
    out = np.zeros(2, dtype=np.int32)
    lib().lfo_block_radius(width, height, out.ctypes.data_as(_i32p))
    return out


def synthetic_lf(n_images, width, height, seed):
    out = np.empty((n_images, height, width, 4), dtype=np.uint8)
    lib().lfo_fill_synthetic(_p(out), n_images, width, height, seed)
    return out


def synthetic_plane(g, width, height, seed):
    out = np.empty((height, width, 4), dtype=np.uint8)
    lib().lfo_fill_synthetic_plane(_p(out), g, width, height, seed)
    return out


# ---- device-side arithmetic ----------------------------------------------------------------------------------

def _rows(height, threads):
    threads = max(1, min(threads, height))
    edges = np.linspace(0, height, threads + 1).astype(int)
    return [(int(edges[i]), int(edges[i + 1])) for i in range(threads) if edges[i + 1] > edges[i]]


def _run_rows(fn, height, threads, rows=None):
    """ctypes drops the GIL during the call, so row bands run on `threads` host cores.
    rows=(y0, y1) restricts the computation to that band (the rest of the output stays zero)."""
    if rows is not None:
        y0, y1 = int(rows[0]), int(rows[1])
        if threads <= 1 or y1 - y0 < 2:
            fn(y0, y1)
            return
        bands = [(y0 + a, y0 + b) for a, b in _rows(y1 - y0, threads)]
    else:
        bands = _rows(height, threads)
    if len(bands) == 1:
        fn(*bands[0])
        return
    with ThreadPoolExecutor(max_workers=len(bands)) as ex:
        list(ex.map(lambda b: fn(*b), bands))


def warp_coords(g, width, height, focused, offs, all_focus=False, map_plane=None, focus=0.0, rng=0.0):
    out = np.empty((height, width, 2), dtype=np.int32)
    foc = np.ascontiguousarray(focused, dtype=np.int32)
    off = np.ascontiguousarray(offs, dtype=np.float32)
    lib().lfo_warp_coords(g, width, height, _p(foc), _p(off), int(all_focus), _p(map_plane), focus, rng, 0, height,
                          _p(out))
    return out


def _blend_args(lf, focused, offs, weights_vn, v0, v1, all_focus, map_plane, focus, rng):
    lf = np.ascontiguousarray(lf, dtype=np.uint8)
    n, h, w, _ = lf.shape
    foc = np.ascontiguousarray(focused, dtype=np.int32)
    off = np.ascontiguousarray(offs, dtype=np.float32)
    wv = np.ascontiguousarray(weights_vn, dtype=np.uint16)
    views = wv.shape[0]
    v1 = views if v1 is None else v1
    if map_plane is not None:
        map_plane = np.ascontiguousarray(map_plane, dtype=np.uint8)
    keep = (lf, foc, off, wv, map_plane)
    args = [_p(lf), n, w, h, _p(foc), _p(off), _p(wv), views, v0, v1, ALL_FOCUS if all_focus else 0, _p(map_plane),
            focus, rng]
    return keep, args, (n, h, w, views, v1)


def blend_std(lf, focused, offs, weights_vn, v0=0, v1=None, all_focus=False, map_plane=None, focus=0.0, rng=0.0,
              return_prequant=False, threads=1, rows=None):
    keep, args, (n, h, w, views, v1) = _blend_args(lf, focused, offs, weights_vn, v0, v1, all_focus, map_plane, focus,
                                                   rng)
    out = np.zeros((views, h, w, 4), dtype=np.uint8)
    pre = np.zeros((views, h, w, 3), dtype=np.float32) if return_prequant else None
    _run_rows(lambda a, b: lib().lfo_blend_std(*args, a, b, _p(out), _p(pre)), h, threads, rows)
    return (out, pre) if return_prequant else out


def blend_ten(lf, focused, offs, weights_vn, model=TEN_M16, v0=0, v1=None, all_focus=False, map_plane=None, focus=0.0,
              rng=0.0, return_prequant=False, threads=1, rows=None):
    keep, args, (n, h, w, views, v1) = _blend_args(lf, focused, offs, weights_vn, v0, v1, all_focus, map_plane, focus,
                                                   rng)
    out = np.zeros((views, h, w, 4), dtype=np.uint8)
    pre = np.zeros((views, h, w, 3), dtype=np.float32) if return_prequant else None
    _run_rows(lambda a, b: lib().lfo_blend_ten(*args, model, a, b, _p(out), _p(pre)), h, threads, rows)
    return (out, pre) if return_prequant else out


def blend_f64(lf, focused, offs, weights_vn, v0=0, v1=None, all_focus=False, map_plane=None, focus=0.0, rng=0.0,
              threads=1):
    keep, args, (n, h, w, views, v1) = _blend_args(lf, focused, offs, weights_vn, v0, v1, all_focus, map_plane, focus,
                                                   rng)
    out = np.zeros((views, h, w, 3), dtype=np.float64)
    _run_rows(lambda a, b: lib().lfo_blend_f64(*args, a, b, _p(out)), h, threads)
    return out


def focus_estimate(lf, offs, ids, focus, rng, radius, threads=1, rows=None):
    lf = np.ascontiguousarray(lf, dtype=np.uint8)
    n, h, w, _ = lf.shape
    off = np.ascontiguousarray(offs, dtype=np.float32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    radius = np.ascontiguousarray(radius, dtype=np.int32)
    out = np.zeros((h, w, 4), dtype=np.uint8)
    _run_rows(lambda a, b: lib().lfo_focus_estimate(_p(lf), n, w, h, _p(off), _p(ids), len(ids), focus, rng,
                                                     radius.ctypes.data_as(_i32p), a, b, _p(out)), h, threads, rows)
    return out


def focus_filter(map0, radius, threads=1, rows=None):
    map0 = np.ascontiguousarray(map0, dtype=np.uint8)
    h, w = map0.shape[:2]
    radius = np.ascontiguousarray(radius, dtype=np.int32)
    out = np.zeros((h, w, 4), dtype=np.uint8)
    _run_rows(lambda a, b: lib().lfo_focus_filter(_p(map0), w, h, radius.ctypes.data_as(_i32p), a, b, _p(out)), h,
              threads, rows)
    return out
