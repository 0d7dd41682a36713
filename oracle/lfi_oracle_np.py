"""numpy restatement of the reference's shift-and-sum path — the second, independent oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by tests/golden/make_golden.py) to cross-check the C oracle
(oracle/lfi_oracle.c).  Never imported by the product package.

PARITY UNPINNED: the reference ships no golden vectors and cannot be built here (CUDA); this file follows the
reference source (citations relative to /root/reference) independently of the C restatement, vectorised over
pixels, so that an error in one restatement shows up as a disagreement between the two.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
TEN_M16 = 0
TEN_EXACT = 1


# ---------------------------------------------------------------------------------------------------------
# host parameterisation: src/interpolator.cu:139-246, 318-337
# ---------------------------------------------------------------------------------------------------------

def interpret_trajectory(text: str, cols: int, rows: int) -> np.ndarray:
    """src/interpolator.cu:318-337 — value_i * (colsRows[i % 2] - 1)."""
    dims = (cols, rows)
    out = np.zeros(4, dtype=F32)
    for i, tok in enumerate(text.split(",")[:4]):
        out[i] = F32(float(tok)) * F32(dims[i % 2] - 1)
    return out


def trajectory_point(se, views: int, i: int) -> np.ndarray:
    """src/interpolator.cu:174-182."""
    se = np.asarray(se, dtype=F32)
    if views <= 1:
        return se[:2].copy()
    step = (se[2:4] - se[0:2]) / F32(views - 1)
    return se[0:2] + step * F32(i)


def trajectory_center(se) -> np.ndarray:
    """src/interpolator.cu:189-192."""
    se = np.asarray(se, dtype=F32)
    return se[0:2] + (se[2:4] - se[0:2]) * F32(0.5)


def _dist(ax, ay, bx, by):
    dx = F32(bx) - F32(ax)
    dy = F32(by) - F32(ay)
    return np.sqrt(F32(dx * dx) + F32(dy * dy), dtype=F32)


def weights_f32(view_xy, cols: int, rows: int, effect: float) -> np.ndarray:
    """src/interpolator.cu:156-172 — g = col*rows + row, sequential float sum."""
    max_d = _dist(0, 0, cols, rows)
    out = np.zeros(cols * rows, dtype=F32)
    total = F32(0)
    g = 0
    for col in range(cols):
        for row in range(rows):
            w = F32(max_d - _dist(view_xy[0], view_xy[1], col, row))
            # powf: double pow rounded to float (agrees with glibc powf except for rare 1-ulp cases)
            w = F32(math.pow(float(w), float(F32(effect)))) if w >= 0 or float(effect).is_integer() else F32(np.nan)
            total = F32(total + w)
            out[g] = w
            g += 1
    return (out / total).astype(F32)


def weight_matrix_f16(se, cols: int, rows: int, views: int, effect: float) -> np.ndarray:
    """src/interpolator.cu:209-224 — [views][N] fp16 bit patterns (uint16)."""
    out = np.zeros((views, cols * rows), dtype=np.uint16)
    for v in range(views):
        line = weights_f32(trajectory_point(se, views, v), cols, rows, effect)
        out[v] = line.astype(np.float16).view(np.uint16)
    return out


def _round_half_away(x: np.ndarray) -> np.ndarray:
    x64 = x.astype(np.float64)
    return (np.sign(x64) * np.floor(np.abs(x64) + 0.5)).astype(np.int32)


def offsets(se, cols: int, rows: int, width: int, height: int, aspect: float, focus: float):
    """src/interpolator.cu:226-246 — returns (float2[N], int2[N])."""
    center = trajectory_center(se)
    offset_aspect = F32(F32(width) / F32(height)) / F32(aspect)
    off = np.zeros((cols * rows, 2), dtype=F32)
    g = 0
    for col in range(cols):
        for row in range(rows):
            ox = F32(F32(center[0] - F32(col)) / F32(cols)) * F32(width)
            oy = F32(F32(center[1] - F32(row)) / F32(rows)) * F32(height)
            oy = F32(oy * offset_aspect)
            off[g] = (ox, oy)
            g += 1
    focused = _round_half_away((off * F32(focus)).astype(F32))
    return off, focused


def focus_map_ids(se, cols: int, rows: int, max_ids: int = 32) -> np.ndarray:
    """src/interpolator.cu:194-207, ties broken by id (the reference leaves them to std::sort)."""
    center = trajectory_center(se)
    d = []
    g = 0
    for col in range(cols):
        for row in range(rows):
            d.append((float(_dist(col, row, center[0], center[1])), g))
            g += 1
    d.sort()
    return np.array([i for _, i in d[:min(max_ids, len(d))]], dtype=np.int32)


def block_radius(width: int, height: int) -> np.ndarray:
    """src/interpolator.cu:139-146 (+ the ≥1 guard for defect D6)."""
    r = [width // 100, height // 100]
    r = [v + 1 if v % 2 else v for v in r]
    return np.array([max(v, 1) for v in r], dtype=np.int32)


# ---------------------------------------------------------------------------------------------------------
# synthetic light field
# ---------------------------------------------------------------------------------------------------------

def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def synthetic_lf(n_images: int, width: int, height: int, seed: int) -> np.ndarray:
    """[N][H][W][4] u8, byte = hash32(seed, g, y, x, c) >> 24, alpha 255."""
    with np.errstate(over="ignore"):
        g = np.arange(n_images, dtype=np.uint32)[:, None, None, None]
        y = np.arange(height, dtype=np.uint32)[None, :, None, None]
        x = np.arange(width, dtype=np.uint32)[None, None, :, None]
        c = np.arange(4, dtype=np.uint32)[None, None, None, :]
        h = _mix32(np.uint32(seed) + g * np.uint32(0x9E3779B9))
        h = _mix32(h + y * np.uint32(0x85EBCA6B))
        h = _mix32(h + x * np.uint32(0xC2B2AE35) + c)
    out = (h >> np.uint32(24)).astype(np.uint8)
    out[..., 3] = 255
    return out


# ---------------------------------------------------------------------------------------------------------
# device-side arithmetic: src/kernels.cu
# ---------------------------------------------------------------------------------------------------------

def _fma32(a, b, c):
    """fmaf for operands whose exact a*b+c fits a double (true for every use below): one rounding to float."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F32)


def decode_focus(map_plane: np.ndarray, focus: float, rng: float) -> np.ndarray:
    """src/kernels.cu:134-137 — fma(map/255, range, focus) per pixel."""
    t = (map_plane[..., 0].astype(F32) / F32(255.0)).astype(F32)
    return _fma32(t, F32(rng), F32(focus))


def warp_coords(g, width, height, focused, offs, all_focus=False, focus_px=None):
    """src/kernels.cu:72-82 — unclamped sample coordinates (x', y') of image g for every pixel."""
    ys, xs = np.meshgrid(np.arange(height, dtype=np.int32), np.arange(width, dtype=np.int32), indexing="ij")
    if all_focus:
        sx = np.trunc(_fma32(focus_px, offs[g, 0], xs.astype(F32))).astype(np.int64).astype(np.int32)
        sy = np.trunc(_fma32(focus_px, offs[g, 1], ys.astype(F32))).astype(np.int64).astype(np.int32)
    else:
        sx = xs + np.int32(focused[g, 0])
        sy = ys + np.int32(focused[g, 1])
    return sx, sy


def fetch(plane: np.ndarray, sx: np.ndarray, sy: np.ndarray) -> np.ndarray:
    """surf2Dread with cudaBoundaryModeClamp: src/kernels.cu:119-126."""
    h, w = plane.shape[:2]
    return plane[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]


def _gather_stack(lf, focused, offs, all_focus, map_plane, focus, rng):
    n, h, w, _ = lf.shape
    fpx = decode_focus(map_plane, focus, rng) if all_focus else None
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    for g in range(n):
        sx, sy = warp_coords(g, w, h, focused, offs, all_focus, fpx)
        out[g] = fetch(lf[g], sx, sy)[..., :3]
    return out


def _f16_bits_to_f64(bits: np.ndarray) -> np.ndarray:
    return np.asarray(bits, dtype=np.uint16).view(np.float16).astype(np.float64)


def blend_std(lf, focused, offs, weights_vn, all_focus=False, map_plane=None, focus=0.0, rng=0.0,
              return_prequant=False):
    """src/kernels.cu:289-343 — ordered fmaf chain over g, RN-even to int, narrowed to u8, alpha 255."""
    n, h, w, _ = lf.shape
    views = weights_vn.shape[0]
    stack = _gather_stack(lf, focused, offs, all_focus, map_plane, focus, rng).astype(np.float64)
    wf = _f16_bits_to_f64(weights_vn)
    out = np.empty((views, h, w, 4), dtype=np.uint8)
    pre = np.empty((views, h, w, 3), dtype=F32) if return_prequant else None
    for v in range(views):
        acc = np.zeros((h, w, 3), dtype=F32)
        for g in range(n):
            acc = _fma32(stack[g], wf[v, g], acc)
        out[v, ..., :3] = np.rint(acc).astype(np.int32).astype(np.uint8)
        out[v, ..., 3] = 255
        if pre is not None:
            pre[v] = acc
    return (out, pre) if return_prequant else out


def _f16_to_u8_rz(h16: np.ndarray) -> np.ndarray:
    """half → unsigned char = __half2uchar_rz: truncate, saturate to 0..255, NaN → 0 (src/kernels.cu:393)."""
    v = h16.astype(np.float64)
    v = np.where(np.isnan(v), 0.0, v)
    return np.floor(np.clip(v, 0.0, 255.0)).astype(np.uint8)


def blend_ten(lf, focused, offs, weights_vn, model=TEN_M16, all_focus=False, map_plane=None, focus=0.0, rng=0.0,
              return_prequant=False):
    """src/kernels.cu:345-462 — per 16-image batch exact products, fp16 accumulator (M16) or one final rounding."""
    n, h, w, _ = lf.shape
    views = weights_vn.shape[0]
    n_pad = (n + 15) // 16 * 16
    stack = np.zeros((n_pad, h, w, 3), dtype=np.float64)
    stack[:n] = _gather_stack(lf, focused, offs, all_focus, map_plane, focus, rng)
    wf = np.zeros((views, n_pad), dtype=np.float64)
    wf[:, :n] = _f16_bits_to_f64(weights_vn)
    out = np.empty((views, h, w, 4), dtype=np.uint8)
    pre = np.empty((views, h, w, 3), dtype=F32) if return_prequant else None
    for v in range(views):
        acc = np.zeros((h, w, 3), dtype=np.float16)
        total = np.zeros((h, w, 3), dtype=np.float64)
        for b0 in range(0, n_pad, 16):
            s = np.tensordot(wf[v, b0:b0 + 16], stack[b0:b0 + 16], axes=(0, 0))  # exact in double
            total += s
            if model == TEN_M16:
                acc = (acc.astype(np.float64) + s).astype(np.float16)
        if model != TEN_M16:
            acc = total.astype(np.float16)
        out[v, ..., :3] = _f16_to_u8_rz(acc)
        out[v, ..., 3] = 255
        if pre is not None:
            pre[v] = acc.astype(F32)
    return (out, pre) if return_prequant else out


def blend_f64(lf, focused, offs, weights_vn, all_focus=False, map_plane=None, focus=0.0, rng=0.0):
    """exact weighted mean, no rounding: [V][H][W][3] float64."""
    stack = _gather_stack(lf, focused, offs, all_focus, map_plane, focus, rng).astype(np.float64)
    return np.tensordot(_f16_bits_to_f64(weights_vn), stack, axes=(1, 0))


def focus_estimate(lf, offs, ids, focus, rng, radius):
    """FocusMap::estimate src/kernels.cu:239-258 (+ focusDispersion :196-217, ElementRange :173-194)."""
    n, h, w, _ = lf.shape
    ys, xs = np.meshgrid(np.arange(h, dtype=np.int32), np.arange(w, dtype=np.int32), indexing="ij")
    step = F32(F32(rng) / F32(31))
    best_d = np.full((h, w), np.finfo(F32).max, dtype=F32)
    best_f = np.zeros((h, w), dtype=F32)
    flt_min = np.finfo(F32).tiny  # FLT_MIN, the reference's initial "max" (:178)
    for i in range(32):
        f = _fma32(step, F32(i), F32(focus))
        lo = np.full((9, h, w, 3), np.finfo(F32).max, dtype=F32)
        hi = np.full((9, h, w, 3), flt_min, dtype=F32)
        for g in ids:
            cx = np.trunc(_fma32(f, offs[g, 0], xs.astype(F32))).astype(np.int32)
            cy = np.trunc(_fma32(f, offs[g, 1], ys.astype(F32))).astype(np.int32)
            t = 0
            for dx in (-radius[0], 0, radius[0]):
                for dy in (-radius[1], 0, radius[1]):
                    px = fetch(lf[g], cx + dx, cy + dy)[..., :3].astype(F32)
                    lo[t] = np.minimum(lo[t], px)
                    hi[t] = np.maximum(hi[t], px)
                    t += 1
        total = np.zeros((h, w), dtype=F32)
        for t in range(9):
            total = (total + np.abs(lo[t] - hi[t]).max(axis=-1)).astype(F32)
        better = total < best_d
        best_d = np.where(better, total, best_d)
        best_f = np.where(better, f, best_f)
    normalized = ((best_f - F32(focus)) / F32(rng)).astype(F32)
    m = np.floor((normalized * F32(255.0)).astype(np.float64) + 0.5).astype(np.uint8)  # round(), values are >= 0
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., :3] = m[..., None]
    out[..., 3] = 255
    return out


def focus_filter(map0, radius):
    """FocusMap::filter src/kernels.cu:260-280 — mean over [x-rx, x+rx) × [y-ry, y+ry), clamped taps."""
    h, w = map0.shape[:2]
    rx, ry = max(int(radius[0]) // 10, 1), max(int(radius[1]) // 10, 1)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.int32), np.arange(w, dtype=np.int32), indexing="ij")
    acc = np.zeros((h, w), dtype=F32)
    count = 0
    for dx in range(-rx, rx):
        for dy in range(-ry, ry):
            acc = (acc + fetch(map0, xs + dx, ys + dy)[..., 0].astype(F32)).astype(F32)
            count += 1
    avg = (acc / F32(count)).astype(F32)
    m = np.floor(avg.astype(np.float64) + 0.5).astype(np.uint8)
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., :3] = m[..., None]
    out[..., 3] = 255
    return out
