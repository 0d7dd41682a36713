"""View sharding of a trajectory over GPUs (one process per GPU): the only multi-GPU decomposition of the hot path.

Every (view, pixel) output is independent, so ranks split the trajectory's views into contiguous ranges and need no
data-path collective; the input grid is broadcast once at load time (RCCL over xGMI through torch.distributed, backend
"nccl" on GPUs / "gloo" in the CPU tests).  Offsets depend only on the trajectory centre and are identical on every
rank; each rank keeps its own rows of the weight matrix.
"""
from __future__ import annotations

from .host import HostParams, build_params


def view_range(total_views: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced split: the first (total % world) ranks get one extra view."""
    if not (0 <= rank < world) or total_views < 0:
        raise ValueError("bad rank/world/total")
    base, extra = divmod(total_views, world)
    v0 = rank * base + min(rank, extra)
    return v0, v0 + base + (1 if rank < extra else 0)


def rank_params(cols: int, rows: int, width: int, height: int, trajectory: str, focus: float, range_: float, effect: float,
                aspect: float, total_views: int, world: int, rank: int) -> tuple[HostParams, int, int]:
    """Parameters of the whole trajectory restricted to this rank's views: (params with V_local weight rows, v0, v1)."""
    hp_all = build_params(cols, rows, width, height, trajectory, focus, range_, effect, aspect, total_views)
    v0, v1 = view_range(total_views, world, rank)
    return hp_all.rows(v0, v1), v0, v1


def _forced() -> bool:
    """LFI_BENCH_FORCE_DIST=1: issue the collectives with one rank too (a one-GPU rehearsal of the RCCL calls, bench.py)"""
    import os
    return os.environ.get("LFI_BENCH_FORCE_DIST") == "1"


def broadcast_grid(grid_tensor, src: int = 0):
    """The job's single collective: rank `src` holds the light field, everybody else receives it (no-op for world 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced()):
        dist.broadcast(grid_tensor, src=src)
    return grid_tensor


def image_slice(n_images: int, world: int, rank: int) -> tuple[int, int]:
    """Images [g0, g1) that rank `rank` produces (loads / generates) for the all-gather distribution: equal shares of
    ceil(N / world) images, the last ranks' shares cut at N."""
    per = -(-n_images // world)
    return min(rank * per, n_images), min((rank + 1) * per, n_images)


def allgather_grid(flat_tensor, rank: int, world: int):
    """All-gather distribution of the light field: rank r has filled its slice of the input planes — bytes
    [r·S, (r+1)·S) of `flat_tensor`, S = len / world (the tensor is padded to a multiple of world) — and ONE in-place all-gather
    gives every rank everything.  Each rank sends 1/G of the bytes instead of rank 0 sending all of them: with xGMI's
    point-to-point links every link carries 1/G of the grid, where a broadcast is bound by the root's links.  No-op for world 1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced()):
        assert flat_tensor.numel() % world == 0
        share = flat_tensor.numel() // world
        dist.all_gather_into_tensor(flat_tensor, flat_tensor[rank * share:(rank + 1) * share].clone())
    return flat_tensor


# ---- row-band (spatial) sharding: each rank renders a band of rows of EVERY view and holds only the input rows the band's
# warp reaches — per-GPU bytes shrink ≈1/G for inputs and outputs alike (view sharding re-reads the whole input on every rank)

def row_band(height: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced split of the image rows."""
    return view_range(height, world, rank)


def input_rows(band: tuple[int, int], focused_offsets, height: int) -> tuple[int, int]:
    """Rows [in_y0, in_y1) of the input images that output rows `band` sample: clamp(y + oy_g, 0, H-1) over all images g."""
    import numpy as np
    oy = np.asarray(focused_offsets)[:, 1].astype(np.int64)
    lo = np.clip(band[0] + oy, 0, height - 1).min()
    hi = np.clip(band[1] - 1 + oy, 0, height - 1).max()
    return int(lo), int(hi) + 1


def input_rows_all_focus(band: tuple[int, int], offsets, focus_map_ids, focus: float, range_: float, block_radius, height: int) -> tuple[int, int]:
    """Rows [in_y0, in_y1) a rank must hold to compute ITS BAND of the focus maps and to render the band all-focused: the
    per-pixel warp (int)fma(f, offset.y, y) at both ends of [focus, focus + range] for every image (render) and, for the images
    the estimate samples, from the band extended by the filter's reach, plus / minus the block radius (the 3×3 taps) — the same
    conservative bounds (±1 for float rounding) the library checks."""
    import math
    import numpy as np
    off_y = np.asarray(offsets, dtype=np.float64)[:, 1]
    f_lo, f_hi = min(focus, focus + range_), max(focus, focus + range_)
    d_lo = np.minimum(f_lo * off_y, f_hi * off_y)
    d_hi = np.maximum(f_lo * off_y, f_hi * off_y)
    lo = math.floor(band[0] + d_lo.min()) - 1
    hi = math.ceil(band[1] - 1 + d_hi.max()) + 1
    ry = int(block_radius[1])
    fry = max(ry // 10, 1)
    e0, e1 = max(band[0] - fry, 0), min(band[1] + fry, height)
    ids = np.asarray(focus_map_ids, dtype=np.int64)
    if len(ids):
        lo = min(lo, math.floor(e0 + d_lo[ids].min()) - 1 - ry)
        hi = max(hi, math.ceil(e1 - 1 + d_hi[ids].max()) + 1 + ry)
    return max(min(lo, height - 1), 0), min(max(hi, 0), height - 1) + 1
