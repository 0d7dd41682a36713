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


def broadcast_grid(grid_tensor, src: int = 0):
    """The job's single collective: rank `src` holds the light field, everybody else receives it (no-op for world 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(grid_tensor, src=src)
    return grid_tensor


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
