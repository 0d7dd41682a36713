"""CPU, world_size 2, gloo: the multi-GPU path's logic — view sharding, per-rank weight rows, the single grid broadcast —
with the oracle standing in for the kernel (tests only).  Real multi-GPU runs are the driver's (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest


def test_view_range_partition(native):
    for total, world in [(64, 1), (64, 2), (64, 8), (256, 8), (45, 4), (7, 8), (0, 3)]:
        ranges = [native.view_range(total, world, r) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 <= a1
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        native.view_range(8, 2, 2)


def test_row_band_input_rows(native, oracle_c):
    """Row-band sharding: the bands tile the image and each band's input rows are exactly the rows its warp samples."""
    cols = rows = 8
    W, H = 64, 90
    hp = native.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 4)
    for world in (1, 2, 3, 8):
        covered = 0
        for rank in range(world):
            band = native.row_band(H, world, rank)
            in_rows = native.input_rows(band, hp.focused_offsets, H)
            sampled = set()
            for g in range(cols * rows):
                for y in (band[0], band[1] - 1):
                    sampled.add(int(np.clip(y + hp.focused_offsets[g, 1], 0, H - 1)))
            assert in_rows == (min(sampled), max(sampled) + 1)
            assert 0 <= in_rows[0] <= band[0] or in_rows[0] <= H - 1
            covered += band[1] - band[0]
        assert covered == H
    # a band in the middle of a tall image needs only band + halo rows, far fewer than H
    band = native.row_band(1080, 8, 3)
    hp_big = native.build_params(8, 8, 1920, 1080, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 4)
    lo, hi = native.input_rows(band, hp_big.focused_offsets, 1080)
    assert hi - lo == (band[1] - band[0]) + 2 * 108 and hi - lo < 1080 // 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import lfinterpolator_amd as L
    from oracle import lfi_oracle_c as oc  # checker standing in for the GPU kernel in this CPU test

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cols = rows = 4
    W, H, total = 48, 20, 10
    grid = torch.zeros((cols * rows, H, W, 4), dtype=torch.uint8)
    if rank == 0:
        grid.copy_(torch.from_numpy(oc.synthetic_lf(cols * rows, W, H, 0x1F1F)))
    L.broadcast_grid(grid, src=0)
    hp, v0, v1 = L.rank_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, total, world, rank)
    assert hp.weights.shape[0] == v1 - v0
    local = oc.blend_std(grid.numpy(), hp.focused_offsets, hp.offsets, hp.weights)
    # MAX over ranks of a per-rank time, as bench.py does
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == world
    dist.barrier()
    np.save(os.path.join(out_dir, f"views_{rank}.npy"), local)
    np.save(os.path.join(out_dir, f"range_{rank}.npy"), np.array([v0, v1]))
    dist.destroy_process_group()


def test_two_rank_view_sharding_matches_single_rank(native, oracle_c, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cols = rows = 4
    W, H, total = 48, 20, 10
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 0x1F1F)
    hp = native.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, total)
    full = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    covered = 0
    for r in range(world):
        v0, v1 = np.load(tmp_path / f"range_{r}.npy")
        part = np.load(tmp_path / f"views_{r}.npy")
        assert (part == full[v0:v1]).all()
        covered += v1 - v0
    assert covered == total


def test_image_slices_and_all_focus_rows(native):
    """All-gather distribution: the ranks' image slices tile the grid in equal shares; the all-focus row reach contains the
    fixed-focus reach and what the per-pixel warp can sample at both ends of the focus range."""
    for n, world in ((64, 1), (64, 8), (225, 8), (225, 2), (9, 4), (3, 8)):
        per = -(-n // world)
        covered = []
        for r in range(world):
            g0, g1 = native.image_slice(n, world, r)
            assert 0 <= g0 <= g1 <= n and g1 - g0 <= per
            covered += list(range(g0, g1))
        assert covered == list(range(n))
    cols = rows = 8
    W, H = 160, 120
    hp = native.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.05, 0.12, 7.0, 1.783, 4)
    for world in (2, 3, 8):
        for rank in range(world):
            band = native.row_band(H, world, rank)
            lo, hi = native.input_rows_all_focus(band, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, H)
            flo, fhi = native.input_rows(band, hp.focused_offsets, H)
            assert 0 <= lo <= flo and fhi <= hi <= H
            for g in range(cols * rows):
                for f in (hp.focus, hp.focus + hp.range):
                    for y in (band[0], band[1] - 1):
                        sy = int(np.clip(int(np.float32(f) * np.float32(hp.offsets[g, 1]) + np.float32(y)), 0, H - 1))
                        assert lo <= sy < hi


def _allgather_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import lfinterpolator_amd as L
    from oracle import lfi_oracle_c as oc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, W, H = 9, 24, 10                      # 9 images over 2 ranks: shares of 5 and 4, the flat buffer is padded to 10 planes
    plane = H * W * 4
    per = -(-n // world)
    flat = torch.zeros(per * world * plane, dtype=torch.uint8)
    grid = flat[: n * plane].view(n, H, W, 4)
    g0, g1 = L.image_slice(n, world, rank)
    for g in range(g0, g1):                  # every rank produces ITS images only (bench.py: lfi_fill_synthetic_images)
        grid[g].copy_(torch.from_numpy(oc.synthetic_plane(g, W, H, 0x1F1F)))
    L.allgather_grid(flat, rank, world)
    np.save(os.path.join(out_dir, f"grid_{rank}.npy"), grid.numpy())
    dist.destroy_process_group()


def test_two_rank_allgather_distribution(native, oracle_c, tmp_path):
    """each rank contributes 1/G of the light field, one in-place all-gather (gloo here, RCCL on GPUs)
    leaves the whole grid on every rank."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_allgather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = oracle_c.synthetic_lf(9, 24, 10, 0x1F1F)
    for r in range(world):
        assert (np.load(tmp_path / f"grid_{r}.npy") == want).all()
