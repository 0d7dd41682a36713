#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X: novel views/s (+ Gpix/s) of the TEN_WM hot path,
8×8 light field @1920×1080, 64-view trajectory per GPU (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one launch of the blend kernel over the synthetic grid resident in HBM = 64 novel views of 1920×1080 from
64 input images on each GPU.  One process per GPU.  Multi-GPU: the trajectory has 64·N views, rank r renders views
[64r, 64r+64) — the path shards over views with no data-path collective (weak scaling: per-GPU work is fixed); the
input grid is generated on rank 0 and broadcast ONCE over RCCL/xGMI before the timed region.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     — HBM roofline of the blend kernel: algorithmic bytes 4·W·H·(N_images + V) per launch ÷ the kernel's
                 average launch time measured with HIP events on the launch stream (through the C-ABI timer);
  cpu_baseline — the CPU oracle (a port of the reference's STD arithmetic) timed on the host cores on the same
                 workload (N = 1 only).  oracle/ is used here ONLY as that baseline, never in the GPU path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COLS, ROWS, WIDTH, HEIGHT = 8, 8, 1920, 1080
VIEWS_PER_GPU = 64
TRAJECTORY, FOCUS, ASPECT, EFFECT = "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0   # reference README.md:7
SEED = 0x1F1F
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3  # dense fp32 MFMA (= fp32 vector) peak: 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz


def cpu_baseline(hp, threads: int) -> dict:
    """The oracle's scalar STD blend on one full step of the same workload, on `threads` host cores."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import lfi_oracle_c as oc
    oc.build()
    n = COLS * ROWS
    lf = np.empty((n, HEIGHT, WIDTH, 4), dtype=np.uint8)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda g: lf.__setitem__(g, oc.synthetic_plane(g, WIDTH, HEIGHT, SEED)), range(n)))
    t0 = time.perf_counter()
    oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=threads)
    dt = time.perf_counter() - t0
    v = hp.weights.shape[0]
    return {"value": v / dt, "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"1 step: {v} views of {WIDTH}x{HEIGHT} from {n} images, scalar fp32 FMA weighted mean "
                      f"(oracle STD), {dt:.2f} s wall",
            "gpix_per_s": v * WIDTH * HEIGHT / dt / 1e9}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--method", default="TEN_WM", choices=["TEN_WM", "STD"])
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", default="views", choices=["views", "rows"],
                    help="views (default, the contract's weak-scaling run): 64 views per GPU, every GPU holds the whole grid; "
                         "rows: strong scaling of ONE 64-view render — each GPU renders a band of rows and holds only the "
                         "input rows the band's warp reaches (SURVEY.md §8(f).2)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed launches during set-up, before the W warm-up steps, so that the clocks have ramped "
                         "(with 5 warm-up launches = 1 ms of work the first timed launches still run at idle clocks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
            return 2
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    import lfinterpolator_amd as L

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    # Rehearsal on a one-GPU box: LFI_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two ranks on one
    # device).  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    rehearse = os.environ.get("LFI_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_images = COLS * ROWS
    ctx = L.Context(device_index)
    ctx.set_grid(COLS, ROWS, WIDTH, HEIGHT)
    stream = torch.cuda.Stream(device=dev)
    if args.shard == "views":
        # input planes live in a torch tensor so that RCCL (torch.distributed "nccl") can broadcast into them
        grid = torch.empty((n_images, HEIGHT, WIDTH, 4), dtype=torch.uint8, device=dev)
        ctx.attach_grid(grid.data_ptr(), grid.numel())
        ctx.set_stream(stream.cuda_stream)
        if rank == 0:
            ctx.fill_synthetic(SEED)
            ctx.sync()
        L.broadcast_grid(grid, src=0)    # the one collective of the job: 531 MB over xGMI (RCCL), outside the timed region
        torch.cuda.synchronize()
        ctx.grid_modified()              # the attached planes were written by the collective, not through the ABI
        # host parameters for the whole trajectory; each rank keeps its own rows of the weight matrix
        total_views = VIEWS_PER_GPU * world
        hp, v0, v1 = L.rank_params(COLS, ROWS, WIDTH, HEIGHT, TRAJECTORY, FOCUS, 0.0, EFFECT, ASPECT, total_views, world, rank)
        assert v1 - v0 == VIEWS_PER_GPU
        in_rows_n = out_rows_n = HEIGHT
    else:
        # one 64-view render split into row bands: no collective at all with synthetic data (every rank generates the rows it
        # holds; real data would be scattered band + halo per rank)
        total_views = VIEWS_PER_GPU
        hp = L.build_params(COLS, ROWS, WIDTH, HEIGHT, TRAJECTORY, FOCUS, 0.0, EFFECT, ASPECT, total_views)
        band = L.row_band(HEIGHT, world, rank)
        held = L.input_rows(band, hp.focused_offsets, HEIGHT)
        ctx.set_row_window(band[0], band[1], held[0], held[1])
        in_rows_n, out_rows_n = held[1] - held[0], band[1] - band[0]
        grid = torch.empty((n_images, in_rows_n, WIDTH, 4), dtype=torch.uint8, device=dev)
        ctx.attach_grid(grid.data_ptr(), grid.numel())
        ctx.set_stream(stream.cuda_stream)
        ctx.fill_synthetic(SEED)
        ctx.sync()
        ctx.grid_modified()              # attached planes: announce that they are final
    ctx.set_params(hp)
    views = torch.empty((VIEWS_PER_GPU, out_rows_n, WIDTH, 4), dtype=torch.uint8, device=dev)
    ctx.attach_views(views.data_ptr(), views.numel())
    ctx.set_variant(args.method, args.variant)

    def barrier():
        if world > 1:
            dist.barrier()

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:   # set-up: bring the GPU out of its idle power state
        for _ in range(20):
            ctx.render(args.method)
        ctx.sync()
    for _ in range(args.warmup):
        ctx.render(args.method)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()                      # HIP event on the launch stream
    for _ in range(args.steps):
        ctx.render(args.method)
    kernel_ms = ctx.timer_stop()           # HIP event + synchronise: time of the K launches on that stream
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed, kernel_ms / 1e3], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max, kernel_s_max = float(t[0]), float(t[1])

    # cheap sanity check that the timed launches rendered something: alpha must be 255 everywhere, RGB not constant
    sample = views[0, out_rows_n // 2, :64].cpu().numpy()
    assert (sample[:, 3] == 255).all() and sample[:, :3].std() > 0, "render produced no image"

    if rank == 0:
        value = total_views * args.steps / elapsed_max
        # bytes per launch on this GPU: rows held of every input plane + rows rendered of every view
        b_alg = 4.0 * WIDTH * (in_rows_n * n_images + out_rows_n * VIEWS_PER_GPU)
        t_launch = kernel_s_max / args.steps
        achieved = b_alg / t_launch / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")           # PMC-derived HBM bytes per launch, if measured
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.method}/{args.variant}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        f_alg = 6.0 * n_images * VIEWS_PER_GPU * WIDTH * out_rows_n   # 3 channels × (multiply + add)
        line = {
            "metric": "novel views/sec + Gpix/sec, 8x8 LF @1080p TEN_WM" if args.method == "TEN_WM"
                      else "novel views/sec + Gpix/sec, 8x8 LF @1080p STD",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.shard == "views" else "strong",
            "vs_baseline": None,
            "dtype": "f16" if args.method == "TEN_WM" else "f32", "data": "synthetic",
            "config": {"workload": f"{COLS}x{ROWS} LF @{WIDTH}x{HEIGHT}, {VIEWS_PER_GPU}-view -t trajectory per GPU, "
                                   f"-m {args.method}, -f {FOCUS} -a {ASPECT} -s {EFFECT:g}",
                       "views_per_gpu": VIEWS_PER_GPU, "images": n_images, "variant": args.variant,
                       "inputs": ("resident in HBM before the timed region: RGBA planes + the planar alpha-free copy the TEN_WM kernel "
                                  "reads (derived once per change of the inputs by planar_build, ~0.7 ms, DESIGN.md 4.1)"
                                  if args.method == "TEN_WM" and args.variant in ("auto", "planar_m2_nt", "planar_m2", "planar_ring2_nt")
                                  else "resident in HBM before the timed region: RGBA planes"),
                       "parallelism": (f"views sharded over {world} GPU(s), grid broadcast once ({'gloo rehearsal' if rehearse else 'RCCL'})"
                                       if args.shard == "views" else
                                       f"rows sharded over {world} GPU(s): {out_rows_n} output rows from {in_rows_n} input rows on rank 0, "
                                       "no collective")},
            "gpix_per_s": value * WIDTH * HEIGHT / 1e9,
            "prewarm_ms": args.prewarm_ms,
            "kernel_ms_per_launch": t_launch * 1e3,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "frac_of_measured_copy_6290": achieved / 6290.0,
                         "mfma_frac_of_2500_tflops": f_alg / t_launch / 2.5e15},
        }
        if args.method == "STD" and args.variant not in ("auto", "filtered_m2_nt"):
            # the exact-fp32 MFMA kernels are bound by the fp32 matrix pipe (the default STD kernel computes on the
            # fp16 matrix pipe and recomputes the sums near x.5 with the fmaf chain: HBM-bound like TEN_WM): F_alg = 6·N·V·W·H flops per launch against
            # the dense fp32 matrix peak (157.3 TFLOP/s: 256 CUs × 4 SIMDs × 64 flop/cycle × 2.4 GHz)
            tflops = f_alg / t_launch / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tflops / FP32_MATRIX_PEAK_TFLOPS, "traffic": traffic,
                                "algorithmic_flops_per_launch": f_alg, "hbm_frac_of_8000_gbs": achieved / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(hp, threads)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
