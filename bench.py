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

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline     — HBM roofline of the blend kernel: the bytes the kernel has to move in the layouts in use (3 B per pixel on an
                 alpha-free side, 4 B on an RGBA side) ÷ the kernel's average launch time measured with HIP events on the launch
                 stream (through the C-ABI timer) = `achieved`, `frac`; `frac_algorithmic` is the same on SURVEY.md §8(d)'s
                 4·W·H·(N_images + V) bytes, which count an alpha byte the planar layouts never touch (comparison only);
  cpu_baseline — the CPU oracle (a port of the reference's STD arithmetic) timed on the host cores on the same
                 workload (N = 1 only).  oracle/ is used here ONLY as that baseline, never in the GPU path;
  cpu_baseline_1thread — the same scalar code on one thread, four full-frame views;
  also_detail  — (N = 1) the other BASELINE configurations and methods, timed in the same run after the headline step:
                 config 3 (15×15 @1080p, 45 views; TEN_WM and STD), config 4 per rank and whole, config 5 fixed focus (TEN_WM, STD,
                 the non-tensor wavefront kernel) and end to end (focus map + all-focus render), cold one-shot renders of configs 2
                 and 5 — each with ms, algorithmic bytes, roofline fraction and the kernel that ran;
  also         — the same table, compact ([ms, fraction, kernel] per key), LAST on the line.

Other modes:  --config 4  strong-scaled BASELINE config 4 (8×8 @4K, one 256-view trajectory split over the GPUs);
              --distribute allgather  every rank generates 1/G of the images, one all-gather instead of the broadcast;
              --shard rows  one render split into row bands (no collective).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x1F1F
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3  # dense fp32 vector = fp32 matrix peak: 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0

# BASELINE.json configs (index = position in `configs`, 1-based like SURVEY.md §8(d)); parameters: reference README.md:7 for the
# 8×8 grids, scripts/focusMapCompare.sh:1-5 (-s 7, focus / range / aspect table) for the focus sweep of config 5
CONFIGS = {
    2: dict(cols=8, rows=8, W=1920, H=1080, views=64, traj="0.0,0.0,1.0,1.0", focus=0.23, rng=0.0, aspect=1.783, effect=3.0),
    3: dict(cols=15, rows=15, W=1920, H=1080, views=45, traj="0,0.5,1,0.5", focus=0.06, rng=0.0, aspect=2.276, effect=3.0),
    4: dict(cols=8, rows=8, W=3840, H=2160, views=256, traj="0.0,0.0,1.0,1.0", focus=0.23, rng=0.0, aspect=1.783, effect=3.0),
    5: dict(cols=15, rows=15, W=3840, H=2160, views=64, traj="0.071,0.071,0.93,0.93", focus=0.22, rng=0.17, aspect=1.783, effect=7.0),
}


def cpu_baseline(cfg, hp, threads: int, sample_views: int | None = None, budget_s: float = 12.0, gpu_rows: dict | None = None) -> dict:
    """The oracle's scalar STD blend on the same workload, on `threads` host cores: whole steps (all views), or — `sample_views` — the
    first few full-frame views of the step (the single-threaded leg renders min(V, 4) views).
    gpu_rows = {view: (y0, y1, rows of the view the timed GPU launches rendered)}: while the oracle and its host copy of the grid are at
    hand, also state how the TEN_WM bytes compare with the oracle's model of the reference's half accumulators (M16): `ten_wm_vs_m16`."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import lfi_oracle_c as oc
    oc.build()
    n, W, H = cfg["cols"] * cfg["rows"], cfg["W"], cfg["H"]
    lf = np.empty((n, H, W, 4), dtype=np.uint8)
    with ThreadPoolExecutor(max_workers=max(threads, 8)) as ex:
        list(ex.map(lambda g: lf.__setitem__(g, oc.synthetic_plane(g, W, H, SEED)), range(n)))
    v = hp.weights.shape[0] if sample_views is None else min(sample_views, hp.weights.shape[0])
    # a bounded sample of the same workload: passes until ≈ budget_s of CPU work (threads × wall) have been spent, at most 8
    steps, t0 = 0, time.perf_counter()
    while steps < 8 and (steps == 0 or (time.perf_counter() - t0) * threads < budget_s):
        oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=0, v1=v, threads=threads)
        steps += 1
    dt = (time.perf_counter() - t0) / steps
    res = {"value": v / dt, "unit": "views/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{steps} pass(es): {v} views of {W}x{H} from {n} images each, scalar fp32 FMA weighted mean "
                     f"(oracle STD), {dt:.2f} s wall per pass, {dt * steps * threads:.0f} s of CPU work",
           "gpix_per_s": v * W * H / dt / 1e9}
    if gpu_rows:
        same = total = worst = 0
        for view, (y0, y1, got) in gpu_rows.items():
            want = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=view, v1=view + 1, rows=(y0, y1), threads=threads, model=oc.TEN_M16)[view, y0:y1]
            d = np.abs(got[..., :3].astype(np.int16) - want[..., :3].astype(np.int16))
            same, total, worst = same + int((d == 0).sum()), total + d.size, max(worst, int(d.max()))
        res["ten_wm_vs_m16"] = {"exact_match_fraction": same / total, "max_abs_diff_lsb": worst, "bytes_compared": total,
                                "rows": {str(v): [y0, y1] for v, (y0, y1, _) in gpu_rows.items()}}
    return res


def b_alg(W, rows_in, n_images, rows_out, views):
    """every RGBA8 input plane read once, every RGBA8 output plane written once."""
    return 4.0 * W * (rows_in * n_images + rows_out * views)


def timed(ctx, fn, iters: int, warm: int = 3, rounds: int = 3, warm_ms: float = 25.0, prepare=None) -> float:
    """ms per call of fn(): HIP events on the context's launch stream around `iters` back-to-back calls — the median of `rounds` such
    measurements, after at least `warm` calls and `warm_ms` of work (a section that starts on an idle GPU otherwise times its clock ramp).
    prepare = (method,) or (method, all_focus): lfi_prepare first — the derived input copy built and tuned for the current offsets
    outside the timed launches (a render retunes by itself only after 32 launches of one parameter set)."""
    if prepare:
        ctx.prepare(*prepare)
    t0, n = time.perf_counter(), 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        ctx.sync()
    res = []
    for _ in range(rounds):
        ctx.timer_start()
        for _ in range(iters):
            fn()
        res.append(ctx.timer_stop() / iters)
    return sorted(res)[len(res) // 2]


def also_table(L, device_index: int, iters: int, layout: str) -> dict:
    """The rest of BASELINE.json's configurations on ONE GPU, after the headline step (HIP-event times, inputs resident)."""
    import numpy as np
    out = {}

    def entry(cfg, ms, views, kernel, note=None, flops_bound=None, n_images=None, in_bytes_extra=0.0, out_bpp=None):
        n = n_images if n_images is not None else cfg["cols"] * cfg["rows"]
        ba = b_alg(cfg["W"], cfg["H"], n, cfg["H"], views) + in_bytes_extra
        fa = 6.0 * n * views * cfg["W"] * cfg["H"]
        # `frac` as on the headline: the bytes the kernel has to MOVE in the layouts in use — 3 B per pixel and image for the kernels that
        # read the alpha-free planar copy, 3 B per pixel and view for blend_p3's planar views, 4 B on RGBA sides — ÷ time ÷ 8 TB/s; the
        # fraction on SURVEY.md's 4·W·H·(N + V) is kept as `frac_algorithmic` (round 3 printed that one as `frac` here: ADVICE r3)
        in_bpp = 3 if any(k in kernel for k in ("blend_p3", "blend_planar", "blend_stdx<")) else 4
        if out_bpp is None:  # blend_p3 writes planar views (3 B) or, named so, RGBA views; blend_stdx writes either: its callers say which
            out_bpp = 3 if ("blend_p3" in kernel and ",rgba>" not in kernel) else 4
        bm = 1.0 * cfg["W"] * cfg["H"] * (in_bpp * n + out_bpp * views) + in_bytes_extra
        e = {"workload": f"{cfg['cols']}x{cfg['rows']} LF @{cfg['W']}x{cfg['H']}, {views} views", "kernel": kernel, "ms": ms,
             "views_per_s": views / ms * 1e3, "algorithmic_bytes": ba, "moved_bytes": bm, "hbm_gbs": bm / ms / 1e6, "frac": bm / ms / 1e6 / HBM_PEAK_GBS,
             "frac_algorithmic": ba / ms / 1e6 / HBM_PEAK_GBS}
        if flops_bound:
            e["bound"] = "fp32"
            e["fp32_tflops"] = fa / ms / 1e9
            e["fp32_frac"] = fa / ms / 1e9 / FP32_PEAK_TFLOPS
        if note:
            e["note"] = note
        return e

    def make_ctx(cfg, views=None, rng=None):
        ctx = L.Context(device_index)
        ctx.set_grid(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"])
        ctx.fill_synthetic(SEED)
        hp = L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], cfg["focus"], cfg["rng"] if rng is None else rng,
                            cfg["effect"], cfg["aspect"], views or cfg["views"])
        ctx.set_params(hp)
        return ctx, hp

    def guarded(section):
        """One configuration's measurements; a failure (e.g. out of memory on a smaller device) is recorded, not raised."""
        try:
            section()
        except Exception as e:
            out[section.__name__ + "_error"] = f"{type(e).__name__}: {e}"

    def fixed_focus_sweep(ctx, cfg, key):
        """A fixed-focus -f sweep (scripts/focusMapCompare.sh varies -f per run; loadGPUOffsets, src/interpolator.cu:226-246): 16 parameter
        sets around the configuration's focus, lfi_set_params + lfi_render each, NO lfi_prepare — every render has new integer offsets, so the
        derived planar copy keeps the per-image phases it was built with (stale: DESIGN.md 3) — per step, next to the same loop over ONE
        parameter set (what lfi_set_params itself adds to a tuned launch).  TEN_WM in the bench's view layout, STD likewise."""
        if layout != "rgba":
            ctx.set_output_layout(layout)
        sets = [L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], f, 0.0, cfg["effect"], cfg["aspect"], cfg["views"])
                for f in np.linspace(cfg["focus"] - 0.02, cfg["focus"] + 0.02, 16)]
        centre = L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], cfg["focus"], 0.0, cfg["effect"], cfg["aspect"], cfg["views"])
        for method, suffix in (("TEN_WM", ""), ("STD", "_std")):
            def loop(params):
                for hp_f in params:
                    ctx.set_params(hp_f)
                    ctx.render(method)
            ctx.set_params(centre)
            ctx.prepare(method)
            same = timed(ctx, lambda: loop([centre] * len(sets)), 1, warm=1, rounds=3) / len(sets)
            loop(sets)                                                     # the copy's padding grows to the sweep's largest offset once
            ms = timed(ctx, lambda: loop(sets), 1, warm=1, rounds=3) / len(sets)
            e = entry(cfg, ms, cfg["views"], "lfi_set_params + " + ctx.last_kernel_name(),
                      "per step of a 16-step fixed-focus sweep (-f ±0.02 around the configuration's): new offsets every render, the planar copy's phases stale",
                      out_bpp=3 if layout != "rgba" else 4)
            e["same_parameters_ms"] = same
            out[f"{key}_fixed_focus_sweep_step{suffix}"] = e
        ctx.set_params(centre)
        ctx.set_output_layout("rgba")

    # ---- config 2: the reference-layout (RGBA) output, STD, the non-tensor wavefront kernel ---------------------------------------
    def config2():
        c2 = CONFIGS[2]
        ctx, hp = make_ctx(c2)
        if layout != "rgba":
            ms = timed(ctx, lambda: ctx.render("TEN_WM"), iters, prepare=("TEN_WM",))
            out["config2_ten_wm_rgba_views"] = entry(c2, ms, 64, ctx.last_kernel_name(), "views stored as RGBA planes (the reference's layout)")
            # the headline kernel with every launch walking the image in the same direction (what a single cold launch does): the
            # default alternates the direction between consecutive launches, which re-reads part of the inputs from the Infinity Cache
            ctx.set_output_layout(layout)
            ctx.set_params(hp, flags=L.LFI_FLAG_SINGLE_SWEEP_DIRECTION)
            ms = timed(ctx, lambda: ctx.render("TEN_WM"), iters, prepare=("TEN_WM",))
            out["config2_ten_wm_single_sweep_direction"] = entry(c2, ms, 64, ctx.last_kernel_name(), "LFI_FLAG_SINGLE_SWEEP_DIRECTION: no Infinity Cache reuse between launches")
            ctx.set_params(hp)
            ctx.set_output_layout("rgba")
        ms = timed(ctx, lambda: ctx.render("STD"), iters, prepare=("STD",))
        out["config2_std"] = entry(c2, ms, 64, ctx.last_kernel_name(), "bit-exact STD (fp16 MFMA sum + exact fmaf chain inside the rounding band)")
        ctx.set_variant("STD", "wave_m2_nt")
        ms = timed(ctx, lambda: ctx.render("STD"), iters, prepare=("STD",))
        out["config2_std_exact_mfma"] = entry(c2, ms, 64, ctx.last_kernel_name(), "exact fp32 on v_mfma_f32_32x32x2_f32", flops_bound=True)
        ctx.set_variant("STD", "vfma")
        ms = timed(ctx, lambda: ctx.render("STD"), max(2, iters // 2), warm=1)
        out["config2_std_valu"] = entry(c2, ms, 64, ctx.last_kernel_name(), "the non-tensor wavefront kernel: one pass over the inputs, v_pk_fma_f32 chains, weights in SGPRs", flops_bound=True)
        ctx.set_variant("STD", "auto")
        fixed_focus_sweep(ctx, c2, "config2")
        ctx.close()

    guarded(config2)

    # ---- config 3: 15×15 @1080p, 45-view quilt -------------------------------------------------------------------------------------
    def config3():
        c3 = CONFIGS[3]
        ctx, hp = make_ctx(c3)
        if layout != "rgba":
            ctx.set_output_layout(layout)
        ms = timed(ctx, lambda: ctx.render("TEN_WM"), iters, prepare=("TEN_WM",))
        out["config3"] = entry(c3, ms, 45, ctx.last_kernel_name())
        ctx.set_output_layout("rgba")
        if layout != "rgba":
            ms = timed(ctx, lambda: ctx.render("TEN_WM"), iters, prepare=("TEN_WM",))
            out["config3_rgba_views"] = entry(c3, ms, 45, ctx.last_kernel_name(), "views stored as RGBA planes (the reference's layout)")
        ms = timed(ctx, lambda: ctx.render("STD"), max(2, iters // 2), prepare=("STD",))
        out["config3_std"] = entry(c3, ms, 45, ctx.last_kernel_name(), "bit-exact STD on a 15x15 grid (four chunks of images): fp16 MFMA sums + the exact chain inside the rounding band")
        ctx.close()

    guarded(config3)

    # ---- config 4: 8×8 @4K, per rank (32 of 256 views) and whole (256 views on this GPU) -------------------------------------------
    def config4():
        c4 = CONFIGS[4]
        hp_r, v0, v1 = L.rank_params(c4["cols"], c4["rows"], c4["W"], c4["H"], c4["traj"], c4["focus"], 0.0, c4["effect"], c4["aspect"], 256, 8, 3)
        ctx = L.Context(device_index)
        ctx.set_grid(c4["cols"], c4["rows"], c4["W"], c4["H"])
        ctx.fill_synthetic(SEED)
        ctx.set_params(hp_r)
        if layout != "rgba":
            ctx.set_output_layout(layout)
        ms = timed(ctx, lambda: ctx.render("TEN_WM"), iters, prepare=("TEN_WM",))
        out["config4_rank"] = entry(c4, ms, v1 - v0, ctx.last_kernel_name(), f"rank 3 of 8: views [{v0},{v1}) of the 256-view trajectory")
        # what one rank of a G-GPU run of config 4 launches, for the prediction of the 1/2/4/8 curve in DESIGN.md §8 (the 8-GPU rank is the
        # entry above; G = 1 is config4_whole_1gpu below): aggregate views/s = 256 / (this time), if the ranks do not disturb each other
        for G in (2, 4):
            hp_g, g0, g1 = L.rank_params(c4["cols"], c4["rows"], c4["W"], c4["H"], c4["traj"], c4["focus"], 0.0, c4["effect"], c4["aspect"], 256, G, G // 2)
            ctx.set_params(hp_g)
            if layout != "rgba":
                ctx.set_output_layout(layout)
            ms_g = timed(ctx, lambda: ctx.render("TEN_WM"), max(2, iters // 2), prepare=("TEN_WM",))
            out[f"config4_rank_of_{G}"] = entry(c4, ms_g, g1 - g0, ctx.last_kernel_name(), f"rank {G // 2} of {G}: views [{g0},{g1}) of the 256-view trajectory")
        hp_all = L.build_params(c4["cols"], c4["rows"], c4["W"], c4["H"], c4["traj"], c4["focus"], 0.0, c4["effect"], c4["aspect"], 256)
        ctx.set_params(hp_all)
        if layout != "rgba":
            ctx.set_output_layout(layout)
        ms = timed(ctx, lambda: ctx.render("TEN_WM"), max(2, iters // 2), prepare=("TEN_WM",))
        out["config4_whole_1gpu"] = entry(c4, ms, 256, ctx.last_kernel_name(), "the whole 256-view trajectory on one GPU (inputs read once)")
        ctx.close()

    guarded(config4)

    # ---- config 5: 15×15 @4K: fixed focus, and end to end = focus map (estimate + filter) + all-focus render, MFMA vs STD -----------
    def config5():
        c5 = CONFIGS[5]
        ctx, hp = make_ctx(c5)
        if layout != "rgba":
            ctx.set_output_layout(layout)
        ms = timed(ctx, lambda: ctx.render("TEN_WM"), max(2, iters // 2), prepare=("TEN_WM",))
        out["config5_fixed_focus"] = entry(c5, ms, 64, ctx.last_kernel_name())
        if layout != "rgba":
            ms = timed(ctx, lambda: ctx.render("STD"), max(2, iters // 4), warm=1, prepare=("STD",))
            out["config5_fixed_focus_std_planar_views"] = entry(c5, ms, 64, ctx.last_kernel_name(), "bit-exact STD into the planar views (written by the kernel itself)", out_bpp=3)
        ctx.set_output_layout("rgba")
        if layout != "rgba":
            ms = timed(ctx, lambda: ctx.render("TEN_WM"), max(2, iters // 2), prepare=("TEN_WM",))
            out["config5_fixed_focus_rgba_views"] = entry(c5, ms, 64, ctx.last_kernel_name(), "views stored as RGBA planes (the reference's layout)")
        # BASELINE config 5's comparison, fixed focus: bit-exact STD by the default kernel and by the NON-TENSOR wavefront kernel
        # (blend_std_vfma, the analogue of Standard::process, src/kernels.cu:312-342)
        ms = timed(ctx, lambda: ctx.render("STD"), max(2, iters // 4), warm=1, prepare=("STD",))
        out["config5_fixed_focus_std"] = entry(c5, ms, 64, ctx.last_kernel_name(), "bit-exact STD, default kernel")
        ctx.set_variant("STD", "vfma")
        ms = timed(ctx, lambda: ctx.render("STD"), 2, warm=1, rounds=2)
        out["config5_fixed_focus_std_nontensor"] = entry(c5, ms, 64, ctx.last_kernel_name(), "the non-tensor wavefront kernel (v_pk_fma_f32 chains)", flops_bound=True)
        ctx.set_variant("STD", "auto")
        fixed_focus_sweep(ctx, c5, "config5")
        ctx.set_params(hp)
        # the focus sweep on a STRUCTURED light field: a texture seen at a piecewise-constant focus inside
        # [focus, focus + range], so that the estimated map is piecewise constant as on real scenes — on hash noise the map is noise and
        # the all-focus gathers touch one cache line per pixel (measured: 17 ms instead of ≈2 ms), which no real input does
        ctx.fill_synthetic_scene(SEED)
        map_in = 4.0 * c5["W"] * c5["H"] * len(hp.focus_map_ids)          # the ≤32 sampled planes, read once by the estimate
        map_io = 4.0 * c5["W"] * c5["H"] * 3                                # map 0 written + read, map 1 written
        ms_map = timed(ctx, lambda: ctx.focus_map(), max(2, iters // 4), warm=1)
        out["config5_focus_map"] = {"workload": "15x15 LF @3840x2160, focus map (estimate over 32 views x 32 candidates x 9 taps + filter)",
                                    "kernel": "focus_factored passes + focus_filter", "ms": ms_map,
                                    "algorithmic_bytes": map_in + map_io, "frac": (map_in + map_io) / ms_map / 1e6 / HBM_PEAK_GBS,
                                    "note": "the range pass is LDS-bound (DESIGN.md 4.3), not HBM-bound: the fraction is reported for completeness"}
        # the same call when the inputs changed since the last one: the padded copies of the sampled images are rebuilt (otherwise kept
        # between calls — a focus sweep over one light field, which is what BASELINE config 5 is, pads once)
        ms_map_cold = timed(ctx, lambda: (ctx.grid_modified(), ctx.focus_map()), max(2, iters // 4), warm=1)
        out["config5_focus_map_inputs_changed"] = {"workload": out["config5_focus_map"]["workload"], "kernel": "focus_pad + focus_factored passes + focus_filter",
                                                   "ms": ms_map_cold, "algorithmic_bytes": map_in + map_io,
                                                   "frac": (map_in + map_io) / ms_map_cold / 1e6 / HBM_PEAK_GBS,
                                                   "note": "lfi_grid_modified before every call: every estimate pads its sampled images again (round 2's behaviour)"}
        for method in ("TEN_WM", "STD"):
            ms_r = timed(ctx, lambda: ctx.render(method, all_focus=True), max(2, iters // 4), warm=1)
            k = ctx.last_kernel_name()
            ms_e = timed(ctx, lambda: (ctx.focus_map(), ctx.render(method, all_focus=True)), max(2, iters // 4), warm=1)
            key = "config5_allfocus_" + method.lower()
            out[key + "_render"] = entry(c5, ms_r, 64, k, "all-focus render from a resident focus map (structured light field)")
            out[key + "_end_to_end"] = entry(c5, ms_e, 64, "focus map + " + k, "lfi_focus_map + all-focus render per iteration (-r 0.17)",
                                             in_bytes_extra=map_in + map_io)
        # BASELINE config 5 as the reference's focusMapCompare.sh uses it: a focus SWEEP over the resident light field — per step NEW parameters
        # (lfi_set_params with another -f: stream-ordered upload, no synchronisation), the focus map, one all-focus render.  16 steps over
        # -f 0.20 … 0.24 (around the focus the scene was built for), host parameter arithmetic outside the timed region.
        sweep = [L.build_params(c5["cols"], c5["rows"], c5["W"], c5["H"], c5["traj"], f, c5["rng"], c5["effect"], c5["aspect"], c5["views"])
                 for f in np.linspace(0.20, 0.24, 16)]

        def sweep_pass():
            for hp_f in sweep:
                ctx.set_params(hp_f)
                ctx.focus_map()
                ctx.render("TEN_WM", all_focus=True)

        sweep_pass()                                                           # the padded planes grow to the sweep's largest shift once
        ms_sweep = timed(ctx, sweep_pass, 1, warm=1, rounds=3) / len(sweep)
        out["config5_focus_sweep_step"] = entry(c5, ms_sweep, 64, "lfi_set_params + focus map + " + ctx.last_kernel_name(),
                                                "per step of a 16-step focus sweep (-f 0.20 … 0.24): new parameters, focus map, all-focus TEN_WM render",
                                                in_bytes_extra=map_in + map_io)
        ctx.set_params(hp)
        ctx.set_variant("STD", "vfma")
        ms_r = timed(ctx, lambda: ctx.render("STD", all_focus=True), 2, warm=1, rounds=2)
        out["config5_allfocus_std_nontensor"] = entry(c5, ms_r, 64, ctx.last_kernel_name(), "all-focus render by the non-tensor wavefront kernel", flops_bound=True)
        ctx.close()

    guarded(config5)

    # ---- cold, one-shot: inputs resident in HBM, caches flushed → everything the library derives from them + the FIRST render ------
    # (the reference's actual flow minus its 100-launch benchmark loop, src/interpolator.cu:248-297; every other number on this line is a
    # steady-state launch with the derived input copy in place and the sweep direction alternating)
    def cold_one_shot():
        import torch
        flush = torch.empty(768 << 20, dtype=torch.uint8, device=f"cuda:{device_index}")   # 3× the Infinity Cache
        for key, cfg in (("config2", CONFIGS[2]), ("config5", CONFIGS[5])):
            res = []
            for rep in range(3):
                ctx, hp = make_ctx(cfg, rng=0.0)
                ctx.set_params(hp, flags=L.LFI_FLAG_SINGLE_SWEEP_DIRECTION)
                if layout != "rgba":
                    ctx.set_output_layout(layout)
                ctx.sync()
                flush.fill_(rep)
                torch.cuda.synchronize()
                ctx.timer_start()
                ctx.render("TEN_WM")
                res.append(ctx.timer_stop())
                k = ctx.last_kernel_name()
                mem = ctx.memory_info()
                ctx.close()
            out[key + "_cold_one_shot"] = entry(cfg, sorted(res)[1], cfg["views"], k,
                                                f"a fresh context's first render, caches flushed, one sweep direction: includes whatever the "
                                                f"library derives from the inputs first ({mem.derived_bytes / 1e6:.0f} MB derived copy); median of 3")
            out[key + "_cold_one_shot"]["reps_ms"] = [round(r, 3) for r in res]   # a first hipMalloc of gigabytes can take 100+ ms on a box
        # The same first render when the application said at load time what it will render (lfi_prepare: the derived copy is built as the
        # images arrive, outside any render) and released the RGBA planes (lfi_release_inputs: the copy is the only copy of the inputs):
        # a launch on cold caches, and the inputs' footprint
        cfg = CONFIGS[5]
        res = []
        for rep in range(3):
            ctx, hp = make_ctx(cfg, rng=0.0)
            ctx.set_params(hp, flags=L.LFI_FLAG_SINGLE_SWEEP_DIRECTION)
            if layout != "rgba":
                ctx.set_output_layout(layout)
            before = ctx.memory_info()
            ctx.prepare("TEN_WM")
            ctx.release_inputs()
            ctx.sync()
            flush.fill_(rep)
            torch.cuda.synchronize()
            ctx.timer_start()
            ctx.render("TEN_WM")
            res.append(ctx.timer_stop())
            k = ctx.last_kernel_name()
            after = ctx.memory_info()
            if rep == 2:
                steady = timed(ctx, lambda: ctx.render("TEN_WM"), max(2, iters // 2))
            ctx.close()
        e = entry(cfg, sorted(res)[1], cfg["views"], k, "a prepared context's first render (lfi_prepare + lfi_release_inputs at load time), caches flushed, one sweep direction; median of 3")
        e["reps_ms"] = [round(r, 3) for r in res]
        e["inputs_bytes_rgba"] = before.grid_bytes
        e["inputs_bytes_after_release"] = after.grid_bytes + after.derived_bytes
        e["inputs_footprint_vs_rgba"] = (after.grid_bytes + after.derived_bytes) / before.grid_bytes
        e["steady_ms_released"] = steady
        out["config5_cold_first_render_prepared"] = e
        del flush

    guarded(cold_one_shot)

    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--method", default="TEN_WM", choices=["TEN_WM", "STD"])
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="2 (default): BASELINE config 2 per GPU, weak-scaled — the contract's run; 4: BASELINE config 4, strong-scaled — "
                         "ONE 256-view trajectory @4K split over the GPUs (V/G views per rank), every rank holds the whole 8x8 grid")
    ap.add_argument("--distribute", default="broadcast", choices=["broadcast", "allgather"],
                    help="how the grid reaches every GPU before the timed region: broadcast from rank 0 (north_star), or every rank "
                         "produces 1/G of the images and ONE all-gather assembles them (each xGMI link carries 1/G: SURVEY.md §5)")
    ap.add_argument("--layout", default="planar", choices=["rgba", "planar"],
                    help="device layout of the rendered views: planar (default) = alpha-free byte planes, the library's opt-in layout "
                         "(the alpha the reference writes is the constant 255 — src/kernels.cu:393 — and is re-created on download, so "
                         "what a caller downloads is byte-identical); rgba = the reference's RGBA planes (also timed, in `also`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the table of the other BASELINE configurations (N = 1 only)")
    ap.add_argument("--also-iters", type=int, default=12)
    ap.add_argument("--shard", default="views", choices=["views", "rows"],
                    help="views (default): every GPU holds the whole grid and renders its views; rows: strong scaling of ONE render — each "
                         "GPU renders a band of rows and holds only the input rows the band's warp reaches (SURVEY.md §8(f).2)")
    ap.add_argument("--prewarm-ms", type=float, default=500.0,
                    help="untimed launches during set-up, before the W warm-up steps, so that the clocks have ramped "
                         "(with 5 warm-up launches = 1 ms of work the first timed launches still run at idle clocks; the first process "
                         "on a fresh box measured 0.167 ms per step after 150 ms of them and 0.150 ms after 600 ms)")
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

    # the host driver of this pool supports dmabuf IPC only: RCCL's intra-node transport needs it (the image exports it already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    # fail fast, with a message that says what is missing (a rank that dies inside set_device / NCCL init otherwise reads as a hang or a
    # stack trace from the launcher): one GPU per rank …
    n_dev = torch.cuda.device_count()
    if not rehearse and n_dev < max(world, local_rank + 1):
        print(f"bench.py: --gpus {world} needs {world} visible GPUs, this node shows {n_dev} (rank {rank}, LOCAL_RANK {local_rank}); "
              f"run with --gpus {n_dev}, or set LFI_BENCH_REHEARSE=1 to rehearse the ranks on one GPU over gloo", file=sys.stderr)
        return 3
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    # LFI_BENCH_FORCE_DIST=1: run the process group, the collectives and the barriers with ONE rank too — the only way to execute the
    # RCCL code path (init with device_id, broadcast / all-gather of the planes, MAX all-reduce) on a one-GPU box
    collectives = world > 1 or os.environ.get("LFI_BENCH_FORCE_DIST") == "1"
    if collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            if rehearse:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                # … and a communicator that works: RCCL creates it lazily, so run one tiny all-reduce NOW — a node whose ranks cannot reach each
                # other (IPC mode, missing xGMI / PCIe peer access, a dead rank) fails here, in set-up, not in the middle of the timed region
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all-reduce over {world} ranks returned {probe.item()}")
        except Exception as e:
            print(f"bench.py: rank {rank} of {world} could not initialise the process group over {'gloo' if rehearse else 'RCCL (backend nccl)'}: "
                  f"{type(e).__name__}: {e}  [MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}]", file=sys.stderr)
            return 4

    cfg = CONFIGS[args.config]
    COLS, ROWS, WIDTH, HEIGHT = cfg["cols"], cfg["rows"], cfg["W"], cfg["H"]
    n_images = COLS * ROWS
    strong = args.config == 4 or args.shard == "rows"
    if args.config == 4:
        total_views = cfg["views"]                       # one trajectory, split over the ranks
        if total_views % world:
            print("bench.py: --config 4 needs a GPU count that divides 256", file=sys.stderr)
            return 2
        views_per_gpu = total_views // world
    else:
        views_per_gpu = cfg["views"]
        total_views = views_per_gpu * (world if args.shard == "views" else 1)

    ctx = L.Context(device_index)
    ctx.set_grid(COLS, ROWS, WIDTH, HEIGHT)
    stream = torch.cuda.Stream(device=dev)
    distribute_ms = 0.0
    if args.shard == "views":
        # input planes live in a torch tensor so that RCCL (torch.distributed "nccl") can move them; padded so that it splits into
        # `world` equal parts for the all-gather
        plane = HEIGHT * WIDTH * 4
        per_rank = -(-n_images // world)
        flat = torch.empty((per_rank * world * plane,), dtype=torch.uint8, device=dev)
        grid = flat[: n_images * plane].view(n_images, HEIGHT, WIDTH, 4)
        ctx.attach_grid(grid.data_ptr(), grid.numel())
        ctx.set_stream(stream.cuda_stream)
        g0, g1 = min(rank * per_rank, n_images), min((rank + 1) * per_rank, n_images)
        if args.distribute == "broadcast":
            if rank == 0:
                ctx.fill_synthetic(SEED)
        else:
            ctx.fill_synthetic(SEED, g0, g1)             # this rank's slice only (real data: this rank's share of the uploads)
        ctx.sync()
        torch.cuda.synchronize()
        if collectives:
            dist.barrier()
            t_d = time.perf_counter()
            if args.distribute == "broadcast":
                L.broadcast_grid(grid, src=0)            # the one collective of the job over xGMI (RCCL), outside the timed region
            else:
                L.allgather_grid(flat, rank, world)      # in place: every rank contributes its 1/G, every link carries 1/G
            torch.cuda.synchronize()
            dist.barrier()
            distribute_ms = (time.perf_counter() - t_d) * 1e3
        ctx.grid_modified()              # the attached planes were written by the collective, not through the ABI
        # host parameters for the whole trajectory; each rank keeps its own rows of the weight matrix
        hp, v0, v1 = L.rank_params(COLS, ROWS, WIDTH, HEIGHT, cfg["traj"], cfg["focus"], 0.0, cfg["effect"], cfg["aspect"], total_views, world, rank)
        assert v1 - v0 == views_per_gpu
        in_rows_n = out_rows_n = HEIGHT
    else:
        # one render split into row bands: no collective at all with synthetic data (every rank generates the rows it
        # holds; real data would be scattered band + halo per rank)
        hp = L.build_params(COLS, ROWS, WIDTH, HEIGHT, cfg["traj"], cfg["focus"], 0.0, cfg["effect"], cfg["aspect"], total_views)
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
    out_bpp = 4 if args.layout == "rgba" else 3
    if args.layout != "rgba":
        ctx.set_output_layout(args.layout)
    views = None
    if args.layout == "rgba":
        # RGBA views in a caller-owned torch tensor (the interop path); the planar layout keeps the library's own allocation, which
        views = torch.empty((views_per_gpu * ctx.view_layout().view_stride_bytes,), dtype=torch.uint8, device=dev)
        ctx.attach_views(views.data_ptr(), views.numel())
    ctx.set_variant(args.method, args.variant)
    ctx.prepare(args.method)             # the derived planar input copy is built (and timed) here, not in the first launch
    mem = ctx.memory_info()

    def barrier():
        if collectives:
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
    kernel_name = ctx.last_kernel_name()

    t = torch.tensor([elapsed, kernel_ms / 1e3], dtype=torch.float64, device=dev)
    if collectives:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max, kernel_s_max = float(t[0]), float(t[1])

    # cheap sanity check that the timed launches rendered something: alpha 255 everywhere, RGB not constant
    sample = ctx.download_view(0)[ctx.out_rows[0] + out_rows_n // 2, :64]
    assert (sample[:, 3] == 255).all() and sample[:, :3].std() > 0, "render produced no image"
    gpu_rows = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 2 and args.method == "TEN_WM" and args.shard == "views":
        # 48 rows (top edge, middle, bottom edge) of three views of the timed launches, for cpu_baseline's ten_wm_vs_m16
        gpu_rows = {}
        for view, y0 in ((0, 0), (views_per_gpu // 2, HEIGHT // 2 - 8), (views_per_gpu - 1, HEIGHT - 16)):
            gpu_rows[view] = (y0, y0 + 16, ctx.download_view(view)[y0:y0 + 16].copy())

    if rank == 0:
        value = total_views * args.steps / elapsed_max
        # bytes per launch on this GPU: rows held of every input plane + rows rendered of every view
        balg = b_alg(WIDTH, in_rows_n, n_images, out_rows_n, views_per_gpu)
        reads_planar = kernel_name.startswith("blend_planar") or kernel_name.startswith("blend_p3")
        b_moved = 1.0 * WIDTH * ((3 if reads_planar else 4) * in_rows_n * n_images + out_bpp * out_rows_n * views_per_gpu)
        t_launch = kernel_s_max / args.steps
        achieved = balg / t_launch / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")           # PMC-derived HBM bytes per launch, if measured
        if os.path.exists(tpath) and args.config == 2 and args.shard == "views":
            try:
                traffic = json.load(open(tpath)).get(f"{args.method}/{args.variant}/{args.layout}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        f_alg = 6.0 * n_images * views_per_gpu * WIDTH * out_rows_n   # 3 channels × (multiply + add)
        how = {"broadcast": "grid broadcast once", "allgather": "grid all-gathered once from 1/G per rank"}[args.distribute]
        line = {
            "metric": f"novel views/sec + Gpix/sec, {COLS}x{ROWS} LF @{'1080p' if HEIGHT == 1080 else '4K'} {args.method}",
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.method == "TEN_WM" else "f32", "data": "synthetic",
            "config": {"workload": f"{COLS}x{ROWS} LF @{WIDTH}x{HEIGHT}, {views_per_gpu}-view -t trajectory per GPU"
                                   + (f" ({total_views} views in all)" if world > 1 else "")
                                   + f", -m {args.method}, -f {cfg['focus']} -a {cfg['aspect']} -s {cfg['effect']:g}",
                       "baseline_config": args.config,
                       "views_per_gpu": views_per_gpu, "images": n_images, "variant": args.variant, "kernel": kernel_name,
                       "view_layout": ("RGBA planes (the reference's)" if args.layout == "rgba" else
                                       "alpha-free byte planes [view][R,G,B][H][W] (opt-in, lfi_set_output_layout; alpha = 255 is re-created "
                                       "on download)"),
                       "sweep": "consecutive launches walk the image in opposite directions (input rows read last are read first by the next "
                                "launch: Infinity Cache reuse; LFI_FLAG_SINGLE_SWEEP_DIRECTION disables, timed in `also`)",
                       "inputs": ("resident in HBM before the timed region: RGBA planes + the derived planar alpha-free copy the kernel "
                                  "reads (built once per change of the inputs, outside every render: DESIGN.md 4.1)"
                                  if reads_planar else "resident in HBM before the timed region: RGBA planes"),
                       "derived_copy_bytes": mem.derived_bytes, "derived_copy_build_ms": mem.derived_build_ms,
                       "grid_bytes": mem.grid_bytes,
                       "parallelism": (f"views sharded over {world} GPU(s), {how} ({'gloo rehearsal' if rehearse else 'RCCL'})"
                                       if args.shard == "views" else
                                       f"rows sharded over {world} GPU(s): {out_rows_n} output rows from {in_rows_n} input rows on rank 0, "
                                       "no collective"),
                       "distribute_ms": distribute_ms,
                       "view_ranges": ([list(L.view_range(total_views, world, r)) for r in range(world)] if args.shard == "views" else None),
                       "row_bands": ([list(L.row_band(HEIGHT, world, r)) for r in range(world)] if args.shard == "rows" else None),
                       "multi_gpu_hardware_runs": ("this line" if world > 1 and not rehearse else
                                                   "none by the builder: gpurun offers one GPU; N > 1 is covered by gloo tests and a one-GPU "
                                                   "rehearsal, the driver's SCALE run is the first RCCL execution")},
            "gpix_per_s": value * WIDTH * HEIGHT / 1e9,
            "prewarm_ms": args.prewarm_ms,
            "kernel_ms_per_launch": t_launch * 1e3,
            # `achieved` / `frac`: the bytes the kernel has to MOVE in the layouts in use (3 B per pixel on a side that is alpha-free,
            # 4 B on an RGBA side) ÷ the launch time — what the memory system delivered.  SURVEY.md §8(d)'s algorithmic figure
            # 4·W·H·(N + V) counts an alpha byte on both sides that the planar layouts never touch; a fraction on it (`frac_algorithmic`)
            # credits the kernel with bytes it did not move (round 2's 0.88 was that figure) and is kept for comparison only.
            "roofline": {"bound": "hbm", "achieved": b_moved / t_launch / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b_moved / t_launch / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiled constant: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/traffic.json "
                                           "(not measured in this run)" if traffic else None,
                         "bytes_per_launch": b_moved,
                         "bytes_definition": f"layout bytes: {3 if reads_planar else 4} B per pixel and image read, {out_bpp} B per pixel and view written",
                         "algorithmic_bytes_per_launch": balg, "achieved_algorithmic": achieved, "frac_algorithmic": achieved / HBM_PEAK_GBS,
                         "mfma_frac_of_2500_tflops": f_alg / t_launch / 1e12 / F16_MFMA_PEAK_TFLOPS},
        }
        if args.method == "STD" and args.variant not in ("auto", "filtered_m2_nt"):
            # the exact-fp32 kernels are bound by the fp32 FMA lanes (the default STD kernel computes on the fp16
            # matrix pipe and recomputes the sums near x.5 with the fmaf chain: HBM-bound like TEN_WM)
            tflops = f_alg / t_launch / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": tflops / FP32_PEAK_TFLOPS, "traffic": traffic,
                                "algorithmic_flops_per_launch": f_alg, "hbm_frac_of_8000_gbs": achieved / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            threads = min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(cfg, hp, threads, gpu_rows=gpu_rows)
            if "ten_wm_vs_m16" in line["cpu_baseline"]:
                m16 = line["cpu_baseline"]["ten_wm_vs_m16"]
                line["config"]["ten_wm_tolerance"] = ("u8 bytes within 1 LSB (= 3.9e-3 normalised) of M16, the oracle's model of the reference's half accumulators; "
                                                      "pre-quantisation within 1e-3 normalised of the exact fp64 blend (tests/test_gpu_parity.py)")
                line["config"]["ten_wm_exact_match_vs_m16"] = m16["exact_match_fraction"]
                line["config"]["ten_wm_max_abs_diff_lsb_vs_m16"] = m16["max_abs_diff_lsb"]
            # the same scalar code on ONE thread, min(V, 4) full-frame views
            line["cpu_baseline_1thread"] = cpu_baseline(cfg, hp, 1, sample_views=4, budget_s=8.0)

    ctx.close()
    del views, grid
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_also and args.config == 2 and args.shard == "views":
            try:
                detail = also_table(L, device_index, args.also_iters, args.layout)
            except Exception as e:      # the headline line must come out whatever happens to the extra configurations
                detail = {"error": f"{type(e).__name__}: {e}"}
            line["also_detail"] = detail
            # the headline's two companions (same kernel family, same inputs): one sweep direction only, and the reference's RGBA views
            for key, name in (("config2_ten_wm_single_sweep_direction", "single_sweep_direction_ms"), ("config2_ten_wm_rgba_views", "rgba_views_ms")):
                if isinstance(detail.get(key), dict):
                    line["roofline"][name] = detail[key]["ms"]
            # LAST on the line, compact, so that a reader who keeps only the tail of the output still has every configuration:
            # [ms per launch, fraction of the bound (HBM 8 TB/s on the bytes MOVED in the layouts in use — as the headline's roofline.frac —, or
            # fp32 157.3 TFLOP/s where the kernel is fp32-bound), kernel]; the fraction on 4·W·H·(N+V) bytes is also_detail's frac_algorithmic
            compact = {"legend": "[ms, frac of 8 TB/s on layout bytes moved | 'fp32:' frac of 157.3 TFLOP/s, kernel]"}
            for key, e in detail.items():
                if isinstance(e, dict) and "ms" in e:
                    fr = f"fp32:{e['fp32_frac']:.3f}" if e.get("bound") == "fp32" else round(e.get("frac", 0.0), 3)
                    compact[key] = [round(e["ms"], 4), fr, e.get("kernel", "")]
                else:
                    compact[key] = e
            if "cpu_baseline_1thread" in line:
                compact["cpu_baseline_1thread_views_per_s"] = round(line["cpu_baseline_1thread"]["value"], 2)
            if "cpu_baseline" in line:
                compact["cpu_baseline_views_per_s"] = [round(line["cpu_baseline"]["value"], 1), line["cpu_baseline"]["cores"]]
            line["also"] = compact
        print(json.dumps(line), flush=True)
    if collectives:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
