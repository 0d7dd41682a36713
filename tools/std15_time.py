"""Exact STD (fp32 matrix pipe) at 15x15: config 3 and config 5 fixed focus, RGBA layout.   usage: python tools/std15_time.py   (LFI_AB_LIB for A/B)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import lfinterpolator_amd as L
import _ablib
for cfg in ((15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0), (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)):
    cols, rows, W, H, V, traj, focus, aspect, effect = cfg
    ctx = L.Context(0)
    ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
    for _ in range(3): ctx.render("STD")
    ctx.sync()
    best = sorted(ctx.benchmark("STD", warmup=2, runs=6).back_to_back_ms for _ in range(3))
    print(f"{W}x{H} {V} views: {ctx.last_kernel_name()} {best[1]:.3f} ms", flush=True)
    ctx.close()
# the same with TEN_WM (RGBA views: blend_planar; planar views: blend_p3) for scale, and STD through the exact kernel
for cfg in ((15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0), (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)):
    cols, rows, W, H, V, traj, focus, aspect, effect = cfg
    ctx = L.Context(0)
    ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
    for what, method, variant, layout in (("TEN_WM rgba", "TEN_WM", "auto", "rgba"), ("TEN_WM planar", "TEN_WM", "auto", "planar"), ("STD planar views", "STD", "auto", "planar"),
                                          ("STD exact mfma", "STD", "wave_m2_nt", "rgba")):
        ctx.set_output_layout(layout)
        ctx.set_variant(method, variant)
        for _ in range(2): ctx.render(method)
        ctx.sync()
        best = sorted(ctx.benchmark(method, warmup=1, runs=4).back_to_back_ms for _ in range(3))
        print(f"{W}x{H} {V} views: {what}: {ctx.last_kernel_name()} {best[1]:.3f} ms", flush=True)
        ctx.set_variant(method, "auto")
    ctx.close()
