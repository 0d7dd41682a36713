"""The RGBA-view kernels (the reference's own view layout) at BASELINE configs 2, 3, 5: blend_p3 with its RGBA epilogue (TEN_WM, default since round 4) against blend_planar (variant planar_m2_nt; STD by the band method on
8x8 grids) and blend_stdx (STD on 15x15 grids), back-to-back launch times.  LFI_AB_LIB for A/B."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
CFG = [("config 2", 8, 1920, 1080, 64, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0), ("config 4 rank", 8, 3840, 2160, 32, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0),
       ("config 4 whole", 8, 3840, 2160, 256, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0), ("config 3", 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       ("config 5", 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)]
for name, cols, W, H, V, traj, focus, aspect, effect in CFG:
    ctx = L.Context(0); ctx.set_grid(cols, cols, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(cols, cols, W, H, traj, focus, 0.0, effect, aspect, V))
    ctx.set_output_layout(os.environ.get("LFI_LAYOUT", "rgba"))   # LFI_LAYOUT=planar: the same renders into the planar view layout
    n = 30 if W == 1920 else 8
    for method, variant in (("TEN_WM", "auto"), ("TEN_WM", "planar_m2_nt"), ("STD", "auto")):
        ctx.set_variant(method, variant)
        ctx.prepare(method)
        for _ in range(2 * n): ctx.render(method)
        ctx.sync()
        res = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(n): ctx.render(method)
            res.append(ctx.timer_stop() / n)
        print(f"{name:14s} {method:6s} {variant:12s} {ctx.last_kernel_name():22s} " + " ".join(f"{r:.4f}" for r in sorted(res)) + " ms", flush=True)
    ctx.close()
