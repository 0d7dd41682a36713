"""Config 4 whole on one GPU (8x8 @4K, 256 views, four view passes per LDS-resident tile), and config 2, timed.  LFI_AB_LIB for A/B."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
for (W, H, V) in ((3840, 2160, 256), (1920, 1080, 64)):
    ctx = L.Context(0); ctx.set_grid(8, 8, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(8, 8, W, H, "0.0,0.0,1.0,1.0", 0.23, 0.0, 3.0, 1.783, V))
    ctx.set_output_layout("planar"); ctx.prepare("TEN_WM")
    for _ in range(20): ctx.render("TEN_WM")
    ctx.sync()
    res = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(10): ctx.render("TEN_WM")
        res.append(ctx.timer_stop() / 10)
    print(f"{W}x{H} {V} views: {ctx.last_kernel_name()} " + " ".join(f"{r:.3f}" for r in sorted(res)) + " ms", flush=True)
    ctx.close()
