"""BASELINE config 4 as ONE launch on one GPU (8x8 @4K, 256 views) in the RGBA view layout: STD and TEN_WM, back-to-back launch times.  LFI_AB_LIB for A/B."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
cols, W, H, V, traj, focus, aspect, effect = 8, 3840, 2160, 256, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0
ctx = L.Context(0); ctx.set_grid(cols, cols, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, cols, W, H, traj, focus, 0.0, effect, aspect, V))
for method in ("STD", "TEN_WM"):
    ctx.prepare(method)
    for _ in range(4): ctx.render(method)
    ctx.sync()
    res = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(4): ctx.render(method)
        res.append(ctx.timer_stop() / 4)
    print(f"config 4 whole, RGBA views: {method:6s} {ctx.last_kernel_name():22s} " + " ".join(f"{r:.4f}" for r in sorted(res)) + " ms", flush=True)
ctx.close()
