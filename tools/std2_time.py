"""STD at BASELINE config 2 (8x8 @1080p, 64 views, RGBA views: blend_planar<STDF>): back-to-back launch times.  LFI_AB_LIB for A/B."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
cols, W, H, V, traj, focus, aspect, effect = 8, 1920, 1080, 64, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0
ctx = L.Context(0); ctx.set_grid(cols, cols, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, cols, W, H, traj, focus, 0.0, effect, aspect, V))
ctx.prepare("STD")
for _ in range(60): ctx.render("STD")
ctx.sync()
res = []
for _ in range(5):
    ctx.timer_start()
    for _ in range(40): ctx.render("STD")
    res.append(ctx.timer_stop() / 40)
print(f"config 2 STD {ctx.last_kernel_name()} " + " ".join(f"{r:.4f}" for r in sorted(res)) + " ms", flush=True)
ctx.close()
