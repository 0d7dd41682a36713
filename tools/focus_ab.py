"""Time lfi_focus_map (estimate + filter) at config 5's shape on the structured scene, and a few all-focus renders.
LFI_AB_LIB=<other build> for A/B runs on one box.   usage: python tools/focus_ab.py [iters] [cols W H]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import _ablib  # noqa: F401  (LFI_AB_LIB)
import lfinterpolator_amd as L

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 15
W = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
H = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
ctx = L.Context(0)
ctx.set_grid(cols, cols, W, H)
hp = L.build_params(cols, cols, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64)
ctx.set_params(hp)
ctx.fill_synthetic_scene(0x1F1F)
for _ in range(5):
    ctx.focus_map()
ctx.sync()
res = []
for _ in range(3):
    ctx.timer_start()
    for _ in range(iters):
        ctx.focus_map()
    res.append(ctx.timer_stop() / iters)
print(f"focus_map {cols}x{cols} @{W}x{H}: " + " ".join(f"{r:.3f}" for r in res) + " ms", flush=True)
for method in ("TEN_WM", "STD"):
    for _ in range(2):
        ctx.render(method, all_focus=True)
    ctx.sync()
    res = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(max(2, iters // 4)):
            ctx.render(method, all_focus=True)
        res.append(ctx.timer_stop() / max(2, iters // 4))
    print(f"all-focus {method} ({ctx.last_kernel_name()}): " + " ".join(f"{r:.3f}" for r in res) + " ms", flush=True)
ctx.close()
