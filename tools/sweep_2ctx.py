"""Focus sweep at config 5 (15x15 @4K, all-focus TEN_WM): frames per second with ONE context (map, render, map, render … on one stream) and
with TWO contexts that share the light field (lfi_attach_grid) and take alternate frames — the map of frame i + 1 (L1-bound) beside the
render of frame i (DMA-bound).  usage: python tools/sweep_2ctx.py [frames]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = rows = 15; W, H, V = 3840, 2160, 64
hp = L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V)
a = L.Context(0); a.set_grid(cols, rows, W, H); a.set_params(hp); a.fill_synthetic_scene(0x1F1F); a.sync()
ptr, nbytes = a.grid_device_ptr()
b = L.Context(0); b.set_grid(cols, rows, W, H); b.attach_grid(ptr, nbytes); b.set_params(hp)
for method in ("TEN_WM", "STD"):
    for ctx in (a, b):
        for _ in range(2):
            ctx.focus_map(); ctx.render(method, all_focus=True)
        ctx.sync()
    for label, ctxs in (("one context", (a,)), ("two contexts", (a, b)), ("one context", (a,)), ("two contexts", (a, b))):
        t0 = time.perf_counter()
        for f in range(frames):
            c = ctxs[f % len(ctxs)]
            c.focus_map(); c.render(method, all_focus=True)
        for c in ctxs:
            c.sync()
        dt = (time.perf_counter() - t0) / frames
        print(f"{method} {label}: {dt*1e3:.3f} ms per frame ({frames} frames)", flush=True)
b.close(); a.close()
