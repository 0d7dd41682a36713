"""BASELINE config 5 as it is used: a focus SWEEP over one resident light field — per step new parameters (lfi_set_params with another -f),
the focus map, one all-focus render.  Times a step (HIP events around the whole loop, host calls included) and counts the focus_pad launches
a rocprofv3 trace would show through lfi_memory_info... (the padded planes are rebuilt only when the shifts outgrow them).
method FIXED: no focus map, a fixed-focus TEN_WM render per step in the planar view layout (the derived planar copy must cover every step's offsets).
usage: python tools/focus_sweep.py [steps=16] [method=TEN_WM|STD|FIXED] [focus_lo focus_hi]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
method = sys.argv[2] if len(sys.argv) > 2 else "TEN_WM"
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
lo, hi = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.10, 0.30)
foci = np.linspace(lo, hi, steps)
hps = [L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", float(f), 0.17, 7.0, 1.783, V) for f in foci]   # host arithmetic outside the timing
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)          # the scene is built for focus 0.22 … 0.39; the sweep looks at it through other windows
fixed = method == "FIXED"
if fixed:
    ctx.set_output_layout("planar")

def step():
    if fixed:
        ctx.render("TEN_WM")
    else:
        ctx.focus_map(); ctx.render(method, all_focus=True)
for order, name in ((list(range(steps)), "ascending focus"), (list(range(steps))[::-1], "descending focus"), (list(range(steps)), "ascending again")):
    ctx.set_params(hps[order[0]]); step(); ctx.sync()
    t0 = time.perf_counter()
    ctx.timer_start()
    for i in order:
        ctx.set_params(hps[i])
        step()
    ms = ctx.timer_stop() / steps
    wall = (time.perf_counter() - t0) * 1e3 / steps
    print(f"{name:18s}: {ms:.3f} ms per step on the GPU, {wall:.3f} ms wall ({method}, {steps} steps, focus {foci[0]:.2f} … {foci[-1]:.2f})", flush=True)
ctx.close()
