"""Does the alignment of the 128-byte input runs matter to blend_p3?  Config 2 with the reference's offsets, and with the x offsets
replaced by (a) multiples of 128 (every run starts on a cache line), (b) multiples of 64, (c) odd values — same rows, same bytes.
usage: python tools/align_probe.py"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import _ablib  # noqa: F401
import lfinterpolator_amd as L

cols = rows = 8; W, H, V = 1920, 1080, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
hp = L.build_params(cols, rows, W, H, "0.0,0.0,1.0,1.0", 0.23, 0.0, 3.0, 1.783, V)
ctx.set_output_layout("planar")
base = hp.focused_offsets.copy()
def run(tag, ox):
    hp.focused_offsets = base.copy(); hp.focused_offsets[:, 0] = ox
    ctx.set_params(hp); ctx.prepare("TEN_WM")
    for _ in range(200): ctx.render("TEN_WM")
    ctx.sync()
    res = []
    for _ in range(5):
        ctx.timer_start()
        for _ in range(50): ctx.render("TEN_WM")
        res.append(ctx.timer_stop() / 50)
    print(f"{tag:34s} {ctx.last_kernel_name()}  " + " ".join(f"{r*1e3:.1f}" for r in sorted(res)) + " us", flush=True)
g = np.arange(64)
for rep in range(2):
    run("reference offsets", base[:, 0])
    run("multiples of 128", 128 * ((g % 3) - 1))
    run("multiples of 64 (odd multiples)", 64 * (2 * (g % 3) - 1))
    run("multiples of 16 + 8", 16 * ((g % 24) - 12) + 8)
    run("odd", 2 * ((g % 190) - 95) + 1)
ctx.close()
