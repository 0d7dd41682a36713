"""Config 2 in the reference's RGBA view layout: STD (band method on the fp16 matrix pipe, bit-exact) against TEN_WM — the same pipeline
without the band test.  The ratio is what the exactness costs; it does not depend on the box.   usage: python tools/std_vs_ten.py"""
import os
import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
sys.path.insert(0, "tools")
import _ablib  # LFI_AB_LIB: A/B runs of differently built libraries (measurement only)
cols = rows = 8; W, H, V = 1920, 1080, 64
ctx = L.Context(0)
ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V))
for _ in range(10): ctx.render("TEN_WM"); ctx.render("STD")
ctx.sync()
for rnd in range(3):
    t = {}
    for m in ("TEN_WM", "STD"):
        t[m] = sorted(ctx.benchmark(m, warmup=3, runs=15).back_to_back_ms for _ in range(3))[1]
        name = ctx.last_kernel_name()
        print(f"{m:7s} {name:24s} {t[m]*1e3:7.1f} us", end="   ")
    print(f"STD / TEN_WM = {t['STD']/t['TEN_WM']:.3f}", flush=True)
ctx.close()
