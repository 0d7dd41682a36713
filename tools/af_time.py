"""Config 5 on the structured scene: all-focus render times, estimated and constant map (LFI_AB_LIB selects a differently built library).
usage: python tools/af_time.py [methods=STD,TEN_WM]"""
import os, sys
sys.path.insert(0, ".")
import tools._ablib  # noqa
import numpy as np
import lfinterpolator_amd as L
methods = (sys.argv[1] if len(sys.argv) > 1 else "STD,TEN_WM").split(",")
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)
ctx.set_output_layout(os.environ.get("LFI_LAYOUT", "rgba"))   # LFI_LAYOUT=planar: the all-focus renders into the planar view layout
if os.environ.get("LFI_TEN_VARIANT"):
    ctx.set_variant("TEN_WM", os.environ["LFI_TEN_VARIANT"])
if os.environ.get("LFI_STD_VARIANT"):
    ctx.set_variant("STD", os.environ["LFI_STD_VARIANT"])
def t(method, n=6):
    for _ in range(3): ctx.render(method, all_focus=True)
    ctx.sync()
    r=[]
    for _ in range(3):
        ctx.timer_start()
        for _ in range(n): ctx.render(method, all_focus=True)
        r.append(ctx.timer_stop()/n)
    return sorted(r)[1]
ctx.focus_map(); ctx.sync()
est = {m: t(m) for m in methods}
m = np.full((H, W, 4), 128, np.uint8); m[..., 3] = 255
ctx.upload_map(0, m); ctx.upload_map(1, m)
con = {m: t(m) for m in methods}
print("  ".join(f"{m}: estimated {est[m]:.3f} constant {con[m]:.3f} ms" for m in methods), flush=True)
ctx.close()
