"""Config 5 on the structured scene: all-focus renders from the estimated map and from a constant map, TEN_WM and STD (round 4: STD by
blend_stdxa and by blend_afs — every sample gathered once, variant "filtered_gather_once").  usage: python tools/allfocus_map_kinds.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)
def t(method, n=6):
    for _ in range(3): ctx.render(method, all_focus=True)
    ctx.sync()
    r=[]
    for _ in range(3):
        ctx.timer_start()
        for _ in range(n): ctx.render(method, all_focus=True)
        r.append(ctx.timer_stop()/n)
    return sorted(r)[1]
def line(tag):
    ten = t("TEN_WM"); k_ten = ctx.last_kernel_name()
    std = t("STD"); k_std = ctx.last_kernel_name()
    ctx.set_variant("STD", "filtered_gather_once")
    std_old = t("STD"); k_old = ctx.last_kernel_name()
    ctx.set_variant("STD", "auto")
    print(f"{tag}: TEN {ten:.3f} ms ({k_ten})  STD {std:.3f} ms ({k_std})  STD gathered once {std_old:.3f} ms ({k_old})", flush=True)
ctx.focus_map(); ctx.sync()
line("estimated map")
m = np.full((H, W, 4), 128, np.uint8); m[..., 3] = 255
ctx.upload_map(0, m); ctx.upload_map(1, m)
line("constant map ")
ctx.close()
