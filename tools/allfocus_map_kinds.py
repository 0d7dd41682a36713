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
ctx.focus_map(); ctx.sync()
print("estimated map: TEN %.3f ms  STD %.3f ms" % (t("TEN_WM"), t("STD")), flush=True)
m = np.full((H, W, 4), 128, np.uint8); m[..., 3] = 255
ctx.upload_map(0, m); ctx.upload_map(1, m)
print("constant map : TEN %.3f ms  STD %.3f ms" % (t("TEN_WM"), t("STD")), flush=True)
ctx.close()
