import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)
for _ in range(4):
    ctx.focus_map(); ctx.render("TEN_WM", all_focus=True)
ctx.sync(); ctx.close()
