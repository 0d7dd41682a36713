import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
cols = rows = 8; W, H = 1920, 1080
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64))
if len(sys.argv) > 1:
    ctx.set_variant("FOCUS", sys.argv[1])
for _ in range(5):
    ctx.focus_map(); ctx.sync()
ctx.close()
