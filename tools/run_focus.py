"""Launch the focus-map estimate + filter a few times (for rocprofv3 runs).  usage: python tools/run_focus.py [variant] [cols W H] [scene]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import _ablib  # noqa: F401  (LFI_AB_LIB)
import lfinterpolator_amd as L
variant = sys.argv[1] if len(sys.argv) > 1 else "auto"
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
ctx = L.Context(0); ctx.set_grid(cols, cols, W, H)
ctx.set_params(L.build_params(cols, cols, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64))
if len(sys.argv) > 5 and sys.argv[5] == "scene":
    ctx.fill_synthetic_scene(0x1F1F)
else:
    ctx.fill_synthetic(0x1F1F)
ctx.set_variant("FOCUS", variant)
for _ in range(5):
    ctx.focus_map(); ctx.sync()
ctx.close()
