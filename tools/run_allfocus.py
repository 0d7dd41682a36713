"""Config 5 on the structured scene: one focus map, then a few all-focus renders (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools/run_allfocus.py [method=TEN_WM] [launches=4] [map=estimated|constant] [variant]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
method = sys.argv[1] if len(sys.argv) > 1 else "TEN_WM"
variant = sys.argv[4] if len(sys.argv) > 4 else None
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
which = sys.argv[3] if len(sys.argv) > 3 else "estimated"
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)
if which == "constant":
    m = np.full((H, W, 4), 128, np.uint8); m[..., 3] = 255
    ctx.upload_map(0, m); ctx.upload_map(1, m)
else:
    ctx.focus_map()
if variant:
    ctx.set_variant(method, variant)
for _ in range(launches):
    ctx.render(method, all_focus=True)
ctx.sync()
print(ctx.last_kernel_name())
ctx.close()
