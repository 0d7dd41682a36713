"""Launch a BASELINE config's TEN_WM render a few times (for rocprofv3 --pmc / --kernel-trace runs of blend_p3 / blend_planar).
usage: python tools/run_p3.py CONFIG [layout=planar] [launches=5] [method=TEN_WM]"""
import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
CFG = {2: (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), 3: (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       4: (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), 5: (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
cfg = int(sys.argv[1]); layout = sys.argv[2] if len(sys.argv) > 2 else "planar"; launches = int(sys.argv[3]) if len(sys.argv) > 3 else 5; method = sys.argv[4] if len(sys.argv) > 4 else "TEN_WM"
cols, rows, W, H, V, traj, focus, aspect, effect = CFG[cfg]
ctx = L.Context(0)
ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
ctx.set_output_layout(layout)
ctx.prepare(method)  # the derived planar copy (with tuned phases) is built here, as in bench.py
for _ in range(launches):
    ctx.render(method)
ctx.sync()
print(ctx.last_kernel_name())
ctx.close()
