"""Launches of a fixed-focus -f sweep for rocprofv3 passes: MODE tuned (lfi_prepare, one parameter set), sweep (16 parameter sets around the
configuration's focus, lfi_set_params + lfi_render each, no lfi_prepare) or same (16 × the SAME parameter set through lfi_set_params: what
lfi_set_params itself costs).  usage: python tools/run_sweep.py CONFIG MODE [layout=planar] [method=TEN_WM] [passes=2]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
import bench
ci, mode = int(sys.argv[1]), sys.argv[2]
layout = sys.argv[3] if len(sys.argv) > 3 else "planar"
method = sys.argv[4] if len(sys.argv) > 4 else "TEN_WM"
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cfg = bench.CONFIGS[ci]
mk = lambda f: L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], f, 0.0, cfg["effect"], cfg["aspect"], cfg["views"])
ctx = L.Context(0)
ctx.set_grid(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"]); ctx.fill_synthetic(bench.SEED)
f0 = cfg["focus"]
ctx.set_params(mk(f0)); ctx.set_output_layout(layout)
ctx.prepare(method)
sets = [mk(f) for f in (np.linspace(f0 - 0.02, f0 + 0.02, 16) if mode == "sweep" else [f0] * 16)]
for _ in range(passes):
    for hp in sets:
        if mode != "tuned":
            ctx.set_params(hp)
        ctx.render(method)
ctx.sync()
print(ctx.last_kernel_name())
ctx.close()
