"""A fixed-focus -f sweep (scripts/focusMapCompare.sh varies -f per run; loadGPUOffsets, src/interpolator.cu:226-246): every render has NEW integer
offsets (lfi_set_params + lfi_render, no lfi_prepare), so the derived planar copy of the inputs keeps the per-image phases it was built with.
Prints ms per step against the tuned steady launch of the same shape.  usage: python tools/fixed_focus_sweep.py [configs e.g. 2,5]"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import numpy as np
import lfinterpolator_amd as L
import bench

configs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,5").split(",")]
for ci in configs:
    cfg = bench.CONFIGS[ci]
    for layout in ("planar", "rgba"):
        for method in ("TEN_WM", "STD"):
            ctx = L.Context(0)
            ctx.set_grid(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"])
            ctx.fill_synthetic(bench.SEED)
            f0 = cfg["focus"]
            hp = L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], f0, 0.0, cfg["effect"], cfg["aspect"], cfg["views"])
            ctx.set_params(hp)
            ctx.set_output_layout(layout)
            iters = 20 if ci == 2 else 6
            tuned = bench.timed(ctx, lambda: ctx.render(method), iters, prepare=(method,))
            k = ctx.last_kernel_name()
            sweep = [L.build_params(cfg["cols"], cfg["rows"], cfg["W"], cfg["H"], cfg["traj"], f, 0.0, cfg["effect"], cfg["aspect"], cfg["views"])
                     for f in np.linspace(f0 - 0.02, f0 + 0.02, 16)]

            def sweep_pass():
                for hp_f in sweep:
                    ctx.set_params(hp_f)
                    ctx.render(method)

            def same_pass():      # lfi_set_params with the SAME parameters before every render: what the call itself costs
                for _ in sweep:
                    ctx.set_params(hp)
                    ctx.render(method)

            ctx.prepare(method)
            same = bench.timed(ctx, same_pass, 2, warm=1, rounds=3) / len(sweep)
            sweep_pass()
            ms = bench.timed(ctx, sweep_pass, 2, warm=1, rounds=3) / len(sweep)
            print(f"config {ci} {layout:6s} {method:6s} {k:28s} tuned {tuned:.4f} ms   set_params + render, same parameters {same:.4f} ms   sweep step {ms:.4f} ms   "
                  f"ratio sweep / tuned {ms / tuned:.3f}, sweep / same {ms / same:.3f}", flush=True)
            ctx.close()
