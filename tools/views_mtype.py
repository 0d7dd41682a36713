"""(measurement build of the library: hipcc -DLFI_MEASUREMENT_BUILD, LFI_AB_LIB)  The views of the planar layout in ordinary / uncached device memory (LFI_VIEWS_MEMORY=default|uncached; the library's choice for its
own planar views is uncached): write-only planes that bypass the caches leave more of the Infinity Cache to the inputs of the next
launch — measured per BASELINE config, alternating processes on one box.   usage: python tools/views_mtype.py"""
import os, subprocess, sys
sys.path.insert(0, ".")
CFG = {2: (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), 3: (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       4: (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), "4w": (8, 8, 3840, 2160, 256, "0,0,1,1", 0.23, 1.783, 3.0),
       5: (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
if len(sys.argv) > 1:
    import lfinterpolator_amd as L
    for cfg in (2, 3, 4, "4w", 5):
        cols, rows, W, H, V, traj, focus, aspect, effect = CFG[cfg]
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
        ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
        ctx.set_output_layout("planar")
        for _ in range(10): ctx.render("TEN_WM")
        ctx.sync()
        best = sorted(ctx.benchmark("TEN_WM", warmup=3, runs=15).back_to_back_ms for _ in range(3))
        v = ctx.download_view(V // 2)
        print(f"views memory {os.environ.get('LFI_VIEWS_MEMORY','(library)'):10s} config {cfg}: b2b median {best[1]*1e3:8.1f} us  (checksum {int(v.sum())})", flush=True)
        ctx.close()
else:
    for rnd in range(2):
        for kind in ("uncached", "default"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LFI_VIEWS_MEMORY=kind), check=False)
