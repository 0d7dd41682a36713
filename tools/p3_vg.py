"""(measurement build of the library: hipcc -DLFI_MEASUREMENT_BUILD, LFI_AB_LIB)  blend_p3 with four waves x 16 views per workgroup (LFI_P3_VG=1) against two waves x 32 views (LFI_P3_VG=2) at configs 2, 3, 4-rank,
4-whole (256 views: four view passes per tile) and 5 — one process per setting, alternating, same box.  (Unset, the library picks 16 views
per wave for one chunk of images and 32 for several.)   usage: python tools/p3_vg.py"""
import os, subprocess, sys
sys.path.insert(0, ".")
CFG = {2: (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), 3: (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       4: (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), "4w": (8, 8, 3840, 2160, 256, "0,0,1,1", 0.23, 1.783, 3.0), 5: (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
if len(sys.argv) > 1:
    import numpy as np
    import lfinterpolator_amd as L
    sys.path.insert(0, "tools")
    import _ablib  # LFI_AB_LIB
    for cfg in (2, 3, 4, "4w", 5):
        cols, rows, W, H, V, traj, focus, aspect, effect = CFG[cfg]
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
        ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
        ctx.set_output_layout("planar")
        for _ in range(10): ctx.render("TEN_WM")
        ctx.sync()
        best = []
        for rnd in range(3):
            st = ctx.benchmark("TEN_WM", warmup=3, runs=15)
            best.append(st.back_to_back_ms)
        v = ctx.download_view(V // 2)
        print(f"vg={os.environ.get('LFI_P3_VG','1')} config {cfg}: b2b median {sorted(best)[1]*1e3:8.1f} us   checksum view {V // 2}: {int(v.astype(np.uint64).sum())}", flush=True)
        ctx.close()
else:
    for rnd in range(2):
        for vg in ("1", "2"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LFI_P3_VG=vg), check=False)
