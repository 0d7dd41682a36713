"""Where does blend_p3's time go?  Runs BASELINE configs 2, 3, 5 (planar view layout) with LFI_P3_ABLATE = 0 (the kernel), 1 (no
k-loop), 2 (no DMA), 3 (no stores) — one process per setting (the switch is read once), same box.
usage: python tools/p3_ablate.py            (spawns itself per setting)"""
import os, subprocess, sys
sys.path.insert(0, ".")
CFG = {2: (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), 3: (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       4: (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), "4w": (8, 8, 3840, 2160, 256, "0,0,1,1", 0.23, 1.783, 3.0), 5: (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
if len(sys.argv) > 1:
    sys.path.insert(0, "tools")
    import _ablib  # noqa: F401  (LFI_AB_LIB: the measurement build, hipcc -DLFI_MEASUREMENT_BUILD)
    import lfinterpolator_amd as L
    for cfg in (2, 3, 4, "4w", 5):
        cols, rows, W, H, V, traj, focus, aspect, effect = CFG[cfg]
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
        ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
        ctx.set_output_layout("planar")
        for _ in range(30): ctx.render("TEN_WM")
        ctx.sync()
        best = []
        for rnd in range(3):
            st = ctx.benchmark("TEN_WM", warmup=3, runs=15)
            best.append(st.back_to_back_ms)
        print(f"ablate={os.environ.get('LFI_P3_ABLATE','0')} config {cfg}: {ctx.last_kernel_name():22s} b2b median {sorted(best)[1]*1e3:8.1f} us", flush=True)
        ctx.close()
else:
    for rnd in range(2):
        for ab in ("0", "1", "2", "3"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LFI_P3_ABLATE=ab), check=False)
