"""Does alternating the sweep direction between consecutive launches pay (Infinity Cache reuse of the inputs)?
LFI_FLAG_SINGLE_SWEEP_DIRECTION (alternate=0) against the default (alternate=1), one process per setting, three rounds.
usage: python tools/p3_alternate.py"""
import os, subprocess, sys
sys.path.insert(0, ".")
CFG = {2: (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), 3: (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       4: (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), 5: (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
if len(sys.argv) > 1:
    import lfinterpolator_amd as L
    for cfg in (2, 3, 4, 5):
        cols, rows, W, H, V, traj, focus, aspect, effect = CFG[cfg]
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
        ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V),
                       flags=0 if os.environ.get("LFI_P3_ALTERNATE", "1") == "1" else L.LFI_FLAG_SINGLE_SWEEP_DIRECTION)
        ctx.set_output_layout("planar")
        for _ in range(30): ctx.render("TEN_WM")
        ctx.sync()
        best = sorted(ctx.benchmark("TEN_WM", warmup=4, runs=20).back_to_back_ms for _ in range(3))
        print(f"alternate={os.environ.get('LFI_P3_ALTERNATE','1')} config {cfg}: b2b median {best[1]*1e3:8.1f} us", flush=True)
        ctx.close()
else:
    for rnd in range(3):
        for alt in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LFI_P3_ALTERNATE=alt), check=False)
