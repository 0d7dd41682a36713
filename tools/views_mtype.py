"""The views of the planar layout in ordinary / uncached / fine-grained device memory (LFI_VIEWS_MEMORY): write-only planes that
bypass the caches leave more of the Infinity Cache to the inputs of the next launch (the library's default for its own planar
views is uncached).  usage: python tools/views_mtype.py"""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import lfinterpolator_amd as L
    ctx = L.Context(0)
    ctx.set_grid(8, 8, 1920, 1080); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(8, 8, 1920, 1080, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 64))
    ctx.set_output_layout("planar")
    for _ in range(30): ctx.render("TEN_WM")
    ctx.sync()
    best = sorted(ctx.benchmark("TEN_WM", warmup=4, runs=20).back_to_back_ms for _ in range(3))
    v = ctx.download_view(5)
    print(f"views memory {os.environ.get('LFI_VIEWS_MEMORY','default'):12s}: b2b median {best[1]*1e3:8.1f} us  (checksum {int(v.sum())})", flush=True)
    ctx.close()
else:
    for rnd in range(3):
        for kind in ("default", "uncached", "finegrained"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LFI_VIEWS_MEMORY=kind), check=False)
