"""PCIe-inclusive rate of the boundary when the caller hands over HOST buffers (lfi_upload_image / lfi_download_view):
upload 64 planes, one launch, download 64 views at config 2.  Reported in DESIGN.md; never the bench `value`."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
cols = rows = 8; W, H, V = 1920, 1080, 64
lf = np.random.default_rng(0).integers(0, 256, size=(64, H, W, 4), dtype=np.uint8)
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V))
pin_in = ctx.pinned_empty(lf.shape); pin_in[...] = lf
pin_out = ctx.pinned_empty((V, H, W, 4))
for kind, src, dst in (("pageable", lf, None), ("page-locked (lfi_alloc_pinned)", pin_in, pin_out)):
    for rep in range(3):
        t0 = time.perf_counter(); ctx.upload_grid(src); t1 = time.perf_counter()
        ctx.render("TEN_WM"); ctx.sync(); t2 = time.perf_counter()
        out = ctx.download_views(out=dst); t3 = time.perf_counter()
        print(f"{kind}: upload {t1-t0:.4f}s ({lf.nbytes/(t1-t0)/1e9:.1f} GB/s)  render {1e3*(t2-t1):.3f} ms  download {t3-t2:.4f}s "
              f"({out.nbytes/(t3-t2)/1e9:.1f} GB/s)  => {V/(t3-t0):.0f} views/s PCIe-inclusive (synchronous copies)")
