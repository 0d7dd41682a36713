"""Host → device upload of a whole light field and the wall time until the first launch has finished: lfi_upload_image (synchronous,
round 1) against lfi_upload_image_async (round 2: page-locked staging ring on a copy stream, joined by an event), from pageable
and from page-locked host arrays.  BASELINE config 3 (15×15 @1080p, 1.87 GB) by default.
usage: python tools/upload_time.py [cols W H]"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc   # only to generate host images
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 15
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
n = cols * cols
oc.build()
lf = np.empty((n, H, W, 4), np.uint8)
with ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda g: lf.__setitem__(g, oc.synthetic_plane(g, W, H, 0x1F1F)), range(n)))
hp = L.build_params(cols, cols, W, H, "0,0.5,1,0.5", 0.06, 0.0, 3.0, 2.276, 45)
ctx = L.Context(0)
ctx.set_grid(cols, cols, W, H)
ctx.set_params(hp)
pinned = ctx.pinned_empty((n, H, W, 4))
pinned[...] = lf
ref = None
for name, src, asynchronous in (("sync, pageable", lf, False), ("async, pageable (staged)", lf, True), ("sync, page-locked", pinned, False),
                                ("async, page-locked (in place)", pinned, True)) * 2:
    ctx.sync()
    t0 = time.perf_counter()
    for g in range(n):
        (ctx.upload_image_async if asynchronous else ctx.upload_image)(g, src[g])
    t1 = time.perf_counter()
    ctx.render("TEN_WM")          # builds the planar copy, then the first launch
    ctx.sync()
    t2 = time.perf_counter()
    v = ctx.download_view(7)
    if ref is None:
        ref = v
    assert (v == ref).all()
    print(f"{name:32s}: upload calls return after {1e3*(t1-t0):7.1f} ms ({lf.nbytes/1e9/(t1-t0):5.1f} GB/s of host time), "
          f"first launch done at {1e3*(t2-t0):7.1f} ms", flush=True)
ctx.close()
