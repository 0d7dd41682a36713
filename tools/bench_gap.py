"""Why is bench.py's per-launch time above the sweep's?  A/B: context-owned (hipMalloc) planes vs torch tensors (as is / 2 MiB aligned),
own stream vs torch stream."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import lfinterpolator_amd as L
cols = rows = 8; W, H, V = 1920, 1080, 64
hp = L.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
dev = torch.device("cuda", 0)
def aligned(nbytes, align):
    buf = torch.empty(nbytes + align, dtype=torch.uint8, device=dev)
    off = (-buf.data_ptr()) % align
    return buf, buf[off:off + nbytes]
def run(mode):
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
    keep = []
    n = 64 * H * W * 4
    if mode.startswith("torch"):
        if "aligned" in mode:
            b, g = aligned(n, 2 << 20); keep += [b, g]
        else:
            g = torch.empty(n, dtype=torch.uint8, device=dev); keep.append(g)
        ctx.attach_grid(g.data_ptr(), n)
    ctx.fill_synthetic(0x1F1F); ctx.set_params(hp)
    if mode.startswith("torch"):
        if "aligned" in mode:
            b, v = aligned(n, 2 << 20); keep += [b, v]
        else:
            v = torch.empty(n, dtype=torch.uint8, device=dev); keep.append(v)
        ctx.attach_views(v.data_ptr(), n)
    if "tstream" in mode:
        s = torch.cuda.Stream(device=dev); keep.append(s); ctx.set_stream(s.cuda_stream)
    gp, _ = ctx.grid_device_ptr(); vp, _ = ctx.views_device_ptr()
    for _ in range(300): ctx.render("TEN_WM")
    ctx.sync()
    res = []
    for rep in range(3):
        ctx.timer_start()
        for _ in range(50): ctx.render("TEN_WM")
        res.append(ctx.timer_stop() / 50)
    print(f"{mode:28s} grid%2MiB={gp % (2<<20):8d} views%2MiB={vp % (2<<20):8d}  {min(res)*1e3:.1f} us (runs {[round(r*1e3,1) for r in res]})", flush=True)
    ctx.close()
for rnd in range(2):
    for mode in ("hipmalloc", "torch", "torch_aligned", "torch_tstream", "torch_aligned_tstream"):
        run(mode)
