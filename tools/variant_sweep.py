"""On-GPU: parity of every kernel variant on small ragged cases + interleaved timing rounds at a BASELINE config.
usage: python tools/variant_sweep.py [--config 2|3|4] [--rounds R] [--filter substr] [--no-parity] [--std]"""
import argparse, sys, json
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--runs", type=int, default=10)
ap.add_argument("--filter", default="")
ap.add_argument("--no-parity", action="store_true")
ap.add_argument("--std", action="store_true")
args = ap.parse_args()

def parity(cols, rows, W, H, V, traj="0,0,1,1", focus=0.23, aspect=1.783, effect=3.0, seed=0x1F1F, rng=0.0):
    ctx = L.Context(0)
    ctx.set_grid(cols, rows, W, H)
    lf = oc.synthetic_lf(cols*rows, W, H, seed)
    ctx.fill_synthetic(seed); ctx.sync()
    hp = L.build_params(cols, rows, W, H, traj, focus, rng, effect, aspect, V)
    ctx.set_params(hp)
    af = rng > 0
    m = None
    if af:
        m = oc.synthetic_lf(1, W, H, 77)[0]
        ctx.upload_map(0, m); ctx.upload_map(1, m)  # TEN_WM reads map 0, STD map 1 (reference defaults)
    ref_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8, all_focus=af, map_plane=m, focus=focus, rng=rng)
    ref_ex = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_EXACT, threads=8, all_focus=af, map_plane=m, focus=focus, rng=rng)
    ref_m16 = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_M16, threads=8, all_focus=af, map_plane=m, focus=focus, rng=rng)
    bad = []
    for name in ctx.list_variants("STD"):
        if args.filter and args.filter not in name: continue
        ctx.set_variant("STD", name); ctx.render("STD", all_focus=af); ctx.sync()
        d = int((ctx.download_views() != ref_std).sum())
        if d: bad.append(("STD/"+name, d))
    for name in ctx.list_variants("TEN_WM"):
        if args.filter and args.filter not in name: continue
        ctx.set_variant("TEN_WM", name); ctx.render("TEN_WM", all_focus=af); ctx.sync()
        out = ctx.download_views()
        dm = int(np.abs(out.astype(int)-ref_m16.astype(int)).max()); de = float((out != ref_ex).mean())
        if dm > 1 or de > 1e-3: bad.append(("TEN/"+name, dm, de))
    print(f"parity {cols}x{rows} {W}x{H} V={V} af={af}: {'OK' if not bad else bad}", flush=True)
    ctx.close()

if not args.no_parity:
    parity(3, 3, 16, 16, 8)
    parity(8, 8, 64, 48, 64)
    parity(4, 4, 33, 17, 5, traj="0.071,0.071,0.93,0.93", effect=7.0, aspect=2.0223, focus=0.3)
    parity(15, 15, 140, 12, 45)
    parity(8, 8, 300, 20, 130)
    parity(8, 8, 130, 9, 64, rng=0.2)
    parity(15, 15, 48, 20, 8, focus=3.0, effect=7.0)

cfgs = {2: (8, 8, 1920, 1080, 64), 3: (15, 15, 1920, 1080, 45), 4: (8, 8, 3840, 2160, 32), 41: (8, 8, 3840, 2160, 256), 31: (15, 15, 1920, 1080, 1), 21: (8, 8, 1920, 1080, 1), 5: (15, 15, 3840, 2160, 64)}
cols, rows, W, H, V = cfgs[args.config]
ctx = L.Context(0)
ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F); ctx.sync()
hp = L.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
ctx.set_params(hp)
B = 4*W*H*(cols*rows+V)
res = {}
methods = ["TEN_WM"] + (["STD"] if args.std else [])
for rnd in range(args.rounds):
    for method in methods:
        for name in ctx.list_variants(method):
            if args.filter and args.filter not in name: continue
            if name == "valu": continue
            ctx.set_variant(method, name)
            st = ctx.benchmark(method, warmup=2, runs=args.runs)
            res.setdefault(method+"/"+name, []).append((st.median_ms, st.min_ms, st.back_to_back_ms))
print(f"config {args.config}: {cols}x{rows} @{W}x{H} V={V}  B_alg={B/1e9:.4f} GB")
for k, v in res.items():
    med = np.median([x[0] for x in v]); mn = min(x[1] for x in v); b2b = np.median([x[2] for x in v])
    print(f"{k:24s} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  b2b {b2b*1e3:8.1f} us  -> {B/med/1e6:7.0f} GB/s ({B/med/1e6/8000:.3f} of 8 TB/s)  {V/med*1e3:9.0f} views/s", flush=True)
ctx.close()
