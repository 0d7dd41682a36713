"""Quick on-GPU sanity run: small-case parity against the oracle and a variant timing sweep at config 2."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

def parity(cols, rows, W, H, V, traj="0,0,1,1", focus=0.23, aspect=1.783, effect=3.0, seed=0x1F1F):
    ctx = L.Context(0)
    ctx.set_grid(cols, rows, W, H)
    lf = oc.synthetic_lf(cols*rows, W, H, seed)
    ctx.fill_synthetic(seed); ctx.sync()
    hp = L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V)
    ctx.set_params(hp)
    ref_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    ref_ex = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_EXACT, threads=8)
    ref_m16 = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_M16, threads=8)
    res = {}
    for name in ctx.list_variants("STD"):
        ctx.set_variant("STD", name); ctx.render("STD"); ctx.sync()
        out = ctx.download_views()
        res["STD/"+name] = int((out != ref_std).sum())
    for name in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", name); ctx.render("TEN_WM"); ctx.sync()
        out = ctx.download_views()
        res["TEN/"+name] = (int((out != ref_ex).sum()), int(np.abs(out.astype(int)-ref_m16.astype(int)).max()))
    print(f"parity {cols}x{rows} {W}x{H} V={V}:", res, flush=True)
    ctx.close()

# MFMA probe with subnormals
ctx = L.Context(0)
rng = np.random.default_rng(1)
a = rng.integers(0, 0x3c00, size=(32,16)).astype(np.uint16)   # weights in [0,1) incl. subnormals
a[0, :4] = [1, 2, 0x3ff, 0x400]
b = rng.integers(0, 256, size=(16,32)).astype(np.uint16)      # pixel bytes as subnormal bit patterns
c = ctx.debug_mfma_f16(a, b)
ref = (a.view(np.float16).astype(np.float64) @ b.view(np.float16).astype(np.float64))
print("mfma probe max rel err", np.abs(c - ref).max() / np.abs(ref).max(), "exact frac", (c.astype(np.float64) == ref.astype(np.float32).astype(np.float64)).mean(), flush=True)
ctx.close()

parity(3, 3, 16, 16, 8)
parity(8, 8, 64, 48, 64)
parity(4, 4, 33, 17, 5, traj="0.071,0.071,0.93,0.93", effect=7.0, aspect=2.0223, focus=0.3)
parity(15, 15, 40, 32, 45)
parity(8, 8, 256, 64, 64)

# timing sweep at config 2
ctx = L.Context(0)
ctx.set_grid(8, 8, 1920, 1080); ctx.fill_synthetic(0x1F1F); ctx.sync()
hp = L.build_params(8, 8, 1920, 1080, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 64)
ctx.set_params(hp)
B = 4*1920*1080*(64+64)
for rnd in range(2):
    for method in ("TEN_WM", "STD"):
        for name in ctx.list_variants(method):
            if name == "valu" and rnd: continue
            ctx.set_variant(method, name)
            st = ctx.benchmark(method, warmup=2, runs=10 if name != "valu" else 3)
            print(f"{method}/{name}: median {st.median_ms:.4f} ms min {st.min_ms:.4f} b2b {st.back_to_back_ms:.4f}  -> {B/st.median_ms/1e9:.1f} GB/s ({B/st.median_ms/1e9/8000:.3f} of 8TB/s)", flush=True)
ctx.close()
