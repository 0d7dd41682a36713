"""All-focus render time as a function of the focus map's smoothness (config 2 shape)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
cols = rows = 8; W, H = 1920, 1080
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64)); ctx.sync()
def map_of(values):
    m = np.empty((H, W, 4), np.uint8); m[..., :3] = values[..., None]; m[..., 3] = 255; return m
yy, xx = np.mgrid[0:H, 0:W]
maps = {"constant": np.full((H, W), 128, np.uint8),
        "smooth gradient": ((xx / W * 255)).astype(np.uint8),
        "piecewise (64px blocks)": (((xx // 64) * 37 + (yy // 64) * 91) % 256).astype(np.uint8),
        "random": np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)}
variants = dict(v.split("=") for v in sys.argv[1:])   # e.g. TEN_WM=wave_m2_nt STD=persist_m2_nt
for meth, var in variants.items():
    ctx.set_variant(meth, var)
for name, mv in maps.items():
    ctx.upload_map(0, map_of(mv)); ctx.upload_map(1, map_of(mv))  # TEN_WM reads map 0, STD map 1 (reference defaults)
    out = []
    for method in ("TEN_WM", "STD"):
        st = ctx.benchmark(method, all_focus=True, warmup=2, runs=5)
        out.append(f"{method} {st.median_ms*1e3:.0f} us")
    print(f"{name:26s}: " + "  ".join(out), flush=True)
st = ctx.benchmark("TEN_WM", all_focus=False, warmup=2, runs=5); print(f"fixed focus: TEN_WM {st.median_ms*1e3:.0f} us")
ctx.close()
