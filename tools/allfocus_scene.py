"""All-focus renders on the structured scene (lfi_fill_synthetic_scene): what does the estimated map look like, and how do the
renders compare with renders from a constant / blocky / the estimated map?   usage: python tools/allfocus_scene.py [cols W H]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 15
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
ctx = L.Context(0)
ctx.set_grid(cols, cols, W, H)
hp = L.build_params(cols, cols, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64)
ctx.set_params(hp)
ctx.fill_synthetic_scene(0x1F1F)
ctx.focus_map(); ctx.sync()
m0, m1 = ctx.download_map(0)[..., 0], ctx.download_map(1)[..., 0]
for name, m in (("map0", m0), ("map1", m1)):
    vals, counts = np.unique(m, return_counts=True)
    top = sorted(zip(counts, vals), reverse=True)[:6]
    rows128 = m[:, : (W // 128) * 128].reshape(H, W // 128, 128)
    uniform = (rows128.max(-1) == rows128.min(-1)).mean()
    dx = (np.diff(m.astype(int), axis=1) != 0).mean()
    print(f"{name}: {len(vals)} distinct values, top {[(int(v), round(c / m.size, 3)) for c, v in top]}, uniform 128-px tiles {uniform:.3f}, neighbour changes {dx:.4f}", flush=True)
def t(method, all_focus=True):
    st = ctx.benchmark(method, all_focus=all_focus, warmup=2, runs=5)
    return st.median_ms
print(f"fixed focus: TEN_WM {t('TEN_WM', False):.3f} ms  STD {t('STD', False):.3f} ms", flush=True)
print(f"estimated maps: TEN_WM {t('TEN_WM'):.3f} ms  STD {t('STD'):.3f} ms", flush=True)
def put(m):
    rgba = np.repeat(m[..., None], 4, axis=-1).astype(np.uint8); rgba[..., 3] = 255
    ctx.upload_map(0, rgba); ctx.upload_map(1, rgba)
put(np.full((H, W), 100, np.uint8)); print(f"constant map: TEN_WM {t('TEN_WM'):.3f} ms  STD {t('STD'):.3f} ms", flush=True)
yy, xx = np.mgrid[0:H, 0:W]
put((((xx >> 8) * 7 + (yy >> 8) * 13) % 4 * 64 + 20).astype(np.uint8)); print(f"256-px blocks: TEN_WM {t('TEN_WM'):.3f} ms  STD {t('STD'):.3f} ms", flush=True)
put((xx * 255 // W).astype(np.uint8)); print(f"horizontal gradient: TEN_WM {t('TEN_WM'):.3f} ms  STD {t('STD'):.3f} ms", flush=True)
put(m1); print(f"estimated map 1 for both: TEN_WM {t('TEN_WM'):.3f} ms  STD {t('STD'):.3f} ms", flush=True)
ctx.close()
