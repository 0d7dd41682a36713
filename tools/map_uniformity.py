"""How coherent is the focus map the all-focus render reads?  Config 5 on the structured scene: fraction of 128-pixel row tiles (the render
kernels' tile) whose map value is one constant, histogram of distinct values per tile, run lengths.  usage: python tools/map_uniformity.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
ctx.set_params(L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V))
ctx.fill_synthetic_scene(0x1F1F)
ctx.focus_map()
ctx.sync()
for idx in (0, 1):
    m = ctx.download_map(idx)[..., 0]
    t = m.reshape(H, W // 128, 128)
    distinct = np.array([[len(np.unique(t[y, x])) for x in range(W // 128)] for y in range(0, H, 8)])
    uni = (t.max(axis=2) == t.min(axis=2))
    runs = np.diff(np.flatnonzero(np.concatenate(([True], m[::8].ravel()[1:] != m[::8].ravel()[:-1], [True]))))
    print(f"map {idx}: values {len(np.unique(m))} distinct; uniform 128-px tiles {uni.mean()*100:.1f} %; distinct values per tile: median {np.median(distinct):.0f}, "
          f"mean {distinct.mean():.1f}; run length along x: median {np.median(runs):.0f}, mean {runs.mean():.1f} px", flush=True)
    t32 = m.reshape(H, W // 32, 32)
    print(f"        uniform 32-px runs {(t32.max(axis=2) == t32.min(axis=2)).mean()*100:.1f} %, 64-px {(m.reshape(H, W//64, 64).max(axis=2) == m.reshape(H, W//64, 64).min(axis=2)).mean()*100:.1f} %")
ctx.close()
