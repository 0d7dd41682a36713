"""Soak test (not part of the suite): all-focus renders from random focus maps (noise, constant rows, blocks) on random small shapes,
STD bit-exact and TEN_WM within one LSB of the oracle, both view layouts.   usage: python tools/fuzz_allfocus.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for i in range(n_cases):
    cols, rows = int(rng.integers(2, 16)), int(rng.integers(2, 16))
    W = int(rng.choice([17, 64, 100, 128, 129, 200, 257, 300, 520]))
    H = int(rng.integers(2, 12))
    V = int(rng.choice([1, 5, 33, 64, 70]))
    focus = float(rng.choice([0.0, 0.05, 0.3, -0.2]))
    frange = float(rng.choice([0.1, 0.5, 1.2, -0.4]))
    hp = L.build_params(cols, rows, W, H, str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "0.5,0.5,0.5,0.5"])), focus, frange,
                        float(rng.choice([1.0, 3.0, 7.0])), 1.783, V)
    lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        lv = rng.integers(0, 256, size=(H, W))
    elif kind == 1:
        lv = np.repeat(rng.integers(0, 256, size=(H, 1)), W, axis=1)
    else:
        lv = np.repeat(np.repeat(rng.integers(0, 256, size=((H + 3) // 4, (W + 89) // 90)), 4, axis=0), 90, axis=1)[:H, :W]
    m = np.repeat(lv[..., None].astype(np.uint8), 4, axis=-1); m[..., 3] = 255
    want_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=m, focus=hp.focus, rng=hp.range, threads=8)
    want_ten = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=m, focus=hp.focus, rng=hp.range, threads=8)
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.upload_grid(lf); ctx.set_params(hp)
    ctx.upload_map(0, m); ctx.upload_map(1, m)
    for layout in ("rgba", "planar"):
        ctx.set_output_layout(layout)
        ctx.render("STD", all_focus=True); ctx.sync()
        ok_std = (ctx.download_views() == want_std).all()
        ctx.render("TEN_WM", all_focus=True); ctx.sync()
        d = np.abs(ctx.download_views().astype(int) - want_ten.astype(int)).max()
        if not ok_std or d > 1:
            bad += 1; print("MISMATCH", layout, ok_std, d, cols, rows, W, H, V, focus, frange, kind)
    ctx.close()
    if (i + 1) % 20 == 0:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
