"""Soak test (not part of the suite): the factored focus-map estimate against the reference-shaped plain kernel on many random
shapes, radii, focus ranges and contents (both are checked against the oracle at small sizes by the suite).
usage: python tools/fuzz_focus.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n_cases):
    cols, rows = int(rng.integers(2, 10)), int(rng.integers(2, 10))
    W = int(rng.choice([8, 33, 64, 100, 130, 257, 300, 640]))
    H = int(rng.integers(2, 48))
    focus = float(rng.choice([0.0, 0.05, 0.22, -0.2, 0.6]))
    frange = float(rng.choice([0.01, 0.1, 0.17, 0.5, 1.0]))
    hp = L.build_params(cols, rows, W, H, str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "0.5,0.5,0.5,0.5"])), focus, frange, 3.0, 1.783, 4)
    if rng.random() < 0.5:
        hp.block_radius = np.array([int(rng.integers(1, 12)), int(rng.integers(1, 6))], np.int32)
    lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
    q = int(rng.choice([1, 32, 64, 255]))
    lf = (lf // q * q).astype(np.uint8)
    if rng.random() < 0.3:
        lf[:, : H // 2, : W // 3, :3] = 0
    lf[..., 3] = 255
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.upload_grid(lf); ctx.set_params(hp)
    maps = {}
    for var in ("factored", "plain", "lds"):
        ctx.set_variant("FOCUS", var); ctx.focus_map(); ctx.sync()
        maps[var] = (ctx.download_map(0), ctx.download_map(1))
    for var in ("factored", "lds"):
        if not ((maps[var][0] == maps["plain"][0]).all() and (maps[var][1] == maps["plain"][1]).all()):
            bad += 1
            print("MISMATCH", var, cols, rows, W, H, focus, frange, list(hp.block_radius), int((maps[var][0] != maps["plain"][0]).sum()))
    ctx.close()
    if (i + 1) % 50 == 0:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
