"""Soak test (not part of the suite): many random small shapes through the default kernels and the main variants, against the
oracle — STD bit-exact, TEN_WM within one LSB of the fp16-accumulate model.  usage: python tools/fuzz_parity.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TEN = ["auto", "persist_m2_nt", "wave_m2_nt"]
STD = ["auto", "persist_m2_nt"]
bad = 0
for i in range(n_cases):
    cols, rows = int(rng.integers(1, 16)), int(rng.integers(1, 16))
    if cols * rows < 2 or cols * rows > 225:
        cols, rows = 3, 4
    W = int(rng.choice([1, 4, 31, 33, 64, 100, 127, 128, 129, 191, 256, 300, 513, 700]))
    H = int(rng.integers(1, 10))
    V = int(rng.choice([1, 3, 31, 32, 33, 64, 65, 100, 129]))
    focus = float(rng.choice([0.0, 0.03, 0.23, 0.5, 1.1, -0.4]))
    traj = str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "1,0,0,1", "0.5,0.5,0.5,0.5", "0.2,0.9,0.8,0.1"]))
    hp = L.build_params(cols, rows, W, H, traj, focus, 0.0, float(rng.choice([1.0, 3.0, 7.0])), 1.783, V)
    lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
    want_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    want_ten = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_M16)
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.upload_grid(lf); ctx.set_params(hp)
    v0 = int(rng.integers(0, V)); v1 = int(rng.integers(v0 + 1, V + 1))
    for var in STD:
        ctx.set_variant("STD", var); ctx.render("STD"); ctx.sync()
        if not (ctx.download_views() == want_std).all():
            bad += 1; print("STD MISMATCH", var, cols, rows, W, H, V, focus, traj)
    for var in TEN:
        ctx.set_variant("TEN_WM", var); ctx.render("TEN_WM"); ctx.sync()
        full = ctx.download_views()
        d = np.abs(full.astype(int) - want_ten.astype(int)).max()
        ctx.render("TEN_WM", v0=v0, v1=v1); ctx.sync()
        part = ctx.download_views()
        if d > 1 or not (part == full).all():
            bad += 1; print("TEN MISMATCH", var, d, cols, rows, W, H, V, focus, traj, v0, v1)
    # the planar view layout (blend_p3): byte-identical to the RGBA layout's default kernel, view ranges included
    ctx.set_variant("TEN_WM", "auto"); ctx.set_variant("STD", "auto")
    ctx.render("TEN_WM"); ctx.sync()
    want_rgba = ctx.download_views()
    ctx.set_output_layout("planar")
    ctx.render("TEN_WM"); ctx.sync()
    got = ctx.download_views()
    ctx.render("TEN_WM", v0=v0, v1=v1); ctx.sync()
    part = ctx.download_views()
    ctx.render("STD"); ctx.sync()
    if not (got == want_rgba).all() or not (part == got).all() or not (ctx.download_views() == want_std).all():
        bad += 1; print("PLANAR LAYOUT MISMATCH", ctx.last_kernel_name(), cols, rows, W, H, V, focus, traj, v0, v1)
    ctx.close()
    if (i + 1) % 20 == 0:
        print(f"{i + 1} cases, {bad} mismatches", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
