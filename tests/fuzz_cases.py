"""Randomised parity cases shared by tests/test_gpu_fuzz.py (fixed seeds, a bounded number of cases) and the soak tools under
tools/ (fuzz_parity.py, fuzz_focus.py, fuzz_allfocus.py: any number of cases, any seed).

Every function renders random small shapes through the HIP library (the C-ABI) and checks them against the CPU oracle — STD and
focus maps bit-exact, TEN_WM within one LSB of the fp16-accumulate model M16 — and returns the list of mismatches (empty = pass).
`L` is the lfinterpolator_amd package, `oc` the oracle binding (oracle/lfi_oracle_c.py): the checker, never the thing measured.
"""
import numpy as np

TEN_TOL_LSB = 1


def fuzz_blend(L, oc, n_cases: int, seed: int, log=None) -> list:
    """Fixed-focus renders: random grids (1–225 images), widths around the 128-pixel tile, 1–129 views, every main variant, view
    ranges, and both view layouts."""
    rng = np.random.default_rng(seed)
    ten_variants = ["auto", "persist_m2_nt", "wave_m2_nt"]
    std_variants = ["auto", "persist_m2_nt"]
    bad = []
    for i in range(n_cases):
        cols, rows = int(rng.integers(1, 16)), int(rng.integers(1, 16))
        if cols * rows < 2 or cols * rows > 225:
            cols, rows = 3, 4
        W = int(rng.choice([1, 4, 31, 33, 64, 100, 127, 128, 129, 191, 256, 300, 513, 700]))
        H = int(rng.integers(1, 10))
        V = int(rng.choice([1, 3, 31, 32, 33, 64, 65, 100, 129, 200, 300]))
        focus = float(rng.choice([0.0, 0.03, 0.23, 0.5, 1.1, -0.4]))
        traj = str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "1,0,0,1", "0.5,0.5,0.5,0.5", "0.2,0.9,0.8,0.1"]))
        effect = float(rng.choice([1.0, 3.0, 7.0]))
        hp = L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, 1.783, V)
        lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
        case = dict(kind="blend", cols=cols, rows=rows, W=W, H=H, V=V, focus=focus, traj=traj, effect=effect)
        want_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
        want_ten = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oc.TEN_M16, threads=8)
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        v0 = int(rng.integers(0, V))
        v1 = int(rng.integers(v0 + 1, V + 1))
        for var in std_variants:
            ctx.set_variant("STD", var)
            ctx.render("STD")
            ctx.sync()
            if not (ctx.download_views() == want_std).all():
                bad.append(dict(case, what="STD", variant=var))
        for var in ten_variants:
            ctx.set_variant("TEN_WM", var)
            ctx.render("TEN_WM")
            ctx.sync()
            full = ctx.download_views()
            d = int(np.abs(full.astype(int) - want_ten.astype(int)).max())
            ctx.render("TEN_WM", v0=v0, v1=v1)
            ctx.sync()
            if d > TEN_TOL_LSB or not (ctx.download_views() == full).all():
                bad.append(dict(case, what="TEN_WM", variant=var, lsb=d, v0=v0, v1=v1))
        # the planar view layout: TEN_WM within one LSB of the oracle and byte-identical to the RGBA layout's default kernel, view ranges
        # included; STD bit-exact (round 4: blend_stdx writes the byte planes itself, one to four chunks of images), a view range over it too
        ctx.set_variant("TEN_WM", "auto")
        ctx.set_variant("STD", "auto")
        ctx.render("TEN_WM")
        ctx.sync()
        want_rgba = ctx.download_views()
        ctx.set_output_layout("planar")
        ctx.render("TEN_WM")
        ctx.sync()
        got = ctx.download_views()
        kernel = ctx.last_kernel_name()
        ctx.render("TEN_WM", v0=v0, v1=v1)
        ctx.sync()
        part = ctx.download_views()
        ctx.render("STD")
        ctx.render("STD", v0=v0, v1=v1)
        ctx.sync()
        d = int(np.abs(got.astype(int) - want_ten.astype(int)).max())
        if d > TEN_TOL_LSB or not (got == want_rgba).all() or not (part == got).all() or not (ctx.download_views() == want_std).all():
            bad.append(dict(case, what="planar layout", kernel=kernel, lsb=d, v0=v0, v1=v1))
        # round 4: the quilt of a random tiling (assembled on the device, filled in two parts) is the montage of the views; and after
        # lfi_release_inputs — the planar copy is the only copy of the inputs — STD stays the oracle's and TEN_WM the same bytes
        ctx.render("TEN_WM")
        ctx.sync()
        tx = int(rng.integers(1, 5))
        ty = int(rng.integers(1, max(2, min(4, V // tx) + 1)))
        if tx * ty <= V:
            quilt = np.zeros((ty * H, tx * W, 4), np.uint8)
            cut = int(rng.integers(0, tx * ty + 1))
            if cut > 0:
                ctx.download_quilt_tiles(quilt, tx, ty, 0, cut)
            if cut < tx * ty:
                ctx.download_quilt_tiles(quilt, tx, ty, cut, tx * ty - cut, v0=cut)
            montage = got[:tx * ty].reshape(ty, tx, H, W, 4).transpose(0, 2, 1, 3, 4).reshape(ty * H, tx * W, 4)
            if not (quilt == montage).all():
                bad.append(dict(case, what="quilt", tiles=(tx, ty), cut=cut))
        try:
            ctx.release_inputs()
            ctx.render("TEN_WM")
            ctx.sync()
            same = bool((ctx.download_views() == got).all())
            ctx.render("STD")
            ctx.sync()
            if not same or not (ctx.download_views() == want_std).all():
                bad.append(dict(case, what="released inputs", ten_same=same))
        except L.LfiError as e:
            if "cannot be built" not in str(e):          # (absurd offsets: the planar copy is refused, and so is the release)
                bad.append(dict(case, what="released inputs", error=str(e)))
        ctx.close()
        if log and (i + 1) % 20 == 0:
            log(f"{i + 1} cases, {len(bad)} mismatches")
    return bad


def fuzz_focus(L, oc, n_cases: int, seed: int, log=None) -> list:
    """Focus maps: the factored estimate, the LDS-staged one and the reference-shaped plain kernel on random shapes, radii, focus
    ranges and contents (quantised and partly black inputs produce ties and FLT_MIN taps) — all against the oracle."""
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(n_cases):
        cols, rows = int(rng.integers(2, 10)), int(rng.integers(2, 10))
        W = int(rng.choice([8, 33, 64, 100, 130, 257, 300, 640]))
        H = int(rng.integers(2, 48))
        focus = float(rng.choice([0.0, 0.05, 0.22, -0.2, 0.6]))
        frange = float(rng.choice([0.01, 0.1, 0.17, 0.5, 1.0]))
        traj = str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "0.5,0.5,0.5,0.5"]))
        hp = L.build_params(cols, rows, W, H, traj, focus, frange, 3.0, 1.783, 4)
        if rng.random() < 0.5:
            hp.block_radius = np.array([int(rng.integers(1, 12)), int(rng.integers(1, 6))], np.int32)
        lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
        q = int(rng.choice([1, 32, 64, 255]))
        lf = (lf // q * q).astype(np.uint8)
        if rng.random() < 0.3:
            lf[:, : H // 2, : W // 3, :3] = 0
        lf[..., 3] = 255
        case = dict(kind="focus", cols=cols, rows=rows, W=W, H=H, focus=focus, range=frange, traj=traj, radius=[int(x) for x in hp.block_radius], q=q)
        want0 = oc.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
        want1 = oc.focus_filter(want0, hp.block_radius)
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        for var in ("factored", "factored_direct", "plain", "lds"):
            ctx.set_variant("FOCUS", var)
            ctx.focus_map()
            ctx.sync()
            m0, m1 = ctx.download_map(0), ctx.download_map(1)
            if not ((m0 == want0).all() and (m1 == want1).all()):
                bad.append(dict(case, what="focus map", variant=var, wrong=int((m0 != want0).any(-1).sum())))
        ctx.close()
        if log and (i + 1) % 20 == 0:
            log(f"{i + 1} cases, {len(bad)} mismatches")
    return bad


def fuzz_allfocus(L, oc, n_cases: int, seed: int, log=None) -> list:
    """All-focus renders from random focus maps (noise, constant rows, blocks) on random small shapes, both view layouts."""
    rng = np.random.default_rng(seed)
    bad = []
    for i in range(n_cases):
        cols, rows = int(rng.integers(2, 16)), int(rng.integers(2, 16))
        W = int(rng.choice([17, 64, 100, 128, 129, 200, 257, 300, 520]))
        H = int(rng.integers(2, 12))
        V = int(rng.choice([1, 5, 33, 64, 70]))
        focus = float(rng.choice([0.0, 0.05, 0.3, -0.2]))
        frange = float(rng.choice([0.1, 0.5, 1.2, -0.4]))
        traj = str(rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "0.5,0.5,0.5,0.5"]))
        effect = float(rng.choice([1.0, 3.0, 7.0]))
        hp = L.build_params(cols, rows, W, H, traj, focus, frange, effect, 1.783, V)
        lf = oc.synthetic_lf(cols * rows, W, H, int(rng.integers(1, 1 << 30)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            lv = rng.integers(0, 256, size=(H, W))
        elif kind == 1:
            lv = np.repeat(rng.integers(0, 256, size=(H, 1)), W, axis=1)
        else:
            lv = np.repeat(np.repeat(rng.integers(0, 256, size=((H + 3) // 4, (W + 89) // 90)), 4, axis=0), 90, axis=1)[:H, :W]
        m = np.repeat(lv[..., None].astype(np.uint8), 4, axis=-1)
        m[..., 3] = 255
        case = dict(kind="allfocus", cols=cols, rows=rows, W=W, H=H, V=V, focus=focus, range=frange, traj=traj, effect=effect, map=kind)
        want_std = oc.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=m, focus=hp.focus, rng=hp.range, threads=8)
        want_ten = oc.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=m, focus=hp.focus, rng=hp.range, threads=8)
        ctx = L.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        ctx.upload_map(0, m)
        ctx.upload_map(1, m)
        for layout in ("rgba", "planar"):
            ctx.set_output_layout(layout)
            ctx.render("STD", all_focus=True)
            ctx.sync()
            ok_std = bool((ctx.download_views() == want_std).all())
            ctx.render("TEN_WM", all_focus=True)
            ctx.sync()
            d = int(np.abs(ctx.download_views().astype(int) - want_ten.astype(int)).max())
            if not ok_std or d > TEN_TOL_LSB:
                bad.append(dict(case, what="all-focus", layout=layout, std_exact=ok_std, lsb=d, kernel=ctx.last_kernel_name()))
            if cols * rows > 128 and layout == "rgba":
                # three or four chunks of images: blend_afs (every sample gathered once; round 4) must give the same bytes
                ctx.set_variant("STD", "filtered_gather_once")
                ctx.render("STD", all_focus=True)
                ctx.sync()
                if ctx.last_kernel_name() != "blend_afs<STD,allfocus>" or not (ctx.download_views() == want_std).all():
                    bad.append(dict(case, what="all-focus", layout=layout, std_exact=False, kernel=ctx.last_kernel_name()))
                ctx.set_variant("STD", "auto")
        ctx.close()
        if log and (i + 1) % 20 == 0:
            log(f"{i + 1} cases, {len(bad)} mismatches")
    return bad
