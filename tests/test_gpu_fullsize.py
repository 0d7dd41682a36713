"""GPU (-m gpu): BASELINE.json's full-size configurations against the oracle.

Configs 2, 3 and one rank of config 4: EVERY view, FULL frame, STD bit-exact against the threaded C oracle (it needs about a
second per configuration on the GPU box's host cores), a handful of full-frame TEN_WM views within one LSB of the oracle's M16
model in both view layouts, plus the size-independent properties (identity under one-hot weights, view-range invariance).
Config 5 (15×15 @4K, 7.5 GB of inputs): the light field is generated on the host plane by plane with the oracle's generator,
and oracle row bands (top edge, interior, bottom edge) check the fixed-focus renders, the focus map and the all-focus renders
that read it."""
from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 16)


def _shifted(img, ox, oy):
    H, W = img.shape[:2]
    ys = np.clip(np.arange(H) + oy, 0, H - 1)
    xs = np.clip(np.arange(W) + ox, 0, W - 1)
    out = img[ys][:, xs].copy()
    out[..., 3] = 255
    return out


def _host_lf(oracle_c, n, W, H):
    lf = np.empty((n, H, W, 4), dtype=np.uint8)
    with ThreadPoolExecutor(max_workers=THREADS) as ex:
        list(ex.map(lambda g: lf.__setitem__(g, oracle_c.synthetic_plane(g, W, H, SEED)), range(n)))
    return lf


# (cols, rows, W, H, views rendered, trajectory, focus, aspect, total views of the job, rank, world)
CONFIGS = [
    (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 64, 0, 1),          # BASELINE config 2
    (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 45, 0, 1),    # config 3: 45-view quilt sweep
    (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 256, 3, 8),         # config 4: rank 3 of 8, views [96,128) of 256
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=["config2_8x8_1080p_64v", "config3_15x15_1080p_45v", "config4_8x8_4k_rank3of8"])
def test_full_frame_parity(cfg, gpu, oracle_c):
    cols, rows, W, H, V, traj, focus, aspect, total_views, rank, world = cfg
    n = cols * rows
    hp, v_first, v_last = gpu.rank_params(cols, rows, W, H, traj, focus, 0.0, 3.0, aspect, total_views, world, rank)
    assert v_last - v_first == V
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.fill_synthetic(SEED)
    probe = [0, V // 2, V - 1]

    # (1) identity: with one-hot weights view v is image picks[v] shifted by its integer offset, clamped at the edges
    picks = (np.arange(V) * max(1, n // V)) % n
    onehot = np.zeros((V, n), np.uint16)
    onehot[np.arange(V), picks] = 0x3c00  # fp16 1.0
    real_weights = hp.weights
    hp.weights = onehot
    ctx.set_params(hp)
    for method in ("STD", "TEN_WM"):
        ctx.render(method)
        ctx.sync()
        for v in probe:
            g = int(picks[v])
            want = _shifted(oracle_c.synthetic_plane(g, W, H, SEED), int(hp.focused_offsets[g, 0]), int(hp.focused_offsets[g, 1]))
            assert (ctx.download_view(v) == want).all(), (method, v)

    # (2) the real weights: every view, full frame, STD bit-exact
    hp.weights = real_weights
    ctx.set_params(hp)
    lf = _host_lf(oracle_c, n, W, H)
    want_std = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=THREADS)
    ctx.render("STD")
    ctx.sync()
    std_probe = {}
    for v in range(V):
        got = ctx.download_view(v)
        assert (got == want_std[v]).all(), ("STD", v, int((got != want_std[v]).sum()))
        if v in probe:
            std_probe[v] = got
    del want_std

    # (3) TEN_WM: full frames of the probe views within one LSB of M16, in both view layouts (identical bytes)
    want_ten = {v: oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, threads=THREADS)[v] for v in probe}
    del lf
    ten_views = {}
    for layout in ("rgba", "planar"):
        ctx.set_output_layout(layout)
        ctx.render("TEN_WM")
        ctx.sync()
        for v in probe:
            got = ctx.download_view(v)
            assert np.abs(got.astype(int) - want_ten[v].astype(int)).max() <= 1, ("TEN_WM", layout, v)
            assert (got[..., 3] == 255).all()
            if layout == "rgba":
                ten_views[v] = got
            else:
                assert (got == ten_views[v]).all(), ("planar layout differs from RGBA layout", v)
            # RN of the fp32 sum vs fp16-truncation of the same sum: at most one step apart, everywhere
            assert np.abs(std_probe[v].astype(int) - got.astype(int)).max() <= 1
        # a one-view range renders the same bytes as the full launch
        ctx.render("TEN_WM", v0=V // 2, v1=V // 2 + 1)
        ctx.sync()
        assert (ctx.download_view(V // 2) == ten_views[V // 2]).all()
    ctx.close()


def test_config5_15x15_4k_bands(gpu, oracle_c):
    """BASELINE config 5: 15×15 @3840×2160, 64 views, the focus sweep's parameters (scripts/focusMapCompare.sh: -s 7, focus 0.22,
    range 0.17).  Oracle row bands of the fixed-focus renders, of the focus maps, and of the all-focus renders reading them."""
    cols = rows = 15
    W, H, V, n = 3840, 2160, 64, 225
    traj, focus, rng, effect, aspect = "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783
    hp = gpu.build_params(cols, rows, W, H, traj, focus, rng, effect, aspect, V)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.fill_synthetic(SEED)
    ctx.set_params(hp)
    lf = _host_lf(oracle_c, n, W, H)                       # 7.5 GB on the host, generated plane by plane
    probe = [0, 31, 63]
    bands = ((0, 2), (H // 2 - 1, H // 2 + 1), (H - 2, H))

    # fixed focus: STD bit-exact, TEN_WM ≤ 1 LSB (both layouts), on the bands
    ctx.render("STD")
    ctx.sync()
    std = {v: ctx.download_view(v) for v in probe}
    ten = {}
    for layout in ("rgba", "planar"):
        ctx.set_output_layout(layout)
        ctx.render("TEN_WM")
        ctx.sync()
        ten[layout] = {v: ctx.download_view(v) for v in probe}
    for v in probe:
        assert (ten["planar"][v] == ten["rgba"][v]).all()
        for y0, y1 in bands:
            ref = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1), threads=THREADS)
            assert (std[v][y0:y1] == ref[v, y0:y1]).all(), ("STD", v, y0)
            ref = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1), threads=THREADS)
            assert np.abs(ten["rgba"][v][y0:y1].astype(int) - ref[v, y0:y1].astype(int)).max() <= 1, ("TEN_WM", v, y0)
    ctx.set_output_layout("rgba")

    # the focus map: both estimate implementations agree everywhere, and equal the oracle on the bands
    maps = {}
    for variant in ("factored", "factored_direct", "lds"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.focus_map()   # a second call reuses the workspace and the side stream
        ctx.sync()
        maps[variant] = (ctx.download_map(0), ctx.download_map(1))
    assert (maps["factored"][0] == maps["lds"][0]).all(), int((maps["factored"][0] != maps["lds"][0]).sum())
    assert (maps["factored"][1] == maps["lds"][1]).all()
    assert (maps["factored_direct"][0] == maps["lds"][0]).all() and (maps["factored_direct"][1] == maps["lds"][1]).all()
    map0, map1 = maps["factored"]
    assert len(np.unique(map0[..., 0])) > 4
    ctx.set_variant("FOCUS", "auto")
    for y0, y1 in bands:
        ref0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, rows=(y0, y1), threads=THREADS)
        assert (map0[y0:y1] == ref0[y0:y1]).all(), ("map 0", y0, int((map0[y0:y1] != ref0[y0:y1]).sum()))
        ref1 = oracle_c.focus_filter(map0, hp.block_radius, rows=(y0, y1), threads=THREADS)   # the filter of the GPU's (verified) map 0
        assert (map1[y0:y1] == ref1[y0:y1]).all(), ("map 1", y0)

    # all-focus renders: STD reads map 1, TEN_WM map 0 (the reference's kernels, src/kernels.cu:326 / :430)
    ctx.render("STD", all_focus=True)
    ctx.sync()
    af_std = {v: ctx.download_view(v) for v in probe}
    ctx.render("TEN_WM", all_focus=True)
    ctx.sync()
    af_ten = {v: ctx.download_view(v) for v in probe}
    for v in probe:
        for y0, y1 in bands:
            ref = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1), threads=THREADS,
                                     all_focus=True, map_plane=map1, focus=hp.focus, rng=hp.range)
            assert (af_std[v][y0:y1] == ref[v, y0:y1]).all(), ("all-focus STD", v, y0)
            ref = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1), threads=THREADS,
                                     all_focus=True, map_plane=map0, focus=hp.focus, rng=hp.range)
            assert np.abs(af_ten[v][y0:y1].astype(int) - ref[v, y0:y1].astype(int)).max() <= 1, ("all-focus TEN_WM", v, y0)

    # the planar view layout at full size (round 4: every kernel of the default variants writes the byte planes itself — blend_stdx, and
    # blend_stdxa / blend_persist with quad transposes in their epilogues): whole frames byte-identical to the RGBA layout's, no scratch copy
    ctx.set_output_layout("planar")
    for what, all_focus, method, want in (("STD", False, "STD", std), ("all-focus STD", True, "STD", af_std), ("all-focus TEN_WM", True, "TEN_WM", af_ten)):
        ctx.render(method, all_focus=all_focus)
        ctx.sync()
        for v in probe:
            got = ctx.download_view(v)
            assert (got == want[v]).all(), (what, "planar layout", v, int((got != want[v]).sum()))
    ctx.close()


def test_focus_map_1080p_variants_agree(gpu):
    """The focus map at 1080p (8×8): the factored estimate and the LDS-staged kernel — independent implementations, each checked
    against the oracle at small sizes and on config 5's bands — produce identical maps."""
    cols, rows, W, H = 8, 8, 1920, 1080
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.fill_synthetic(SEED)
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 8)
    ctx.set_params(hp)
    maps = {}
    for variant in ("factored", "lds"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.sync()
        maps[variant] = (ctx.download_map(0), ctx.download_map(1))
    assert (maps["factored"][0] == maps["lds"][0]).all(), int((maps["factored"][0] != maps["lds"][0]).sum())
    assert (maps["factored"][1] == maps["lds"][1]).all()
    ctx.close()
