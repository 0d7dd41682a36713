"""GPU (-m gpu): BASELINE.json's full-size configurations through size-independent properties.

The oracle needs seconds per view at 1080p, so full frames are checked through (1) identity under one-hot weights — the
output must be the clamped, shifted input image, which numpy slices in milliseconds and which exercises the integer
warp, clamp-to-edge and both quantisers at every pixel; (2) oracle row bands (top edge, interior, bottom edge);
(3) view-range invariance; (4) STD vs TEN_WM agreement within one step."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


def _shifted(img, ox, oy):
    H, W = img.shape[:2]
    ys = np.clip(np.arange(H) + oy, 0, H - 1)
    xs = np.clip(np.arange(W) + ox, 0, W - 1)
    out = img[ys][:, xs].copy()
    out[..., 3] = 255
    return out


# (cols, rows, W, H, views rendered, trajectory, focus, aspect, total views of the job, rank, world, oracle row bands?)
CONFIGS = [
    (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 64, 0, 1, True),          # BASELINE config 2
    (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 45, 0, 1, True),    # config 3: 45-view quilt sweep
    (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 256, 3, 8, True),         # config 4: rank 3 of 8, views [96,128) of 256
    (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 64, 0, 1, False),  # config 5 (7.5 GB grid: no host copy)
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=["config2_8x8_1080p_64v", "config3_15x15_1080p_45v", "config4_8x8_4k_rank3of8",
                                              "config5_15x15_4k_64v"])
def test_full_size_properties(cfg, gpu, oracle_c):
    cols, rows, W, H, V, traj, focus, aspect, total_views, rank, world, with_bands = cfg
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

    # (2) oracle row bands with the real weights
    hp.weights = real_weights
    ctx.set_params(hp)
    lf = oracle_c.synthetic_lf(n, W, H, SEED) if with_bands else None
    ctx.render("STD")
    ctx.sync()
    std_views = {v: ctx.download_view(v) for v in probe}
    ctx.render("TEN_WM")
    ctx.sync()
    ten_views = {v: ctx.download_view(v) for v in probe}
    for y0, y1 in ((0, 2), (H // 2, H // 2 + 2), (H - 2, H)) if with_bands else ():
        for v in probe:
            ref = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1))
            assert (std_views[v][y0:y1] == ref[v, y0:y1]).all(), ("STD", v, y0)
            ref = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, v0=v, v1=v + 1, rows=(y0, y1))
            assert np.abs(ten_views[v][y0:y1].astype(int) - ref[v, y0:y1].astype(int)).max() <= 1, ("TEN_WM", v, y0)

    # (4) RN of the fp32 sum vs fp16-truncation of the same sum: at most one step apart, everywhere
    for v in probe:
        assert np.abs(std_views[v].astype(int) - ten_views[v].astype(int)).max() <= 1

    # (3) a one-view range renders the same bytes as the full launch
    ctx.render("TEN_WM", v0=V // 2, v1=V // 2 + 1)
    ctx.sync()
    assert (ctx.download_view(V // 2) == ten_views[V // 2]).all()
    ctx.close()


@pytest.mark.parametrize("shape", [(8, 8, 1920, 1080), (15, 15, 3840, 2160)], ids=["8x8_1080p", "15x15_4k"])
def test_focus_map_full_size_variants_agree(shape, gpu):
    """The focus map at BASELINE's sizes (config 5's all-focus parameters): the factored estimate (range images, line images
    for flagged rows / columns, tap-by-tap keys for the rest) and the LDS-staged kernel — independent implementations, each
    checked against the oracle at small sizes — must produce identical maps, and the all-focus render must read them."""
    cols, rows, W, H = shape
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.fill_synthetic(SEED)
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 8)
    ctx.set_params(hp)
    maps = {}
    for variant in ("factored", "lds"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.focus_map()   # a second call reuses the workspace and the side stream
        ctx.sync()
        maps[variant] = (ctx.download_map(0), ctx.download_map(1))
    assert (maps["factored"][0] == maps["lds"][0]).all(), int((maps["factored"][0] != maps["lds"][0]).sum())
    assert (maps["factored"][1] == maps["lds"][1]).all()
    assert len(np.unique(maps["factored"][0][..., 0])) > 4
    ctx.set_variant("FOCUS", "auto")
    ctx.render("TEN_WM", all_focus=True)
    ctx.render("STD", all_focus=True)
    ctx.sync()
    ctx.close()
