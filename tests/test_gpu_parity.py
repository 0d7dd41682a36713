"""GPU (-m gpu): the HIP path through the C-ABI against the CPU oracle and the golden fixtures.

Contract:
  * integer / float warp coordinates: bit-exact;
  * STD: u8 outputs and fp32 accumulators bit-exact;
  * TEN_WM: u8 within 1 LSB of the fp16-accumulate model M16, pre-quantisation within 1e-3 (normalised to [0,1]) of
    the exact fp64 blend; the per-batch debug mode reproduces M16.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, SEED, SMALL_CASES

pytestmark = pytest.mark.gpu

TEN_TOL_LSB = 1          # |u8(HIP) - u8(M16)| ≤ 1
TEN_TOL_PREQUANT = 1e-3  # |prequant(HIP)/255 - exact/255| ≤ 1e-3


def _ctx(gpu, cols, rows, W, H, hp, lf=None, seed=SEED, flags=0):
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    if lf is None:
        ctx.fill_synthetic(seed)
    else:
        ctx.upload_grid(lf)
    ctx.set_params(hp, flags)
    return ctx


def test_mfma_fragment_maps_and_fp16_subnormals(gpu):
    """One v_mfma_f32_32x32x16_f16 on exact integer data with an asymmetric A: checks the A/B/D lane maps the kernels
    rely on, and that fp16 subnormal operands (pixel bytes b·2^-24, tiny weights) are not flushed."""
    ctx = gpu.Context(0)
    a = np.zeros((32, 16), np.float16)
    b = np.zeros((16, 32), np.float16)
    for i in range(32):
        for k in range(16):
            a[i, k] = (i * 3 + k * 5) % 17 - 8      # small integers, asymmetric
    for k in range(16):
        for j in range(32):
            b[k, j] = (k * 7 + j * 2) % 13 - 6
    c = ctx.debug_mfma_f16(a.view(np.uint16), b.view(np.uint16))
    assert (c == a.astype(np.float64) @ b.astype(np.float64)).all()
    # subnormal pixels × normal weights, exact in fp32: one non-zero product per output element
    a2 = np.zeros((32, 16), np.uint16)
    b2 = np.zeros((16, 32), np.uint16)
    for i in range(32):
        a2[i, i % 16] = 0x3c00 - i * 13          # weights just below 1
    for j in range(32):
        b2[:, j] = (np.arange(16) * 16 + j * 7) % 256  # pixel bytes as fp16 subnormal bit patterns
    c2 = ctx.debug_mfma_f16(a2, b2)
    want = a2.view(np.float16).astype(np.float64) @ b2.view(np.float16).astype(np.float64)
    assert (c2.astype(np.float64) == want).all()
    # subnormal weights (effect 7 produces them) × subnormal pixels: 2^-24 · 255·2^-24, still exact in fp32
    a3 = np.zeros((32, 16), np.uint16)
    a3[:, 0] = np.arange(1, 33)
    c3 = ctx.debug_mfma_f16(a3, b2)
    want3 = a3.view(np.float16).astype(np.float64) @ b2.view(np.float16).astype(np.float64)
    assert (c3.astype(np.float64) == want3).all()
    ctx.close()


def test_pk_minmax3_f16_on_bytes(gpu):
    """The focus-map range passes reduce two views per instruction with v_pk_minimum3_f16 / v_pk_maximum3_f16 on u16 lanes that hold
    bytes — fp16 SUBNORMAL bit patterns, whose float order is their integer order.  That is exact only if the instructions do not
    flush subnormals: all 256³ byte triples, on the device, against integer min / max."""
    ctx = gpu.Context(0)
    assert ctx.debug_pk_minmax3_f16() == 0
    ctx.close()


def test_synthetic_fill_matches_oracle(gpu, oracle_c):
    ctx = gpu.Context(0)
    ctx.set_grid(3, 2, 37, 11)
    ctx.fill_synthetic(1234)
    ctx.sync()
    hp = gpu.build_params(3, 2, 37, 11, "0,0,1,1", 0.0, 0.0, 3.0, 1.0, 6)
    onehot = np.zeros((6, 6), np.uint16)
    onehot[np.arange(6), np.arange(6)] = 0x3c00
    hp.weights = onehot
    hp.focused_offsets = np.zeros((6, 2), np.int32)
    ctx.set_params(hp)
    ctx.render("STD")
    ctx.sync()
    assert (ctx.download_views() == oracle_c.synthetic_lf(6, 37, 11, 1234)).all()
    ctx.close()


@pytest.mark.parametrize("case", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_golden_fixtures(case, gpu, oracle_c):
    name, cols, rows, W, H, V, traj, focus, aspect, effect = case
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    rng = float(g["range"])
    hp = gpu.build_params(cols, rows, W, H, traj, focus, rng, effect, aspect, V)
    assert (hp.weights == g["weights"]).all() and (hp.focused_offsets == g["focused"]).all()
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=g["lf"])
    exact = oracle_c.blend_f64(g["lf"], hp.focused_offsets, hp.offsets, hp.weights)
    for variant in ctx.list_variants("STD"):
        ctx.set_variant("STD", variant)
        ctx.render("STD")
        ctx.sync()
        assert (ctx.download_views() == g["std"]).all(), variant
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM")
        ctx.sync()
        out = ctx.download_views()
        assert np.abs(out.astype(int) - g["ten_m16"].astype(int)).max() <= TEN_TOL_LSB, variant
        assert (out[..., 3] == 255).all()
        # single rounding of an fp32 sum vs single rounding of the exact sum: equal except on razor-edge ties
        assert (out != g["ten_exact"]).mean() < 1e-3, variant
        for v in (0, V - 1):
            pre = ctx.download_prequant("TEN_WM", v)
            assert np.abs(pre / 255.0 - exact[v] / 255.0).max() <= TEN_TOL_PREQUANT, variant
    # focus map + all-focus renders
    ctx.set_variant("STD", "auto")
    ctx.set_variant("TEN_WM", "auto")
    ctx.focus_map()
    ctx.sync()
    assert (ctx.download_map(0) == g["map0"]).all()
    assert (ctx.download_map(1) == g["map1"]).all()
    for variant in ctx.list_variants("STD"):
        ctx.set_variant("STD", variant)
        ctx.render("STD", all_focus=True)
        ctx.sync()
        assert (ctx.download_views() == g["af_std"]).all(), variant
    # default: the reference's maps — Tensors::process<true> reads the unfiltered map 0 (src/kernels.cu:430)
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM", all_focus=True)
        ctx.sync()
        assert np.abs(ctx.download_views().astype(int) - g["af_ten_m16_map0"].astype(int)).max() <= TEN_TOL_LSB, variant
    # opt-in: both methods read the filtered map 1
    ctx.set_params(hp, flags=gpu.LFI_FLAG_UNIFIED_FOCUS_MAP)
    ctx.set_variant("TEN_WM", "auto")
    ctx.render("TEN_WM", all_focus=True)
    ctx.sync()
    assert np.abs(ctx.download_views().astype(int) - g["af_ten_m16"].astype(int)).max() <= TEN_TOL_LSB
    ctx.render("STD", all_focus=True)
    ctx.sync()
    assert (ctx.download_views() == g["af_std"]).all()
    ctx.close()


def test_warp_coordinates_bit_exact(gpu, oracle_c):
    cols = rows = 8
    W, H = 96, 40
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 8)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    map1 = (oracle_c.synthetic_lf(1, W, H, 99)[0]).copy()  # arbitrary per-pixel focus bytes
    ctx.upload_map(1, map1)
    for g in (0, 7, 27, 63):
        assert (ctx.download_coords(g) == oracle_c.warp_coords(g, W, H, hp.focused_offsets, hp.offsets)).all()
        got = ctx.download_coords(g, all_focus=True, map_index=1)
        want = oracle_c.warp_coords(g, W, H, hp.focused_offsets, hp.offsets, True, map1, hp.focus, hp.range)
        assert (got == want).all()
    ctx.close()


def test_std_accumulators_bit_exact_and_ten_per_batch_mode(gpu, oracle_c):
    cols = rows = 8
    W, H, V = 64, 48, 64
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(64, W, H, SEED)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    _, pre = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, return_prequant=True)
    for variant in ctx.list_variants("STD"):
        ctx.set_variant("STD", variant)
        for v in (0, 31, 63):
            assert (ctx.download_prequant("STD", v) == pre[v]).all(), variant
    ctx.close()
    # debug numerics: fp16 re-rounding per 16-image batch reproduces the reference model M16 — byte for byte since round 5 (the inner sums
    # of a batch in double precision on the vector pipe, one rounding to fp16: blend_ten_m16; the matrix pipe's fp32 accumulation made ≈ 1e-4
    # of the bytes differ), fixed focus and all-focus, also on a 15×15 grid (K padded from 225 to 240) with subnormal weights (-s 7)
    ctx = _ctx(gpu, cols, rows, W, H, hp, flags=gpu.LFI_FLAG_TEN_ROUND_PER_BATCH)
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name() == "blend_ten_m16"
    m16, pre16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, return_prequant=True)
    assert (ctx.download_views() == m16).all()
    assert (ctx.download_prequant("TEN_WM", 17) == pre16[17]).all()
    ctx.close()
    cols = rows = 15
    W, H, V = 70, 10, 21
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.1, 0.3, 7.0, 1.783, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED + 1)
    map0 = oracle_c.synthetic_lf(1, W, H, 3)[0]
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=gpu.LFI_FLAG_TEN_ROUND_PER_BATCH)
    ctx.upload_map(0, map0)
    for all_focus in (False, True):
        ctx.render("TEN_WM", all_focus=all_focus)
        ctx.sync()
        want = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, all_focus=all_focus, map_plane=map0,
                                  focus=hp.focus, rng=hp.range)
        assert (ctx.download_views() == want).all(), all_focus
    ctx.close()


@pytest.mark.parametrize("shape", [(8, 8, 130, 9, 64), (15, 15, 70, 6, 45), (3, 3, 256, 256, 1), (2, 5, 31, 33, 70),
                                   (1, 1, 17, 5, 3), (8, 8, 512, 4, 130)],
                         ids=lambda s: "x".join(map(str, s)))
def test_ragged_shapes_all_variants(shape, gpu, oracle_c):
    """width not a multiple of the pixel tile, N not a multiple of 16 (zero-padded K), views not a multiple of 32,
    more than 64 views (several passes), 1 view (BASELINE config 1 shape 3×3@256²), 1×1 grid."""
    cols, rows, W, H, V = shape
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.3, 0.0, 3.0, 1.5, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    std = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    for variant in ctx.list_variants("STD"):
        ctx.set_variant("STD", variant)
        ctx.render("STD")
        ctx.sync()
        assert (ctx.download_views() == std).all(), variant
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM")
        ctx.sync()
        assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, variant
    ctx.close()


def test_offsets_larger_than_image_and_subnormal_weights(gpu, oracle_c):
    cols = rows = 15
    W, H, V = 48, 20, 8
    # focus 3.0: integer offsets up to ±(3·W) — every sample clamps to an edge for the outer images
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 3.0, 0.0, 7.0, 2.0223, V)
    assert np.abs(hp.focused_offsets).max() > W
    assert ((hp.weights & 0x7c00) == 0).any()  # fp16 subnormal weights present (-s 7)
    lf = oracle_c.synthetic_lf(225, W, H, 5)
    ctx = _ctx(gpu, cols, rows, W, H, hp, seed=5)
    for variant in ctx.list_variants("STD"):
        ctx.set_variant("STD", variant)
        ctx.render("STD")
        ctx.sync()
        assert (ctx.download_views() == oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)).all()
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights)
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM")
        ctx.sync()
        assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB
    ctx.close()


def test_view_range_sharding_is_exact(gpu):
    """Rendering [v0,v1) ranges (what each rank of a multi-GPU job does) gives the bytes of the full render."""
    cols = rows = 8
    W, H, V = 128, 24, 64
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    for method in ("STD", "TEN_WM"):
        ctx.render(method)
        ctx.sync()
        full = ctx.download_views()
        for v0, v1 in ((0, 8), (8, 16), (24, 56), (63, 64)):
            ctx.render(method, v0=0, v1=V)  # refill
            ctx.render(method, v0=v0, v1=v1)
            ctx.sync()
            assert (ctx.download_views(v0, v1) == full[v0:v1]).all()
        # a rank that only owns rows [16,24) of the weight matrix
        ctx2 = _ctx(gpu, cols, rows, W, H, hp.rows(16, 24))
        ctx2.render(method)
        ctx2.sync()
        assert (ctx2.download_views() == full[16:24]).all()
        ctx2.close()
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_row_band_sharding_matches_full_render(world, gpu, oracle_c):
    """every rank renders a band of rows from the input rows its warp reaches; the bands together are the
    full render, byte for byte, and each rank holds far fewer input rows than the image has."""
    cols = rows = 8
    W, H, V = 200, 96, 64
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.1, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(64, W, H, SEED)
    # the checker is the ORACLE (STD bit-exact, TEN_WM within one LSB of M16); the library's own full render is compared too, byte for
    # byte — a band must not depend on how the image is split — but a coordinate error shared by both paths would pass that alone
    oracle_views = {"STD": oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8),
                    "TEN_WM": oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, threads=8)}
    full = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    for method in ("STD", "TEN_WM"):
        full.render(method)
        full.sync()
        want = full.download_views()
        got = np.zeros_like(want)
        held = []
        for rank in range(world):
            band = gpu.row_band(H, world, rank)
            in_rows = gpu.input_rows(band, hp.focused_offsets, H)
            held.append(in_rows[1] - in_rows[0])
            ctx = gpu.Context(0)
            ctx.set_grid(cols, rows, W, H)
            ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
            if rank % 2:
                ctx.fill_synthetic(SEED)     # device-side generation of just the held rows
            else:
                ctx.upload_grid(lf)          # whole-image host pointers, only the held rows are copied
            ctx.set_params(hp)
            assert ctx.grid_device_ptr()[1] == 64 * (in_rows[1] - in_rows[0]) * W * 4
            ctx.render(method)
            ctx.sync()
            part = ctx.download_views()
            assert (part[:, :band[0]] == 0).all() and (part[:, band[1]:] == 0).all()
            got |= part
            ctx.close()
        assert (got == want).all(), method
        if method == "STD":
            assert (got == oracle_views["STD"]).all(), "row bands: STD differs from the oracle"
        else:
            assert np.abs(got.astype(int) - oracle_views["TEN_WM"].astype(int)).max() <= TEN_TOL_LSB, "row bands: TEN_WM differs from M16"
        assert max(held) < H
    if True:
        # an input window that misses sampled rows is refused
        ctx = gpu.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.set_row_window(10, 20, 10, 20)
        with pytest.raises(gpu.LfiError, match="does not cover"):
            ctx.set_params(hp)
        ctx.close()
    full.close()


@pytest.mark.parametrize("layout", ["rgba", "planar"])
def test_quilt_download(layout, gpu):
    """lfi_download_quilt: the views as ONE image of tiles, assembled on the device (one kernel, the planar layout expanded on the fly) and
    copied in one rectangle: byte for byte the montage of the views' own downloads (scripts/viewsToQuilt.sh montages the NN.png files).
    lfi_download_quilt_tiles: the same image filled by several contexts, each with the tiles of its views (a trajectory sharded over
    GPUs) — ranges that start and end in the middle of a row of tiles — and under a row window."""
    cols = rows = 3
    W, H, V = 41, 12, 10                      # a width that is not a multiple of four
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.0, V)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    ctx.set_output_layout(layout)
    ctx.render("TEN_WM")
    ctx.sync()
    views = ctx.download_views()
    quilt = ctx.download_quilt(4, 2, v0=1)
    for i in range(8):
        ty, tx = divmod(i, 4)
        assert (quilt[ty * H:(ty + 1) * H, tx * W:(tx + 1) * W] == views[1 + i]).all()
    with pytest.raises(gpu.LfiError, match="quilt needs"):
        ctx.download_quilt(4, 3)
    # the same quilt from "three GPUs": contexts that hold views [0, 3), [3, 8), [8, 10) of the trajectory fill tiles 0-2, 3-7, 8-9 of a 5 x 2 quilt
    want = ctx.download_quilt(5, 2)
    ctx.close()
    got = np.zeros_like(want)
    for v0, v1 in ((0, 3), (3, 8), (8, 10)):
        part = _ctx(gpu, cols, rows, W, H, hp.rows(v0, v1))
        part.set_output_layout(layout)
        part.render("TEN_WM")
        part.sync()
        part.download_quilt_tiles(got, 5, 2, v0, v1 - v0)
        with pytest.raises(gpu.LfiError, match="quilt needs"):
            part.download_quilt_tiles(got, 5, 2, 9, 2)
        part.close()
    assert (got == want).all()
    # a row window: only the band's rows of every tile are written
    band = (3, 9)
    win = gpu.Context(0)
    win.set_grid(cols, rows, W, H)
    in_rows = gpu.input_rows(band, hp.focused_offsets, H)
    win.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
    win.fill_synthetic(SEED)
    win.set_params(hp)
    win.set_output_layout(layout)
    win.render("TEN_WM")
    win.sync()
    got = np.zeros_like(want)
    win.download_quilt_tiles(got, 5, 2, 2, 6, v0=2)
    win.close()
    for i in range(10):
        ty, tx = divmod(i, 5)
        tile = got[ty * H:(ty + 1) * H, tx * W:(tx + 1) * W]
        if 2 <= i < 8:
            assert (tile[band[0]:band[1]] == views[i][band[0]:band[1]]).all() and not tile[:band[0]].any() and not tile[band[1]:].any()
        else:
            assert not tile.any()


def test_error_behaviour(gpu):
    ctx = gpu.Context(0)
    with pytest.raises(gpu.LfiError, match="lfi_set_grid"):
        ctx.render("STD", v1=1)
    ctx.set_grid(2, 2, 8, 8)
    with pytest.raises(gpu.LfiError, match="lfi_set_params"):
        ctx.render("STD", v1=1)
    hp = gpu.build_params(2, 2, 8, 8, "0,0,1,1", 0.1, 0.0, 3.0, 1.0, 4)
    ctx.set_params(hp)
    with pytest.raises(gpu.LfiError, match="does not exist"):  # the reference's runtime_error text
        ctx.render(7)
    with pytest.raises(gpu.LfiError, match="view range"):
        ctx.render("STD", v0=2, v1=9)
    with pytest.raises(gpu.LfiError, match="range must be > 0"):
        ctx.focus_map()
    with pytest.raises(gpu.LfiError, match="unknown kernel variant"):
        ctx.set_variant("STD", "nope")
    with pytest.raises(gpu.LfiError):
        ctx.set_grid(17, 17, 8, 8)  # more than 256 images
    ctx.close()


def test_weights_outside_unit_range_use_the_generic_kernel(gpu, oracle_c):
    """The packed epilogue needs weights in [0, 2); anything else (negative, ≥ 2: not produced by the reference's
    generator but legal through the C-ABI) must fall back to the generic kernel and still saturate like __half2uchar_rz."""
    cols = rows = 4
    W, H, V = 96, 10, 8
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.0, V)
    w = hp.weights.view(np.float16).astype(np.float32)
    w[0] *= 3.0           # sums > 255 → saturate at 255
    w[1] = -w[1]          # negative sums → 0
    w[2, 0] = 2.5         # a single weight ≥ 2
    w[3] *= 0.25
    hp.weights = w.astype(np.float16).view(np.uint16)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights)
    assert (m16[0] == 255)[..., :3].mean() > 0.5 and (m16[1][..., :3] == 0).all()
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM")
        ctx.sync()
        assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, variant
    # in-range but large weights (1 ≤ w < 2) stay on the packed path and saturate there
    hp2 = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.0, V)
    w2 = hp2.weights.view(np.float16).astype(np.float32)
    w2 = w2 * (1.9 / w2.max())
    assert w2.max() < 2.0 and w2.max() > 1.0
    hp2.weights = w2.astype(np.float16).view(np.uint16)
    ctx.set_params(hp2)
    m16b = oracle_c.blend_ten(lf, hp2.focused_offsets, hp2.offsets, hp2.weights)
    assert (m16b == 255).mean() > 0.3
    for variant in ctx.list_variants("TEN_WM"):
        ctx.set_variant("TEN_WM", variant)
        ctx.render("TEN_WM")
        ctx.sync()
        assert np.abs(ctx.download_views().astype(int) - m16b.astype(int)).max() <= TEN_TOL_LSB, variant
    ctx.close()


@pytest.mark.parametrize("shape", [(8, 8, 384, 6), (15, 15, 301, 5), (4, 4, 130, 3)], ids=lambda s: "x".join(map(str, s)))
def test_focus_map_wide_rows(shape, gpu, oracle_c):
    """Rows wide enough for the packed focus-estimate kernel's wide-load path (interior lanes) next to its per-pixel
    clamped path (border lanes, ragged right edge): both must give the oracle's bytes."""
    cols, rows, W, H = shape
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.06, 0.24, 7.0, 2.276, 4)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 21)
    # piecewise-constant content so that the dispersion has real minima (random noise ties everywhere at 255)
    lf = (lf // 64 * 64).astype(np.uint8)
    lf[..., 3] = 255
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    want0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    for variant in ctx.list_variants("FOCUS"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.sync()
        assert (ctx.download_map(0) == want0).all(), variant
        assert (ctx.download_map(1) == oracle_c.focus_filter(want0, hp.block_radius)).all(), variant
    # an all-black grid exercises the reference's FLT_MIN initial maximum (src/kernels.cu:178): every candidate ties at
    # 9*FLT_MIN, the first one wins → map 0 everywhere
    ctx2 = _ctx(gpu, cols, rows, W, H, hp, lf=np.zeros_like(lf))
    want_black = oracle_c.focus_estimate(np.zeros_like(lf), hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    for variant in ctx2.list_variants("FOCUS"):
        ctx2.set_variant("FOCUS", variant)
        ctx2.focus_map()
        ctx2.sync()
        assert (ctx2.download_map(0) == want_black).all(), variant
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("radius", [None, (3, 1), (5, 2)], ids=["radius_ref", "radius_3x1", "radius_5x2"])
def test_focus_map_realistic_geometry(radius, gpu, oracle_c):
    """The factored estimate (per-candidate range image + exact path for flagged columns / rows) at a size where most pixels
    take the factored path and the flagged bands are real bands: shifts of tens of pixels in both directions, several block
    rows per XCD stripe.  Odd radii take the one-pixel-per-lane pick kernel.  Every variant must give the oracle's bytes."""
    cols, rows, W, H = 8, 8, 600, 300
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 4)
    if radius is not None:
        hp.block_radius = np.array(radius, np.int32)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 33)
    lf = (lf // 32 * 32).astype(np.uint8)
    lf[:, :40, :70, :3] = 0      # an all-black corner: FLT_MIN taps next to ordinary ones
    lf[..., 3] = 255
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    want0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    assert len(np.unique(want0)) > 8
    for variant in ("factored", "factored_direct", "lds"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.sync()
        got = ctx.download_map(0)
        assert (got == want0).all(), (variant, int((got != want0).sum()))
        assert (ctx.download_map(1) == oracle_c.focus_filter(want0, hp.block_radius)).all(), variant
    ctx.close()



@pytest.mark.parametrize("case", [(150, 70, (37, 25)), (333, 90, (100, 60)), (64, 32, (19, 9)), (65, 33, (20, 20)), (70, 40, (1290, 30)), (90, 50, (1000, 700))],
                         ids=lambda c: "%dx%d_r%dx%d" % (c[0], c[1], c[2][0], c[2][1]))
def test_focus_filter_window_sizes(case, gpu, oracle_c):
    """Map 1 = the box mean of map 0 over (2·radius/10)² taps (src/kernels.cu:260-280), clamped at the borders: windows of 2 × 2 up to 20 × 12
    taps from LDS (focus_filter_tiled: separable integer sums), tiles cut by the right and lower border, and windows too wide for the tiled
    kernel's counters (radius 1290) or for the LDS (1000 × 700): the plain kernel — byte for byte the oracle's filter of the same map 0."""
    W, H, rad = case
    cols, rows = 3, 3
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.2, 0.3, 3.0, 1.783, 2)
    hp.block_radius = np.array(rad, np.int32)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 5 + W)
    lf[..., 3] = 255
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    ctx.focus_map()
    ctx.sync()
    map0 = ctx.download_map(0)
    assert len(np.unique(map0)) > 4
    want1 = oracle_c.focus_filter(map0, hp.block_radius)
    assert (ctx.download_map(1) == want1).all(), int((ctx.download_map(1) != want1).sum())
    ctx.close()


@pytest.mark.parametrize("case", [(1, 1, 64, 20, 0.2, 0.3, None), (1, 2, 100, 33, 0.1, 0.2, None), (2, 1, 7, 5, 0.3, 0.5, (1, 1)), (3, 3, 1024, 70, 0.22, 0.17, None),
                                  (8, 8, 2048, 96, 0.05, 0.04, (20, 10)), (15, 15, 640, 64, 0.22, 0.17, (6, 4)), (6, 6, 513, 40, -0.3, 0.6, None),
                                  (9, 9, 300, 48, 0.0, 1.0, (11, 5)), (4, 4, 4096, 36, 0.22, 0.17, None), (5, 3, 190, 130, 0.5, 0.01, None)],
                         ids=lambda c: "x".join(str(v) for v in c[:4]))
def test_focus_map_edge_shapes_both_range_passes(case, gpu, oracle_c):
    """Shapes at the edges of focus_range_t's dispatch (round 5): one and two sampled views (a single step, an odd tail), images narrower than
    a tile and as wide as the stripes' XCD mapping, shifts whose span per candidate group forces groups of four or the fall-back to
    focus_range (range 1.0), negative focus, radii from 1 to 20 — `factored` (focus_range_t where its preconditions hold) and
    `factored_direct` (focus_range), twice each (the padded planes kept), maps 0 and 1 against the oracle."""
    cols, rows, W, H, f, r, rad = case
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", f, r, 3.0, 1.783, 2)
    if rad:
        hp.block_radius = np.array(rad, np.int32)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 11 + cols + W)
    lf = (lf // 16 * 16).astype(np.uint8)
    lf[..., 3] = 255
    want0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    want1 = oracle_c.focus_filter(want0, hp.block_radius)
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    for variant in ("factored", "factored_direct"):
        ctx.set_variant("FOCUS", variant)
        for _ in range(2):
            ctx.focus_map()
            ctx.sync()
        assert (ctx.download_map(0) == want0).all() and (ctx.download_map(1) == want1).all(), variant
    ctx.close()


@pytest.mark.parametrize("case", [(129, 50, (2, 1)), (257, 70, (64, 3)), (300, 70, (66, 2)), (640, 200, (8, 6)), (1000, 33, (38, 22)), (130, 40, (40, 30)),
                                  (256, 97, (4, 1)), (2050, 48, (20, 2))],
                         ids=lambda c: "%dx%d_r%dx%d" % (c[0], c[1], c[2][0], c[2][1]))
def test_focus_map_pick_and_keys_at_their_dispatch_edges(case, gpu, oracle_c):
    """focus_pick_sep (round 5: the tap block taken apart, sixteen waves on rows radius_y apart, two rows of the map per wave) and the comb form
    of focus_line_keys' column part at the edges of their geometry: widths one past a multiple of 128, radius_x of 2 (one lane of extra columns),
    64 (the last radius the kernel takes: taps two whole waves further on) and 66 (focus_pick<2> instead), radius_y of 1 (a band of 32 rows) and
    above the image's height (one band, most waves idle; combs with a single row), heights that end inside a band.  Quantised inputs with a black
    region (ties, FLT_MIN taps), a corner where rows AND columns are flagged; `factored` and `factored_direct` (the other pick), maps 0 and 1."""
    W, H, rad = case
    cols, rows = 6, 5
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.3, 7.0, 1.783, 2)
    hp.block_radius = np.array(rad, np.int32)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 77 + W + rad[0])
    lf = (lf // 16 * 16).astype(np.uint8)
    lf[:, : H // 3, W // 2 :, :3] = 0
    lf[..., 3] = 255
    want0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    want1 = oracle_c.focus_filter(want0, hp.block_radius)
    assert len(np.unique(want0[..., 0])) > 2
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    for variant in ("factored", "factored_direct"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map()
        ctx.sync()
        got0 = ctx.download_map(0)
        assert (got0 == want0).all(), (variant, int((got0 != want0).any(-1).sum()))
        assert (ctx.download_map(1) == want1).all(), variant
    ctx.close()


def test_focus_map_padded_planes_are_kept_between_calls(gpu, oracle_c):
    """lfi_focus_map keeps the padded copies of the sampled images while the inputs are unchanged (a focus sweep over one light field pads
    once).  Every call must still give the oracle's bytes: the same parameters again (planes reused), a smaller and a larger focus
    (reused / rebuilt with more padding), another block radius (rebuilt), one sampled image replaced (rebuilt), a write through the raw
    grid pointer announced by lfi_grid_modified (rebuilt)."""
    cols, rows, W, H = 8, 8, 333, 90
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 77)
    lf = (lf // 16 * 16).astype(np.uint8)
    lf[..., 3] = 255

    def params(focus, rng, radius=None):
        hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", focus, rng, 7.0, 1.783, 4)
        if radius is not None:
            hp.block_radius = np.array(radius, np.int32)
        return hp

    def check(ctx, hp, lf_now, what):
        ctx.set_params(hp)
        ctx.focus_map()
        ctx.sync()
        want0 = oracle_c.focus_estimate(lf_now, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
        got = ctx.download_map(0)
        assert (got == want0).all(), (what, int((got != want0).sum()))
        assert (ctx.download_map(1) == oracle_c.focus_filter(want0, hp.block_radius)).all(), what

    hp = params(0.22, 0.17)
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    ctx.set_variant("FOCUS", "factored")
    check(ctx, hp, lf, "first call")
    check(ctx, hp, lf, "same parameters: planes reused")
    check(ctx, params(0.10, 0.05), lf, "smaller shifts: planes reused")
    check(ctx, params(0.22, 0.17), lf, "back")
    check(ctx, params(0.9, 0.6), lf, "larger shifts: rebuilt")
    check(ctx, params(0.22, 0.17, (3, 1)), lf, "another block radius")
    lf2 = lf.copy()
    g = int(hp.focus_map_ids[1])
    lf2[g] = lf2[g][::-1, ::-1]              # one of the sampled images, replaced through the library
    ctx.upload_image(g, lf2[g])
    check(ctx, params(0.22, 0.17, (3, 1)), lf2, "a sampled image replaced")
    check(ctx, params(0.22, 0.17, (3, 1)), lf2, "and reused again")
    ctx.close()


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        cols, rows = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if cols * rows < 2:
            cols = 2
        W = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 160, 257, 300]))
        H = int(rng.integers(1, 12))
        V = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 70]))
        focus = float(rng.choice([0.0, 0.05, 0.23, 0.6, -0.3]))
        traj = rng.choice(["0,0,1,1", "0.071,0.071,0.93,0.93", "1,0,0,1", "0.5,0.5,0.5,0.5"])
        cases.append((f"r{i}_{cols}x{rows}_{W}x{H}_v{V}", cols, rows, W, H, V, str(traj), focus))
    return cases


@pytest.mark.parametrize("case", _random_cases(16, 2025), ids=lambda c: c[0])
def test_random_shapes_default_kernels(case, gpu, oracle_c):
    """Ragged widths around the 32- and 128-pixel tile edges, one-row images, image counts that are not multiples of 16, view
    counts around the 32- and 64-view pass edges, and a view sub-range: the default kernels (blend_wave / blend_persist behind
    `auto`) against the oracle — STD bit-exact, TEN_WM within one LSB of the fp16-accumulate model."""
    name, cols, rows, W, H, V, traj, focus = case
    hp = gpu.build_params(cols, rows, W, H, traj, focus, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 77)
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    want_std = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    want_ten = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16)
    ctx.render("STD")
    ctx.sync()
    assert (ctx.download_views() == want_std).all()
    ctx.render("TEN_WM")
    ctx.sync()
    assert np.abs(ctx.download_views().astype(int) - want_ten.astype(int)).max() <= TEN_TOL_LSB
    if V >= 3:      # a view sub-range leaves the other views untouched
        before = ctx.download_views()
        v0, v1 = 1, V - 1
        ctx.render("STD", v0=v0, v1=v1)
        ctx.sync()
        after = ctx.download_views()
        assert (after[v0:v1] == want_std[v0:v1]).all()
        assert (after[:v0] == before[:v0]).all() and (after[v1:] == before[v1:]).all()
    ctx.close()


def _f16_bits(x):
    return np.asarray(x, np.float16).view(np.uint16)


@pytest.mark.parametrize("kind", ["ties_everywhere", "flat_images", "sum_1p999", "sum_above_2", "tiny_weights"])
def test_std_rounding_band_adversarial(kind, gpu, oracle_c):
    """STD through blend_planar<STDF> decides most bytes from the MFMA sum and recomputes those near x.5 with the chain.  Inputs
    built to sit in or on the band: exact ties for every pixel (two weights of 0.5, odd pixel differences) — every lane queues
    many sums —, flat images, weights summing to just under 2 (sums up to 510), above 2 (the exact-MFMA kernel must take over),
    and weights deep in the fp16 subnormals.  Bit-exact against the oracle in every case."""
    cols, rows, W, H, V = 8, 8, 200, 7, 64
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.1, 0.0, 3.0, 1.783, V)
    rng = np.random.default_rng(5)
    lf = oracle_c.synthetic_lf(n, W, H, 9)
    w = np.zeros((V, n), np.float32)
    if kind == "ties_everywhere":
        for v in range(V):
            a, b = rng.choice(n, 2, replace=False)
            w[v, a] = w[v, b] = 0.5
        lf[..., :3] = (lf[..., :3] // 2) * 2          # even values …
        lf[::2, :, :, :3] += 1                         # … and odd ones in every other image: half of the pairs tie at x.5
    elif kind == "flat_images":
        w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
        w /= w.sum(1, keepdims=True)
        lf[..., :3] = rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8)
    elif kind == "sum_1p999":
        w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
        w *= 1.99 / w.sum(1, keepdims=True)
    elif kind == "sum_above_2":
        w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
        w *= 2.6 / w.sum(1, keepdims=True)
    else:
        w = (rng.random((V, n)) * 3e-5).astype(np.float32)   # fp16 subnormals and the smallest normals
        w[:, 0] = 0.75
    hp.weights = _f16_bits(w)
    lf[..., 3] = 255
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    for flags in (0, gpu.LFI_FLAG_STD_MEASURED_BAND):          # the default (analytic) band and the narrower measured one
        ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=flags)
        for variant in ("auto", "wave_m2_nt", "persist_m2_nt") if flags == 0 else ("auto",):
            ctx.set_variant("STD", variant)
            ctx.render("STD")
            ctx.sync()
            got = ctx.download_views()
            assert (got == want).all(), (kind, variant, flags, int((got != want).sum()))
        # the planar view layout: blend_stdx with ONE chunk of images writes the byte planes itself (round 4; blend_planar<STDF> would need an
        # RGBA scratch copy of the views and a conversion pass)
        ctx.set_variant("STD", "auto")
        ctx.set_output_layout("planar")
        ctx.render("STD")
        ctx.sync()
        if kind != "sum_above_2":   # (weights summing above 2: the exact kernel through the scratch copy, as in the RGBA layout)
            assert ctx.last_kernel_name() == "blend_stdx<STD>" and ctx.memory_info().workspace_bytes == 0
        got = ctx.download_views()
        assert (got == want).all(), (kind, "planar views", flags, int((got != want).sum()))
        ctx.close()


@pytest.mark.parametrize("cols,rows,W,H,V,kind", [
    (9, 9, 200, 6, 64, "random"),            # two chunks of images (81 → k_pad 96: a short second chunk)
    (12, 12, 131, 5, 37, "random"),          # three chunks; ragged width, a wave with five views, an idle wave
    (15, 15, 300, 4, 64, "random"),          # four chunks (BASELINE configs 3 and 5)
    (15, 15, 140, 3, 70, "random"),          # two launches of views (64 + 6)
    (13, 10, 257, 3, 64, "ties_everywhere"), # every sum an exact tie: the queue overflows, the spill path computes from global memory
    (15, 15, 128, 3, 16, "flat_images"),
    (11, 11, 260, 4, 48, "sum_1p999"),       # sums up to 510: the top binade of the band
    (15, 15, 200, 3, 33, "tiny_weights"),
])
def test_std_band_method_over_several_chunks(cols, rows, W, H, V, kind, gpu, oracle_c):
    """STD on grids of more than 64 images goes through blend_stdx: fp16-MFMA sums over all chunks, the sums inside the rounding band
    recomputed with the chain from a second fetch of the tile's chunks.  Bit-exact against the oracle on random inputs and on inputs
    built to sit on the band; the exact-fp32 MFMA kernel and the analytic band agree."""
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.1, 0.0, 3.0, 1.783, V)
    rng = np.random.default_rng(n + V)
    lf = oracle_c.synthetic_lf(n, W, H, 9 + n)
    if kind != "random":
        w = np.zeros((V, n), np.float32)
        if kind == "ties_everywhere":
            for v in range(V):
                a, b = rng.choice(n, 2, replace=False)
                w[v, a] = w[v, b] = 0.5
            lf[..., :3] = (lf[..., :3] // 2) * 2
            lf[::2, :, :, :3] += 1
        elif kind == "flat_images":
            w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
            w /= w.sum(1, keepdims=True)
            lf[..., :3] = rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8)
        elif kind == "sum_1p999":
            w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
            w *= 1.99 / w.sum(1, keepdims=True)
        else:
            w = (rng.random((V, n)) * 3e-5).astype(np.float32)
            w[:, n // 2] = 0.75
        hp.weights = _f16_bits(w)
    lf[..., 3] = 255
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    for flags in (0, gpu.LFI_FLAG_STD_ANALYTIC_BAND, gpu.LFI_FLAG_SINGLE_SWEEP_DIRECTION):
        ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=flags)
        ctx.render("STD")
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_stdx<STD>"
        got = ctx.download_views()
        assert (got == want).all(), (kind, flags, int((got != want).sum()))
        if flags == 0:
            ctx.render("STD")          # the second launch walks the image in the other direction
            ctx.sync()
            assert (ctx.download_views() == want).all(), (kind, "reverse sweep")
            v0, v1 = V // 3, V // 3 + min(20, V - V // 3)
            ctx.render("STD", v0=v0, v1=v1)
            ctx.sync()
            assert (ctx.download_views(v0, v1) == want[v0:v1]).all(), (kind, "view range")
            ctx.set_output_layout("planar")      # byte planes written by blend_stdx itself (round 4; before: RGBA scratch + conversion)
            ctx.render("STD")
            ctx.sync()
            assert ctx.last_kernel_name() == "blend_stdx<STD>"
            assert ctx.memory_info().workspace_bytes == 0, "no RGBA scratch copy of the views"
            assert (ctx.download_views() == want).all(), (kind, "planar layout")
            ctx.render("STD")                    # the other sweep direction
            ctx.render("STD", v0=v0, v1=v1)      # and a view range over it
            ctx.sync()
            assert (ctx.download_views() == want).all(), (kind, "planar layout, reverse sweep + view range")
        ctx.close()


@pytest.mark.parametrize("cols,rows,W,H,V,kind", [
    (8, 8, 200, 6, 64, "random"),             # one chunk: no C units at all
    (3, 3, 97, 5, 5, "random"),               # nine images: one short chunk, ragged width, five views
    (9, 9, 260, 5, 64, "random"),             # two chunks
    (12, 12, 131, 4, 37, "random"),           # three chunks; ragged width
    (15, 15, 300, 4, 64, "random"),           # four chunks (BASELINE config 5)
    (15, 15, 140, 3, 70, "random"),           # two launches of views (64 + 6)
    (13, 10, 257, 3, 64, "ties_everywhere"),  # every sum an exact tie: the queue overflows, the spill path computes from global memory
    (11, 11, 260, 4, 48, "sum_1p999"),        # sums up to 510
    (15, 15, 200, 3, 33, "tiny_weights"),
])
def test_all_focus_std_band_method(cols, rows, W, H, V, kind, gpu, oracle_c):
    """All-focus STD by the band method: fp16-MFMA sums of the per-pixel gathered samples over all chunks, the sums inside the rounding band
    recomputed with the chain: blend_stdxa (128-pixel tiles; chunks 2 and 3 of a tile are gathered a second time for the chain).  For three
    or four chunks blend_afs (round 4: 64-pixel tiles whose whole stack stays in LDS, every sample gathered once; variant
    "filtered_gather_once") must give the same bytes.  Bit-exact against the
    oracle on random inputs and on inputs built to sit on the band, with noisy and blocky focus maps, the analytic band, a view range, a row
    band and the planar view layout."""
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.05, 0.3, 3.0, 1.783, V)
    rng = np.random.default_rng(n + V)
    lf = oracle_c.synthetic_lf(n, W, H, 9 + n)
    if kind != "random":
        w = np.zeros((V, n), np.float32)
        if kind == "ties_everywhere":
            for v in range(V):
                a_, b_ = rng.choice(n, 2, replace=False)
                w[v, a_] = w[v, b_] = 0.5
            lf[..., :3] = (lf[..., :3] // 2) * 2
            lf[::2, :, :, :3] += 1
        elif kind == "sum_1p999":
            w = np.abs(rng.standard_normal((V, n))).astype(np.float32)
            w *= 1.99 / w.sum(1, keepdims=True)
        else:
            w = (rng.random((V, n)) * 3e-5).astype(np.float32)
            w[:, n // 2] = 0.75
        hp.weights = _f16_bits(w)
    lf[..., 3] = 255
    levels = np.repeat(np.repeat(rng.integers(0, 256, size=((H + 1) // 2, (W + 15) // 16)), 2, axis=0), 16, axis=1)[:H, :W]
    levels[:, W // 2:] = rng.integers(0, 256, size=(H, W - W // 2))        # the right half: a noise map
    m = np.repeat(levels[..., None].astype(np.uint8), 4, axis=-1)
    m[..., 3] = 255
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=m, focus=hp.focus, rng=hp.range, threads=8)
    for flags in (0, gpu.LFI_FLAG_STD_ANALYTIC_BAND):
        ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=flags)
        ctx.upload_map(1, m)                   # Standard::process<true> reads map 1 (src/kernels.cu:326)
        ctx.render("STD", all_focus=True)
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_stdxa<STD,allfocus>"
        got = ctx.download_views()
        assert (got == want).all(), (kind, flags, int((got != want).sum()))
        if n > 128:
            ctx.set_variant("STD", "filtered_gather_once")
            ctx.render("STD", all_focus=True)
            ctx.sync()
            assert ctx.last_kernel_name() == "blend_afs<STD,allfocus>" and (ctx.download_views() == want).all(), (kind, flags, "blend_afs")
            ctx.set_variant("STD", "auto")
        if flags == 0:
            v0, v1 = V // 3, V // 3 + min(20, V - V // 3)
            ctx.render("STD", all_focus=True, v0=v0, v1=v1)
            ctx.sync()
            assert (ctx.download_views(v0, v1) == want[v0:v1]).all(), (kind, "view range")
            ctx.set_variant("STD", "wave_m2_nt")          # the exact-fp32 kernel agrees
            ctx.render("STD", all_focus=True)
            ctx.sync()
            assert ctx.last_kernel_name() == "blend_persist<STD,allfocus>" and (ctx.download_views() == want).all()
            ctx.set_variant("STD", "auto")
            ctx.set_output_layout("planar")               # round 4: blend_stdxa writes the byte planes itself (quad transposes, byte patches into planes)
            ctx.render("STD", all_focus=True)
            ctx.sync()
            assert ctx.last_kernel_name() == "blend_stdxa<STD,allfocus>" and ctx.memory_info().workspace_bytes == 0
            assert (ctx.download_views() == want).all(), (kind, "planar layout")
        ctx.close()
    if H >= 4:
        band = (1, H - 1)
        in_rows = gpu.input_rows_all_focus(band, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, H)
        ctx = gpu.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        ctx.upload_map(1, m)
        ctx.render("STD", all_focus=True)
        ctx.sync()
        assert (ctx.download_views()[:, band[0]:band[1]] == want[:, band[0]:band[1]]).all(), (kind, "row band")
        ctx.close()


def test_std_analytic_band_flag(gpu, oracle_c):
    """blend_planar<STDF> (up to 64 images) sizes its band with the analytic accumulation bound (a whole ulp per addend) by default since
    round 4; LFI_FLAG_STD_MEASURED_BAND selects the measured quarter ulp (rounds 2-3's default), LFI_FLAG_STD_ANALYTIC_BAND still names the
    default explicitly: same kernel, same bytes, another number of sums recomputed — bit-exact every way."""
    cols, rows, W, H, V = 8, 8, 300, 5, 64
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.1, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 21)
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    for flags in (0, gpu.LFI_FLAG_STD_ANALYTIC_BAND, gpu.LFI_FLAG_STD_MEASURED_BAND):
        ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=flags)
        ctx.render("STD")
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_planar<STDF>"
        assert (ctx.download_views() == want).all(), flags
        ctx.close()


def _exact_products(a_bits, b_bits):
    """Exact A·B of fp16 bit patterns as int64 scaled by 2^48: every fp16 is an integer multiple of 2^-24, the A operands are weights × 2^15
    (≤ 65504 → ≤ 2^40 in those units), the B operands pixel bytes as subnormals (≤ 255 units), and a dot product has ≤ 256 terms: < 2^57."""
    a = np.round(a_bits.view(np.float16).astype(np.float64) * 2.0 ** 24).astype(np.int64)
    b = np.round(b_bits.view(np.float16).astype(np.float64) * 2.0 ** 24).astype(np.int64)
    assert a.max() < 2 ** 41 and b.max() < 2 ** 8 and a.shape[1] <= 256
    return a @ b


def _adversarial_weight_row(rng, K, kind):
    """One view's K weights (float64, before the fp16 rounding), in [0, 2) and summing to at most 2 — the kernels' preconditions — built to
    stress an accumulator that aligns addends to the running sum and drops low bits."""
    full = 2047.0 / 2048.0                                   # all 11 mantissa bits set
    w = np.zeros(K)
    if kind == 0:      # one dominant weight, the rest 10 … 24 binades below with full mantissas
        w[:] = full * 2.0 ** (-rng.integers(10, 25, K).astype(np.float64))
        w[rng.integers(0, K)] = full
    elif kind == 1:    # a convex combination with 11 significant bits each (what generateWeights produces, -s 7)
        w = rng.random(K) ** 7
        w /= w.sum()
    elif kind == 2:    # equal weights 1/K: every addend in one binade, carries ripple through the whole sum
        w[:] = 1.0 / K
    elif kind == 3:    # sums up to 510: the top of the band's validity range
        w = rng.random(K)
        w *= 1.999 / w.sum()
    elif kind == 4:    # a random exponent per addend, uniform over 0 … 24 binades, random mantissas
        w = (1.0 + rng.integers(0, 1024, K) / 1024.0) * 2.0 ** (-1.0 - rng.integers(0, 25, K))
        w *= min(1.0, 1.99 / w.sum())
    elif kind == 5:    # magnitudes ascending: the running sum is always small against the next addend's low bits …
        w = np.sort(full * 2.0 ** (-rng.integers(1, 25, K).astype(np.float64)))
        w *= min(1.0, 1.99 / w.sum())
    elif kind == 6:    # … and descending: every later addend is aligned far down
        w = np.sort(full * 2.0 ** (-rng.integers(1, 25, K).astype(np.float64)))[::-1].copy()
        w *= min(1.0, 1.99 / w.sum())
    elif kind == 7:    # alternating large / tiny
        w[0::2] = full / K
        w[1::2] = full * 2.0 ** -24
    elif kind == 8:    # a geometric decay: every addend one binade below the previous, repeated
        w = full * 2.0 ** (-1.0 - (np.arange(K) % 24))
        w *= min(1.0, 1.99 / w.sum())
    else:              # just below powers of two, random binades near the top
        w = (1.0 - 2.0 ** -11) * 2.0 ** (-rng.integers(1, 8, K).astype(np.float64))
        w *= min(1.0, 1.99 / w.sum())
    return w


def _adversarial_pixel_column(rng, K, kind):
    if kind == 0:
        return np.full(K, 255)
    if kind == 1:
        return rng.integers(0, 256, K)
    if kind == 2:
        col = rng.integers(0, 256, K) | 1      # low bit always set
        col[0] = 255
        return col
    if kind == 3:
        return np.where(np.arange(K) % 2 == 0, 255, 1)
    if kind == 4:
        return 2 ** rng.integers(0, 8, K)      # single bits: products with one-bit pixel factors
    return np.where(rng.random(K) < 0.1, 255, rng.integers(0, 4, K))   # mostly tiny, a few saturated


MFMA_PROBE_SETS = 200


@pytest.mark.parametrize("shape", [0, 1], ids=["32x32x16_chain", "16x16x32_chain"])
def test_mfma_f16_accumulation_error_bound(shape, gpu):
    """The band method for MORE than 64 images (blend_stdx, blend_stdxa, blend_afs) rounds the fp16-MFMA sum wherever it is farther than the
    chain's bound + N·2^-17 + 2^-12 from a half-integer; the N·2^-17 part ASSUMES that the matrix pipe's fp32 accumulation of the exactly
    representable products errs by at most a quarter ulp of a value below 512 per addend.  (Up to 64 images the default band uses the analytic
    N·2^-15 instead — it costs nothing there, tools/std_band_cost.py — so blend_planar<STDF> does not rest on this measurement; with four chunks
    the analytic band doubles the launch time, so there the measured bound stays and this probe is what it rests on.)
    Round 4: MFMA_PROBE_SETS seeded operand sets per chain depth (64 and 256) and MFMA shape, 1,024 sums each, fed exactly as the kernels feed
    them — weights ×2^15 as the A operand, pixel bytes as fp16 subnormals (b·2^-24) as the B operand, acc = S·2^-9 — from ten weight families
    (dominant + tiny, convex combinations, equal, sums near 510, random exponents over 24 binades, ascending / descending magnitudes,
    alternating large / tiny, geometric decays, just below powers of two) × six pixel families.  Exact sums in int64.  Asserts the bound,
    reports the worst case as a fraction of it (rounds 2-3 probed 2 sets: 0.34 of the bound = 0.086 ulp per addend)."""
    ctx = gpu.Context(0)
    worst, worst_at = 0.0, None
    for K in (64, 256):
        bound = K * 2.0 ** -26           # acc = S·2^-9: the kernels' assumption N·2^-17 on S is N·2^-26 on acc
        for seed in range(MFMA_PROBE_SETS):
            rng = np.random.default_rng(1000 * K + seed)
            a = np.zeros((32, K), np.float16)   # weights × 2^15 (exact in fp16 for weights in [0, 2))
            b = np.zeros((K, 32), np.uint16)    # pixel bytes = mantissas of fp16 subnormals
            row_kind = np.arange(32) % 10 if seed == 0 else rng.integers(0, 10, 32)
            col_kind = np.arange(32) % 6 if seed == 0 else rng.integers(0, 6, 32)
            for i in range(32):
                w16 = _adversarial_weight_row(rng, K, int(row_kind[i])).astype(np.float16).astype(np.float64)
                if w16.sum() > 2.0:                      # the fp16 rounding pushed the sum over the precondition: scale down a binade
                    w16 *= 0.5
                a[i] = (w16 * 32768.0).astype(np.float16)
            for j in range(32):
                b[:, j] = _adversarial_pixel_column(rng, K, int(col_kind[j]))
            a_bits = a.view(np.uint16)
            got = ctx.debug_mfma_f16_chain(a_bits, b, shape=shape).astype(np.float64)
            got_i = np.round(got * 2.0 ** 48).astype(np.int64)          # fp32 values below 1: exact multiples of 2^-48 here
            assert (got_i.astype(np.float64) == got * 2.0 ** 48).all()
            err = np.abs(got_i - _exact_products(a_bits, b)).astype(np.float64) / 2.0 ** 48
            e = float(err.max())
            if e / bound > worst:
                i, j = np.unravel_index(int(err.argmax()), err.shape)
                worst, worst_at = e / bound, (K, seed, int(row_kind[i]), int(col_kind[j]))
            assert e <= bound, (K, seed, e, bound)
    print(f"MFMA f16 accumulation (shape {shape}): {2 * MFMA_PROBE_SETS} operand sets, {2 * MFMA_PROBE_SETS * 1024} sums; worst |error| = {worst:.4f} of the assumed bound "
          f"K·2^-26 (= {worst / 4:.4f} ulp(512) per addend) at (K, seed, weight family, pixel family) = {worst_at}")
    ctx.close()


def test_std_band_self_check_on_the_device(gpu, oracle_c):
    """The measured bound behind the band of STD on more than 64 images is checked on the device in use (round 5, lfi_std_band_info):
    the first such launch measures the matrix pipe's accumulation error over adversarial operand sets; the record says the bound holds
    (worst case below the budget) and what the check cost.  LFI_FLAG_STD_BAND_PROBE_FAIL makes a context behave as if its device had
    failed: the launches take the analytic band — and still give the oracle's bytes (fixed focus and all-focus, both band kernels)."""
    cols, rows, W, H, V = 15, 15, 256, 12, 24
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.1, 0.3, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(n, W, H, 5)
    lf[..., :3] = (lf[..., :3] // 2) * 2          # many sums on or next to x.5
    lf[::2, :, :, :3] += 1
    lf[..., 3] = 255
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    map1 = np.zeros((H, W, 4), np.uint8)
    map1[..., 0] = (np.arange(W)[None, :] // 16 * 37 + np.arange(H)[:, None] * 5) % 256
    map1[..., 1] = map1[..., 2] = map1[..., 0]
    map1[..., 3] = 255
    want_af = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=map1, focus=hp.focus, rng=hp.range, threads=8)
    for flags, forced in ((0, False), (gpu.LFI_FLAG_STD_BAND_PROBE_FAIL, True), (gpu.LFI_FLAG_STD_BAND_PROBE_FAIL | gpu.LFI_FLAG_STD_ANALYTIC_BAND, False)):
        ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf, flags=flags)
        ctx.render("STD")
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_stdx<STD>"
        assert (ctx.download_views() == want).all(), flags
        ctx.upload_map(1, map1)
        ctx.render("STD", all_focus=True)
        ctx.sync()
        assert (ctx.download_views() == want_af).all(), (flags, "all-focus")
        info = ctx.std_band_info()
        assert info.probed == 1 and info.within_budget == 1 and info.sums >= 20000 and 0.0 < info.worst_fraction <= 1.0, info.message
        assert info.analytic_forced == int(forced), (flags, info.message)
        assert info.probe_ms < 200.0
        ctx.close()
    print(f"band self-check: {info.message.decode()}; cost {info.probe_ms:.2f} ms once per device")


def test_std_near_half_integer_sums_from_precise_weights(gpu, oracle_c):
    """End-to-end companion of the accumulation-bound test: sums S that land within 2^-12 of x.5 (inside the rounding band, but
    not exact ties) built from weights with ≥ 10 significant bits.  Zero offsets (focus 0), so output pixel x blends pixel x of
    every image: for each pixel a vector of 64 bytes is searched whose exact blend with view 0's weights is that close to a
    half-integer; the other views see the same pixels with other weights.  STD must stay bit-exact against the oracle."""
    cols = rows = 8
    n, W, H, V = 64, 256, 3, 64
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.0, 0.0, 7.0, 1.783, V)
    assert (hp.focused_offsets == 0).all()
    w0 = hp.weights[0].view(np.float16).astype(np.float64)
    assert (np.frexp(w0[w0 > 0])[0] * 2048 % 2 == 1).mean() > 0.2   # plenty of weights use all 11 bits
    rng = np.random.default_rng(77)
    lf = np.zeros((n, H, W, 4), np.uint8)
    lf[..., 3] = 255
    found = 0
    for y in range(H):
        for c in range(3):
            need = np.ones(W, bool)
            while need.any():
                cand = rng.integers(0, 256, (4096, n))
                s = cand @ w0
                close = np.abs(s - np.floor(s) - 0.5) < 2.0 ** -12
                for row in cand[close]:
                    idx = np.flatnonzero(need)
                    if not len(idx):
                        break
                    lf[:, y, idx[0], c] = row
                    need[idx[0]] = False
                    found += 1
    assert found == W * H * 3
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    for variant in ("auto", "wave_m2_nt"):
        ctx.set_variant("STD", variant)
        ctx.render("STD")
        ctx.sync()
        got = ctx.download_views()
        assert (got == want).all(), (variant, int((got != want).sum()))
    ctx.set_variant("STD", "auto")
    ctx.render("STD")
    assert ctx.last_kernel_name() == "blend_planar<STDF>"
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_row_band_sharding_all_focus(world, gpu, oracle_c):
    """SURVEY.md §8(f).2 for the all-focus path (BASELINE config 5 sharded by rows): every rank computes ITS BAND of the focus maps
    from the input rows it holds (band + the warp's reach over the focus range + the block radius) and renders the band
    all-focused; maps and views reassemble to the single-GPU result byte for byte.  Too few input rows are refused."""
    cols = rows = 8
    W, H, V = 160, 120, 8
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.05, 0.12, 7.0, 1.783, V)
    lf = oracle_c.synthetic_lf(64, W, H, SEED)
    full = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    full.focus_map()
    full.sync()
    want_maps = (full.download_map(0), full.download_map(1))
    want = {}
    for method in ("STD", "TEN_WM"):
        full.render(method, all_focus=True)
        full.sync()
        want[method] = full.download_views()
    full.close()
    got = {m: np.zeros_like(want[m]) for m in want}
    got_map1 = np.zeros_like(want_maps[1])
    held = []
    for rank in range(world):
        band = gpu.row_band(H, world, rank)
        in_rows = gpu.input_rows_all_focus(band, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, H)
        held.append(in_rows[1] - in_rows[0])
        ctx = gpu.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        ctx.focus_map()
        ctx.sync()
        m0, m1 = ctx.download_map(0), ctx.download_map(1)
        assert (m0[band[0]:band[1]] == want_maps[0][band[0]:band[1]]).all()
        assert (m1[band[0]:band[1]] == want_maps[1][band[0]:band[1]]).all()
        got_map1[band[0]:band[1]] = m1[band[0]:band[1]]
        for method in ("STD", "TEN_WM"):
            ctx.render(method, all_focus=True)
            ctx.sync()
            part = ctx.download_views()
            got[method] |= part
            # the band into the planar view layout (round 4: the all-focus kernels write the byte planes themselves): the same bytes
            ctx.set_output_layout("planar")
            ctx.render(method, all_focus=True)
            ctx.sync()
            assert (ctx.download_views() == part).all(), (method, "planar views of the band", rank)
            ctx.set_output_layout("rgba")
        ctx.close()
    assert (got_map1 == want_maps[1]).all()
    # against the oracle: both maps of the bands, STD bit-exact from map 1, TEN_WM within one LSB from map 0 (src/kernels.cu:326, :430)
    o_map0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
    o_map1 = oracle_c.focus_filter(o_map0, hp.block_radius)
    assert (want_maps[0] == o_map0).all() and (got_map1 == o_map1).all()
    o_std = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=o_map1, focus=hp.focus, rng=hp.range, threads=8)
    o_ten = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=o_map0, focus=hp.focus, rng=hp.range, threads=8)
    assert (got["STD"] == o_std).all(), "all-focus row bands: STD differs from the oracle"
    assert np.abs(got["TEN_WM"].astype(int) - o_ten.astype(int)).max() <= TEN_TOL_LSB, "all-focus row bands: TEN_WM differs from M16"
    for method in want:
        assert (got[method] == want[method]).all(), method
    assert max(held) < H or world == 2
    # a window that covers the fixed-focus reach only is refused by the all-focus paths
    band = gpu.row_band(H, 3, 1)
    small = gpu.input_rows(band, hp.focused_offsets, H)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.set_row_window(band[0], band[1], small[0], small[1])
    ctx.fill_synthetic(SEED)
    ctx.set_params(hp)
    with pytest.raises(gpu.LfiError, match="does not cover"):
        ctx.focus_map()
    with pytest.raises(gpu.LfiError, match="does not cover"):
        ctx.render("TEN_WM", all_focus=True)
    ctx.close()
