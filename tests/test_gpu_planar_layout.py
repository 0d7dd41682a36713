"""GPU (-m gpu): the opt-in alpha-free view layout (lfi_set_output_layout, blend_p3.hpp).

The reference's kernels write alpha = 255 into every output pixel (uchar4{…, 255}, src/kernels.cu:393, :309).  In the planar
layout the device stores byte planes [view][R,G,B][rows][pitch] and downloads re-create the alpha, so everything a caller
downloads must be byte-identical to the RGBA layout: the TEN_WM parity contract (≤ 1 LSB vs the oracle's M16 model) is
checked again in this layout, and every other render (STD, all-focus, view ranges, row windows, quilts) must give the bytes
of the RGBA layout."""
import numpy as np
import pytest

from conftest import GOLDEN_DIR, SEED, SMALL_CASES
import os

pytestmark = pytest.mark.gpu

TEN_TOL_LSB = 1


def _ctx(gpu, cols, rows, W, H, hp, lf=None, seed=SEED, flags=0, layout="planar"):
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    if lf is None:
        ctx.fill_synthetic(seed)
    else:
        ctx.upload_grid(lf)
    ctx.set_params(hp, flags)
    ctx.set_output_layout(layout)
    return ctx


@pytest.mark.parametrize("case", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_golden_fixtures_planar_layout(case, gpu):
    name, cols, rows, W, H, V, traj, focus, aspect, effect = case
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    rng = float(g["range"])
    hp = gpu.build_params(cols, rows, W, H, traj, focus, rng, effect, aspect, V)
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=g["lf"])
    vl = ctx.view_layout()
    assert vl.layout == gpu.LFI_LAYOUT_PLANAR_RGB and vl.row_pitch_bytes % 16 == 0 and vl.row_pitch_bytes >= W
    assert vl.view_stride_bytes == 3 * H * vl.row_pitch_bytes and ctx.views_device_ptr()[1] == V * vl.view_stride_bytes
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name() == "blend_p3<TEN_WM>"
    out = ctx.download_views()
    assert np.abs(out.astype(int) - g["ten_m16"].astype(int)).max() <= TEN_TOL_LSB
    assert (out[..., 3] == 255).all()
    assert (out != g["ten_exact"]).mean() < 1e-3
    # STD: blend_stdx writes the byte planes itself (round 4); all-focus renders go through the RGBA kernels and a conversion: same bytes as ever
    ctx.render("STD")
    ctx.sync()
    assert ctx.last_kernel_name() == "blend_stdx<STD>"
    assert (ctx.download_views() == g["std"]).all()
    ctx.focus_map()
    ctx.render("STD", all_focus=True)
    ctx.sync()
    assert (ctx.download_views() == g["af_std"]).all()
    ws_before = ctx.memory_info().workspace_bytes
    ctx.render("TEN_WM", all_focus=True)
    ctx.sync()
    # round 4: blend_persist writes the byte planes itself (quad transposes in its epilogue): no growth of the RGBA scratch copy
    assert ctx.last_kernel_name() == "blend_persist<TEN_WM,allfocus>" and ctx.memory_info().workspace_bytes == ws_before
    assert np.abs(ctx.download_views().astype(int) - g["af_ten_m16_map0"].astype(int)).max() <= TEN_TOL_LSB
    ctx.close()


@pytest.mark.parametrize("shape", [(8, 8, 130, 9, 64), (15, 15, 70, 6, 45), (3, 3, 256, 256, 1), (2, 5, 31, 33, 70), (1, 1, 17, 5, 3),
                                   (8, 8, 512, 4, 130), (8, 8, 1000, 3, 17), (12, 12, 257, 5, 64), (10, 10, 129, 7, 33), (8, 8, 128, 16, 64)],
                         ids=lambda s: "x".join(map(str, s)))
def test_ragged_shapes_planar_equals_rgba(shape, gpu, oracle_c):
    """Widths that are not multiples of the 128-pixel tile / of 8 / of 16, 1–4 chunks of 64 images (N not a multiple of 16),
    view counts that leave waves idle or need several launches: the planar layout's TEN_WM bytes are the RGBA layout's, and
    within the contract of the oracle."""
    cols, rows, W, H, V = shape
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.3, 0.0, 3.0, 1.5, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8)
    rgba = _ctx(gpu, cols, rows, W, H, hp, layout="rgba")
    rgba.render("TEN_WM")
    rgba.sync()
    want = rgba.download_views()
    rgba.render("STD")
    rgba.sync()
    want_std = rgba.download_views()
    rgba.close()
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name() == "blend_p3<TEN_WM>"
    got = ctx.download_views()
    assert np.abs(got.astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB
    assert (got == want).all(), int((got != want).sum())
    # consecutive launches walk the image in opposite directions (Infinity Cache reuse): same bytes, with and without the alternation
    for flags in (0, gpu.LFI_FLAG_SINGLE_SWEEP_DIRECTION):
        ctx.set_params(hp, flags)
        for _ in range(3):
            ctx.render("TEN_WM")
            ctx.sync()
            assert (ctx.download_views() == want).all()
    ctx.set_params(hp)
    # a second launch over a sub-range leaves the other views alone and reproduces its own
    v0, v1 = V // 3, max(V // 3 + 1, (2 * V) // 3)
    ctx.render("TEN_WM", v0=v0, v1=v1)
    ctx.sync()
    assert (ctx.download_views() == want).all()
    ctx.render("STD")
    ctx.sync()
    assert (ctx.download_views() == want_std).all()
    ctx.close()


def test_planar_layout_offsets_larger_than_image(gpu, oracle_c):
    cols = rows = 15
    W, H, V = 48, 20, 8
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 3.0, 0.0, 7.0, 2.0223, V)
    lf = oracle_c.synthetic_lf(225, W, H, 5)
    ctx = _ctx(gpu, cols, rows, W, H, hp, seed=5)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights)
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name() == "blend_p3<TEN_WM>"
    assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB
    ctx.close()


def test_planar_layout_weights_outside_unit_range_fall_back(gpu, oracle_c):
    """Weights outside [0, 2) cannot take the packed epilogue: the generic RGBA kernel + conversion serves them."""
    cols, rows, W, H, V = 4, 4, 40, 6, 5
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.0, V)
    w = hp.weights.view(np.float16).copy()
    w[0, 0] = np.float16(2.5)
    w[1, 3] = np.float16(-0.25)
    hp.weights = w.view(np.uint16)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name() != "blend_p3<TEN_WM>"
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights)
    assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB
    ctx.close()


@pytest.mark.parametrize("world,cols,W", [(2, 8, 200), (3, 8, 203), (2, 15, 200), (3, 11, 131)])
def test_planar_layout_row_bands(world, cols, W, gpu, oracle_c):
    """Row-band sharding in the planar layout: every band renders its rows of every view from the input rows it holds — with one chunk
    of images (four waves of 16 views per workgroup) and with several (two waves of 32 views); ragged widths (203, 131: not a
    multiple of the 16-byte view pitch); the first and last bands' halos clamp at the image edges (offsets reach beyond them).
    Checked against the ORACLE (≤ 1 LSB of M16) and, byte for byte, against the library's unsharded RGBA render."""
    rows = cols
    H, V = 96, 64 if cols == 8 else 40
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.1, 0.0, 3.0, 1.783, V)
    assert hp.focused_offsets[:, 1].min() < 0 < hp.focused_offsets[:, 1].max()    # rows above row 0 / below row H − 1 are sampled
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, threads=8)
    full = _ctx(gpu, cols, rows, W, H, hp, layout="rgba")
    full.render("TEN_WM")
    full.sync()
    want = full.download_views()
    full.render("STD")
    full.sync()
    want_std = full.download_views()
    full.close()
    got = np.zeros_like(want)
    got_std = np.zeros_like(want)
    for rank in range(world):
        band = gpu.row_band(H, world, rank)
        in_rows = gpu.input_rows(band, hp.focused_offsets, H)
        ctx = gpu.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
        ctx.fill_synthetic(SEED)
        ctx.set_params(hp)
        ctx.set_output_layout("planar")
        assert ctx.view_layout().rows == band[1] - band[0]
        ctx.render("TEN_WM")
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_p3<TEN_WM>"
        part = ctx.download_views()
        assert (part[:, :band[0]] == 0).all() and (part[:, band[1]:] == 0).all()
        got |= part
        ctx.render("STD")           # round 4: blend_stdx writes the band's byte planes itself (one chunk of images or several)
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_stdx<STD>"
        got_std |= ctx.download_views()
        ctx.close()
    assert (got_std == want_std).all(), "planar row bands of STD differ from the unsharded RGBA render"
    assert np.abs(got.astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, "planar row bands differ from the oracle's M16"
    assert (got[..., 3] == 255).all()
    assert (got == want).all()


@pytest.mark.parametrize("V,W,H", [(129, 200, 9), (200, 131, 7), (256, 384, 5), (333, 140, 4)])
def test_planar_layout_three_and_more_view_passes(V, W, H, gpu, oracle_c):
    """256 views from 64 images (BASELINE config 4 on one GPU): three or more 64-view passes per LDS-resident tile in blend_p3 (more than
    four: a second launch).  Within
    one LSB of the oracle's M16, identical to the RGBA layout's kernel, both sweep directions, a view range that starts inside a pass,
    and a row band."""
    cols = rows = 8
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.15, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, threads=8)
    ctx = _ctx(gpu, cols, rows, W, H, hp, lf=lf)
    for sweep in range(2):
        ctx.render("TEN_WM")
        ctx.sync()
        assert ctx.last_kernel_name() == "blend_p3<TEN_WM>"
        got = ctx.download_views()
        assert np.abs(got.astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, sweep
    ctx.render("TEN_WM", v0=5, v1=V - 3)            # 3 or 4 passes from an odd first view
    ctx.sync()
    assert (ctx.download_views(5, V - 3) == got[5:V - 3]).all()
    ctx.set_output_layout("rgba")
    ctx.render("TEN_WM")
    ctx.sync()
    assert (ctx.download_views() == got).all()
    ctx.close()
    if H >= 7:
        band = (2, H - 2)
        in_rows = gpu.input_rows(band, hp.focused_offsets, H)
        ctx = gpu.Context(0)
        ctx.set_grid(cols, rows, W, H)
        ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
        ctx.upload_grid(lf)
        ctx.set_params(hp)
        ctx.set_output_layout("planar")
        ctx.render("TEN_WM")
        ctx.sync()
        assert (ctx.download_views()[:, band[0]:band[1]] == got[:, band[0]:band[1]]).all()
        ctx.close()


def test_contexts_after_planar_view_contexts_render_correctly(gpu, oracle_c):
    """Regression (round 3): the library's planar views live in uncached device memory.  Handed back to the HIP runtime with hipFree,
    those address ranges were recycled for ordinary allocations, and contexts created after a few such cycles rendered garbage (inputs,
    parameters or RGBA views on a range that had been uncached).  Released blocks now stay in a process-wide free list.  The sequence
    that exposed it: a few contexts that switch to the planar layout and are closed, then fresh contexts of other shapes."""
    for cols, W in [(8, 203), (11, 131), (8, 200)]:
        rows = cols
        H, V = 96, 64 if cols == 8 else 40
        hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.1, 0.0, 3.0, 1.783, V)
        lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
        m16 = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, threads=8)
        for rep in range(3):
            ctx = _ctx(gpu, cols, rows, W, H, hp, layout="rgba")
            for variant in ("persist_m2_nt", "auto"):
                ctx.set_variant("TEN_WM", variant)
                ctx.render("TEN_WM")
                ctx.sync()
                assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, (cols, W, rep, variant)
            ctx.set_output_layout("planar")
            ctx.render("TEN_WM")
            ctx.sync()
            assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= TEN_TOL_LSB, (cols, W, rep, "planar")
            ctx.close()


def test_planar_layout_quilt_and_attached_views(gpu):
    import torch
    cols = rows = 3
    W, H, V = 70, 11, 6
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.0, V)
    ctx = _ctx(gpu, cols, rows, W, H, hp)
    vl = ctx.view_layout()
    buf = torch.zeros(V * vl.view_stride_bytes, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(gpu.LfiError):
        ctx.attach_views(buf.data_ptr(), buf.numel() - 1)
    ctx.attach_views(buf.data_ptr(), buf.numel())
    ctx.render("TEN_WM")
    ctx.sync()
    views = ctx.download_views()
    quilt = ctx.download_quilt(3, 2)
    for v in range(V):
        ty, tx = divmod(v, 3)
        assert (quilt[ty * H:(ty + 1) * H, tx * W:(tx + 1) * W] == views[v]).all()
    # the attached buffer holds the byte planes: [view][channel][row][pitch]
    planes = buf.cpu().numpy().reshape(V, 3, H, vl.row_pitch_bytes)[..., :W]
    assert (planes.transpose(0, 2, 3, 1) == views[..., :3]).all()
    # back to the reference's layout: views are reallocated, renders write RGBA again
    ctx.set_output_layout("rgba")
    ctx.render("TEN_WM")
    ctx.sync()
    assert ctx.last_kernel_name().startswith("blend_planar") and (ctx.download_views() == views).all()
    ctx.close()
