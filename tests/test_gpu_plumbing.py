"""GPU (-m gpu): the plumbing entry points of the C-ABI — caller-owned device memory, streams, timers, the benchmark loop,
two contexts side by side — and bench.py's JSON contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SEED

pytestmark = pytest.mark.gpu


def test_attached_buffers_and_caller_stream(gpu, oracle_c):
    torch = pytest.importorskip("torch")
    cols = rows = 4
    W, H, V = 96, 16, 8
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.5, V)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    dev = torch.device("cuda", 0)
    grid = torch.from_numpy(lf).to(dev)
    views = torch.zeros((V, H, W, 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.attach_grid(grid.data_ptr(), grid.numel())
    ctx.set_params(hp)
    ctx.attach_views(views.data_ptr(), views.numel())
    assert ctx.grid_device_ptr() == (grid.data_ptr(), grid.numel())
    assert ctx.views_device_ptr() == (views.data_ptr(), views.numel())
    ctx.set_stream(stream.cuda_stream)
    ctx.render("STD")
    stream.synchronize()
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    assert (views.cpu().numpy() == want).all()          # the kernel wrote into the caller's tensor on the caller's stream
    ctx.set_stream(None)
    with pytest.raises(gpu.LfiError, match="smaller than"):
        ctx.attach_views(views.data_ptr(), 16)
    ctx.close()
    assert (views.cpu().numpy() == want).all()          # attached memory survives the context


def test_timer_and_benchmark_stats(gpu):
    ctx = gpu.Context(0)
    ctx.set_grid(4, 4, 256, 64)
    ctx.fill_synthetic(1)
    ctx.set_params(gpu.build_params(4, 4, 256, 64, "0,0,1,1", 0.1, 0.0, 3.0, 1.0, 16))
    st = ctx.benchmark("TEN_WM", warmup=1, runs=5)
    assert st.runs == 5 and 0 < st.min_ms <= st.median_ms <= st.max_ms and st.mean_ms > 0 and st.back_to_back_ms > 0
    ctx.timer_start()
    for _ in range(3):
        ctx.render("STD")
    assert ctx.timer_stop() > 0
    ctx.close()


def test_two_contexts_are_independent(gpu, oracle_c):
    a, b = gpu.Context(0), gpu.Context(0)
    a.set_grid(3, 3, 40, 10)
    b.set_grid(2, 2, 24, 6)
    a.fill_synthetic(5)
    b.fill_synthetic(6)
    ha = gpu.build_params(3, 3, 40, 10, "0,0,1,1", 0.3, 0.0, 3.0, 1.0, 4)
    hb = gpu.build_params(2, 2, 24, 6, "1,0,0,1", 0.1, 0.0, 2.0, 1.0, 3)
    a.set_params(ha)
    b.set_params(hb)
    a.render("TEN_WM")
    b.render("STD")
    a.sync()
    b.sync()
    assert (b.download_views() == oracle_c.blend_std(oracle_c.synthetic_lf(4, 24, 6, 6), hb.focused_offsets, hb.offsets, hb.weights)).all()
    ten = oracle_c.blend_ten(oracle_c.synthetic_lf(9, 40, 10, 5), ha.focused_offsets, ha.offsets, ha.weights)
    assert np.abs(a.download_views().astype(int) - ten.astype(int)).max() <= 1
    with pytest.raises(gpu.LfiError):
        a.render("STD", v0=0, v1=99)
    b.render("STD")  # an error on one context leaves the other usable
    b.sync()
    a.close()
    b.close()


def test_single_process_broadcast(gpu):
    """lfi_broadcast_grid: a no-op for one context; contexts on one device are refused before RCCL is touched
    (two distinct GPUs are needed for a real broadcast, which only the multi-GPU node has)."""
    from lfinterpolator_amd.abi import broadcast_grid
    a, b = gpu.Context(0), gpu.Context(0)
    for c in (a, b):
        c.set_grid(2, 2, 32, 8)
    a.fill_synthetic(3)
    broadcast_grid([a])
    with pytest.raises(gpu.LfiError, match="distinct devices"):
        broadcast_grid([a, b], root=0)
    b.set_grid(2, 2, 32, 16)
    with pytest.raises(gpu.LfiError, match="same grid"):
        broadcast_grid([a, b], root=0)
    a.close()
    b.close()


def test_bench_json_contract(gpu):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--prewarm-ms", "0", "--also-iters", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "views/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.05 < r["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "views/s" and "sample" in c
    assert d["value"] > 100 * c["value"]
    # `frac` is on the bytes the layouts in use move (3 B per pixel on both sides by default); the §8(d) figure is the secondary field
    assert r["frac"] <= r["frac_algorithmic"] + 1e-9 and r["bytes_per_launch"] <= r["algorithmic_bytes_per_launch"]
    assert abs(r["bytes_per_launch"] - 3.0 * 1920 * 1080 * 128) < 1 and "frac_of_measured_copy_6290" not in r
    assert r["single_sweep_direction_ms"] > 0 and r["rgba_views_ms"] > 0
    assert d["config"]["derived_copy_bytes"] > 0 and d["config"]["kernel"] and d["config"]["view_ranges"] == [[0, 64]]
    c1 = d["cpu_baseline_1thread"]
    assert c1["cores"] == 1 and c1["kind"] == "port" and 0 < c1["value"] <= c["value"]
    # the other BASELINE configurations, timed in the same run: verbose in also_detail, compact — and LAST on the line — in also
    detail, also = d["also_detail"], d["also"]
    assert list(d.keys())[-1] == "also"
    # round 5 hygiene: the box's core count beside the threads used; how the timed TEN_WM launches compare with M16; the fixed-focus sweeps
    assert c["host_cores"] >= c["cores"] and c1["host_cores"] == c["host_cores"]
    m16 = c["ten_wm_vs_m16"]
    assert m16["max_abs_diff_lsb"] <= 1 and 0.9 < m16["exact_match_fraction"] <= 1.0 and m16["bytes_compared"] >= 3 * 16 * 1920 * 3
    assert d["config"]["ten_wm_exact_match_vs_m16"] == m16["exact_match_fraction"] and "1 LSB" in d["config"]["ten_wm_tolerance"]
    for key in ("config2_fixed_focus_sweep_step", "config2_fixed_focus_sweep_step_std", "config5_fixed_focus_sweep_step", "config5_fixed_focus_sweep_step_std"):
        assert detail[key]["ms"] > 0 and 0 < detail[key]["same_parameters_ms"] < 1.5 * detail[key]["ms"] and "lfi_set_params" in detail[key]["kernel"], key
    keys = ("config2_std", "config2_std_valu", "config3", "config3_std", "config4_rank", "config4_whole_1gpu", "config5_fixed_focus",
            "config5_fixed_focus_std", "config5_fixed_focus_std_nontensor", "config5_focus_map", "config5_allfocus_ten_wm_end_to_end",
            "config5_allfocus_std_end_to_end", "config5_allfocus_std_nontensor", "config2_cold_one_shot", "config5_cold_one_shot")
    for key in keys:
        assert key in detail and detail[key]["ms"] > 0 and 0 < detail[key]["frac"] < 1.2 and detail[key]["kernel"], key
        assert also[key][0] == round(detail[key]["ms"], 4), key
    assert detail["config5_fixed_focus_std_nontensor"]["kernel"] == "blend_std_vfma" and detail["config5_allfocus_std_nontensor"]["kernel"] == "blend_std_vfma"
    assert len(json.dumps(also)) < 3000          # fits the tail a log reader keeps (the driver keeps 8,000 characters)


def test_pinned_host_buffers(gpu, oracle_c):
    """Uploads from / downloads into page-locked arrays (lfi_alloc_pinned) give the same bytes as pageable ones."""
    cols, rows, W, H = 4, 4, 96, 40
    hp = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.1, 0.0, 2.0, 1.5, 6)
    lf = oracle_c.synthetic_lf(cols * rows, W, H, 5)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    pin_in = ctx.pinned_empty(lf.shape)
    pin_in[...] = lf
    ctx.upload_grid(pin_in)
    ctx.set_params(hp)
    ctx.render("STD")
    ctx.sync()
    want = ctx.download_views()
    pin_out = ctx.pinned_empty(want.shape)
    got = ctx.download_views(out=pin_out)
    assert got is pin_out and (got == want).all()
    assert (want == oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)).all()
    ctx.close()



def test_async_uploads_match_synchronous_ones(gpu, oracle_c):
    """lfi_upload_image_async: pageable sources have copy semantics (one buffer reused by the caller right after every call),
    page-locked sources are DMA'd in place; renders and a re-upload in the middle of a sequence of launches are ordered after the
    copies without a host wait.  Bytes identical to synchronous uploads."""
    cols, rows, W, H, V = 4, 4, 96, 40, 8
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.set_params(hp)
    scratch = np.empty((H, W, 4), np.uint8)
    for g in range(16):                       # one pageable buffer reused for every image: the call must have copy semantics
        scratch[...] = lf[g]
        ctx.upload_image_async(g, scratch)
        scratch[...] = 0
    ctx.render("STD")                          # no explicit wait: the launch is ordered after the copies
    ctx.sync()
    assert (ctx.download_views() == want).all()
    pinned = ctx.pinned_empty((16, H, W, 4))
    pinned[...] = lf[::-1]                     # a different grid: images reversed
    for g in range(16):
        ctx.upload_image_async(g, pinned[g])
    ctx.render("TEN_WM")
    ctx.upload_wait()
    ctx.sync()
    got = ctx.download_views()
    want_rev = oracle_c.blend_ten(lf[::-1].copy(), hp.focused_offsets, hp.offsets, hp.weights)
    assert np.abs(got.astype(int) - want_rev.astype(int)).max() <= 1
    ctx.close()


def test_input_changes_reach_the_renders(gpu, oracle_c):
    """TEN_WM renders read a derived (planar) copy of the inputs: every way of changing the inputs must reach them.
    Uploads through the ABI invalidate the copy; an attached buffer is read directly until lfi_grid_modified has been called,
    and after that call the caller's announcements are honoured."""
    torch = pytest.importorskip("torch")
    cols, rows, W, H = 4, 4, 160, 12
    hp = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.2, 0.0, 2.0, 1.5, 8)
    lf_a = oracle_c.synthetic_lf(cols * rows, W, H, 1)
    lf_b = oracle_c.synthetic_lf(cols * rows, W, H, 2)

    def want(lf):
        return oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16)

    def close(out, lf):
        return np.abs(out.astype(int) - want(lf).astype(int)).max() <= 1

    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf_a)
    ctx.set_params(hp)
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), lf_a)
    ctx.upload_image(3, lf_b[3])                       # one image replaced through the ABI
    mixed = lf_a.copy(); mixed[3] = lf_b[3]
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), mixed)
    ctx.fill_synthetic(0x77)                           # all of them regenerated on the device
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), oracle_c.synthetic_lf(cols * rows, W, H, 0x77))
    # caller-owned planes: no announcement yet → every launch reads them as they are
    t = torch.from_numpy(lf_a).cuda()
    ctx.attach_grid(t.data_ptr(), t.numel())
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), lf_a)
    t.copy_(torch.from_numpy(lf_b)); torch.cuda.synchronize()
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), lf_b)
    # from the first announcement on, the copy is used and refreshed on every further announcement
    ctx.grid_modified()
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), lf_b)
    t.copy_(torch.from_numpy(lf_a)); torch.cuda.synchronize()
    ctx.grid_modified()
    ctx.render("TEN_WM"); ctx.sync()
    assert close(ctx.download_views(), lf_a)
    # a new parameter set with larger offsets rebuilds the copy with more padding
    hp2 = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.9, 0.0, 2.0, 1.5, 8)
    ctx.set_params(hp2)
    ctx.render("TEN_WM"); ctx.sync()
    w2 = oracle_c.blend_ten(lf_a, hp2.focused_offsets, hp2.offsets, hp2.weights, model=oracle_c.TEN_M16)
    assert np.abs(ctx.download_views().astype(int) - w2.astype(int)).max() <= 1
    ctx.close()


def test_stream_switch_orders_derived_state(gpu):
    """The context keeps derived device state (the planar copy of the inputs, built asynchronously on whatever stream is
    current; the synthetic fill itself) that launches on ANOTHER stream depend on.  lfi_set_stream orders the old stream's work
    before the new stream's (an event, no host sync): a render that switches streams right after a large fill + planar build
    must produce the bytes of a fully synchronised render."""
    import torch
    cols = rows = 8
    W, H, V = 1920, 540, 8          # large enough that fill (265 MB) + planar_build are still running when the next launch is enqueued
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, V)
    ref = gpu.Context(0)
    ref.set_grid(cols, rows, W, H)
    ref.fill_synthetic(31)
    ref.set_params(hp)
    ref.render("TEN_WM")
    ref.sync()
    want = ref.download_views()
    ref.close()
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.set_params(hp)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for seed in (7, 31):            # the second round also exercises the rebuild of the planar copy after a change of the inputs
        ctx.set_stream(s1.cuda_stream)
        ctx.fill_synthetic(seed)    # asynchronous on s1
        ctx.render("TEN_WM")        # builds the planar copy on s1, renders on s1
        ctx.fill_synthetic(seed)    # inputs "change" again: the copy is stale, the next render rebuilds it …
        ctx.set_stream(s2.cuda_stream)
        ctx.render("TEN_WM")        # … on s2, which must wait for the fill still running on s1
        s2.synchronize()
    ctx.set_stream(None)
    assert (ctx.download_views() == want).all()
    ctx.close()


def test_render_stream_matches_block_by_block_renders(gpu, oracle_c):
    """lfi_render_stream: a 37-view camera path rendered in blocks of 8 views with no host synchronisation in between (weights of
    block b+1 staged and copied while block b renders, views of block b downloaded from a second buffer while block b+1
    renders) gives, view for view, what separate set_params + render + download calls give — the last, partial block included."""
    cols, rows, W, H, V, total = 4, 4, 200, 30, 8, 37
    hp_all = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.5, total)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf)
    for method in ("TEN_WM", "STD"):
        want = np.zeros((total, H, W, 4), np.uint8)
        for b in range(0, total, V):
            hp = hp_all.rows(b, min(b + V, total))
            ctx.set_params(hp)
            ctx.render(method)
            ctx.sync()
            want[b:b + hp.weights.shape[0]] = ctx.download_views()
        ctx.set_params(hp_all.rows(0, V))                    # the block size of the stream = the views of the parameters
        out = ctx.pinned_empty((total, H, W, 4))
        out[...] = 0
        ctx.render_stream(method, hp_all.weights, out)
        assert (out == want).all(), method
        if method == "STD":
            assert (want == oracle_c.blend_std(lf, hp_all.focused_offsets, hp_all.offsets, hp_all.weights)).all()
        ctx.render_stream(method, hp_all.weights, None)     # render only: the last block's views stay on the device
        assert (ctx.download_views(0, total - (total // V) * V) == want[(total // V) * V:]).all()
    ctx.close()


def _ssim_psnr_numpy(a, b):
    """The definitions of csrc/hip/quality.hpp restated: per channel MSE over all pixels; SSIM = mean over 8×8 windows at stride 4."""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    mse = ((a - b) ** 2).mean(axis=(0, 1))
    H, W = a.shape[:2]
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    ssim = np.zeros(3)
    count = 0
    for y in range(0, H - 7, 4):
        for x in range(0, W - 7, 4):
            wa, wb = a[y:y + 8, x:x + 8].reshape(64, 3), b[y:y + 8, x:x + 8].reshape(64, 3)
            mu1, mu2 = wa.mean(0), wb.mean(0)
            var1, var2 = (wa * wa).mean(0) - mu1 * mu1, (wb * wb).mean(0) - mu2 * mu2
            cov = (wa * wb).mean(0) - mu1 * mu2
            ssim += ((2 * mu1 * mu2 + C1) * (2 * cov + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (var1 + var2 + C2))
            count += 1
    return mse, ssim / max(count, 1)


def test_compare_view_psnr_ssim(gpu, oracle_c):
    """lfi_compare_view (replaces scripts/imageQualityMetrics.sh): PSNR / SSIM of a rendered view against a host image, reduced on
    the device, against the same definitions in numpy — STD vs TEN_WM renders of one view, an unrelated image, and the view itself."""
    cols, rows, W, H, V = 4, 4, 150, 61, 4
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.0, 3.0, 1.5, V)
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    lf[..., :3] = (lf[..., :3].astype(np.int32) // 3 + np.arange(W)[None, None, :, None] // 2).clip(0, 255).astype(np.uint8)  # some structure
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf)
    ctx.set_params(hp)
    ctx.render("STD")
    ctx.sync()
    std = ctx.download_views()
    for layout in ("rgba", "planar"):
        ctx.set_output_layout(layout)
        ctx.render("TEN_WM")
        ctx.sync()
        ten = ctx.download_view(1)
        for ref in (std[1], lf[3], ten):
            q = ctx.compare_view(1, ref)
            mse, ssim = _ssim_psnr_numpy(ten, ref)
            assert np.allclose(list(q.mse), mse, rtol=1e-12, atol=0)
            assert np.allclose(list(q.ssim), ssim, rtol=1e-9)
            assert abs(q.ssim_all - ssim.mean()) < 1e-9
            if mse.max() == 0:
                assert q.psnr_all == float("inf") and abs(q.ssim_all - 1.0) < 1e-12
            else:
                assert abs(q.psnr_all - 10 * np.log10(255.0 ** 2 / mse.mean())) < 1e-9
    ctx.close()


def test_prepare_memory_info_and_partial_fills(gpu, oracle_c):
    """lfi_prepare builds (and times) the derived planar copy ahead of the first render; lfi_memory_info reports what the context
    holds; lfi_fill_synthetic_images fills a slice of the grid (the all-gather distribution's per-rank share); the structured scene
    fill runs and changes the inputs."""
    cols, rows, W, H, V = 4, 4, 256, 64, 8
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    hp = gpu.build_params(cols, rows, W, H, "0,0,1,1", 0.2, 0.1, 3.0, 1.5, V)
    ctx.set_params(hp)
    ctx.fill_synthetic(SEED, 0, 5)
    ctx.fill_synthetic(SEED, 5, 16)
    mi = ctx.memory_info()
    assert mi.grid_bytes == 16 * W * H * 4 and mi.derived_bytes == 0 and mi.views_bytes == V * W * H * 4
    ctx.prepare("TEN_WM")
    mi = ctx.memory_info()
    # ONE padded byte plane per image and channel (round 3; rounds 1–2 kept four byte-shifted copies: 12 B per pixel·image)
    assert 3 * 16 * W * H <= mi.derived_bytes <= 3 * 16 * (W + 2 * 300) * H and mi.derived_build_ms > 0
    assert mi.derived_bytes < mi.grid_bytes * 3
    ctx.render("STD")
    ctx.sync()
    lf = oracle_c.synthetic_lf(16, W, H, SEED)
    assert (ctx.download_views() == oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)).all()
    with pytest.raises(gpu.LfiError):
        ctx.fill_synthetic(SEED, 3, 17)
    ctx.fill_synthetic_scene(SEED)
    ctx.render("STD")
    ctx.sync()
    scene = ctx.download_views()
    assert (scene[..., 3] == 255).all() and (scene != oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)).any()
    ctx.focus_map()
    ctx.sync()
    assert len(np.unique(ctx.download_map(0)[..., 0])) >= 1
    ctx.close()


def test_set_params_in_place_is_stream_ordered(gpu, oracle_c):
    """lfi_set_params with an unchanged array size (a focus sweep) uploads in stream order without draining the stream: a render already
    enqueued keeps its parameters, the next one sees the new ones — over more calls than there are staging buffers, for both methods, and
    for the focus map + all-focus render of each step."""
    cols = rows = 8
    W, H, V = 640, 120, 64
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    foci = [0.05, 0.23, 0.4, 0.11, 0.3]
    hps = [gpu.build_params(cols, rows, W, H, "0,0,1,1", f, 0.0, 3.0, 1.783, V) for f in foci]
    want = [oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, threads=8) for hp in hps]
    assert not (want[0] == want[1]).all()
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf)
    ctx.set_params(hps[0])
    for i in range(len(hps)):
        ctx.render("STD")                                   # enqueued with hps[i] …
        ctx.set_params(hps[(i + 1) % len(hps)])             # … and NOT waited for: the next parameters are on their way behind it
        got = ctx.download_views()                          # (synchronises)
        assert (got == want[i]).all(), (i, int((got != want[i]).sum()))
    # TEN_WM after the sweep came round: the first parameters again
    ctx.render("TEN_WM")
    ctx.sync()
    m16 = oracle_c.blend_ten(lf, hps[0].focused_offsets, hps[0].offsets, hps[0].weights, model=oracle_c.TEN_M16, threads=8)
    assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= 1
    ctx.close()
    # the sweep with focus maps: every step's map and all-focus render against the oracle, parameters replaced without a wait in between
    W2, H2, V2 = 200, 40, 8
    lf2 = (oracle_c.synthetic_lf(cols * rows, W2, H2, SEED + 1) // 32 * 32).astype(np.uint8)
    lf2[..., 3] = 255
    sw = [gpu.build_params(cols, rows, W2, H2, "0.071,0.071,0.93,0.93", f, 0.17, 3.0, 1.783, V2) for f in (0.1, 0.22, 0.3, 0.15)]
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W2, H2)
    ctx.upload_grid(lf2)
    ctx.set_params(sw[0])
    for i, hp in enumerate(sw):
        ctx.focus_map()
        ctx.render("STD", all_focus=True)
        ctx.set_params(sw[(i + 1) % len(sw)])
        map0 = oracle_c.focus_estimate(lf2, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
        map1 = oracle_c.focus_filter(map0, hp.block_radius)
        assert (ctx.download_map(0) == map0).all() and (ctx.download_map(1) == map1).all(), i
        ref = oracle_c.blend_std(lf2, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=map1, focus=hp.focus, rng=hp.range, threads=8)
        assert (ctx.download_views() == ref).all(), i
    ctx.close()


def test_replaced_images_refresh_only_their_planes(gpu, oracle_c):
    """One or a few images replaced (lfi_upload_image, asynchronous uploads, a partial device fill): the derived planar copy and the
    focus map's padded planes are refreshed for THOSE images only — every render and map must still be the oracle's for the mixed light
    field, in both view layouts, with first / last / neighbouring / scattered images replaced."""
    cols, rows, W, H, V = 5, 3, 200, 24, 8
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.25, 0.17, 2.0, 1.5, V)
    lf = (oracle_c.synthetic_lf(n, W, H, 11) // 16 * 16).astype(np.uint8)
    lf[..., 3] = 255
    other = (oracle_c.synthetic_lf(n, W, H, 12) // 16 * 16).astype(np.uint8)
    other[..., 3] = 255
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf)
    ctx.set_params(hp)

    def check(cur, what):
        for layout in ("rgba", "planar"):
            ctx.set_output_layout(layout)
            ctx.render("STD"); ctx.sync()
            assert (ctx.download_views() == oracle_c.blend_std(cur, hp.focused_offsets, hp.offsets, hp.weights, threads=8)).all(), (what, layout, "STD")
            ctx.render("TEN_WM"); ctx.sync()
            m16 = oracle_c.blend_ten(cur, hp.focused_offsets, hp.offsets, hp.weights, model=oracle_c.TEN_M16, threads=8)
            assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= 1, (what, layout, "TEN_WM")
        ctx.set_output_layout("rgba")
        ctx.focus_map(); ctx.sync()
        map0 = oracle_c.focus_estimate(cur, hp.offsets, hp.focus_map_ids, hp.focus, hp.range, hp.block_radius, threads=8)
        assert (ctx.download_map(0) == map0).all(), (what, "map")

    cur = lf.copy()
    check(cur, "initial")
    sampled = [int(g) for g in hp.focus_map_ids]
    unsampled = [g for g in range(n) if g not in sampled]
    for what, ids, asynchronous in (("first image", [0], False), ("last image", [n - 1], False), ("two neighbours", [6, 7], False),
                                    ("scattered, asynchronous", [1, 5, 9, 13], True), ("a sampled image", sampled[:1], False),
                                    ("an image the map does not sample", unsampled[:1], True)):
        if not ids:
            continue
        for g in ids:
            cur[g] = other[g] if (cur[g] == lf[g]).all() else lf[g]
            (ctx.upload_image_async if asynchronous else ctx.upload_image)(g, cur[g])
        check(cur, what)
    ctx.fill_synthetic(0x55, 4, 9)                       # a partial fill on the device: images 4 … 8
    cur[4:9] = oracle_c.synthetic_lf(n, W, H, 0x55)[4:9]
    check(cur, "partial fill")
    ctx.close()


@pytest.mark.parametrize("cols,rows,V", [(5, 3, 8), (13, 13, 70)])
def test_release_inputs(cols, rows, V, gpu, oracle_c):
    """lfi_release_inputs: the planar copy becomes the only copy of the inputs (round 4; the reference keeps its cudaArrays for the object's
    lifetime, src/interpolator.cu:73-137).  Fixed-focus STD (bit-exact) and TEN_WM (≤ 1 LSB of M16) still match the oracle in both view
    layouts and for parameters with smaller offsets; an image uploaded afterwards replaces its planes through the staging plane; what needs
    the RGBA planes is refused; the footprint of the inputs is below 1.1× the RGBA bytes; lfi_set_grid starts over."""
    W, H = 200, 24
    n = cols * rows
    hp = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.25, 0.17, 2.0, 1.5, V)
    hp_small = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.11, 0.17, 2.0, 1.5, V)       # smaller offsets: the padding covers them
    hp_large = gpu.build_params(cols, rows, W, H, "0.1,0.2,0.8,0.9", 0.9, 0.17, 2.0, 1.5, V)        # larger ones: it does not
    assert np.abs(hp_small.focused_offsets[:, 0]).max() < np.abs(hp.focused_offsets[:, 0]).max() * 1.2 < np.abs(hp_large.focused_offsets[:, 0]).max()
    lf = oracle_c.synthetic_lf(n, W, H, 31)
    other = oracle_c.synthetic_lf(n, W, H, 32)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(lf)
    ctx.set_params(hp)
    rgba_bytes = ctx.memory_info().grid_bytes
    assert rgba_bytes == n * W * H * 4
    ctx.release_inputs()
    ctx.release_inputs()                         # idempotent
    mem = ctx.memory_info()
    assert mem.grid_bytes == 0 and 0 < mem.derived_bytes

    def check(cur, p, what):
        for layout in ("rgba", "planar"):
            ctx.set_output_layout(layout)
            ctx.render("STD"); ctx.sync()
            assert (ctx.download_views() == oracle_c.blend_std(cur, p.focused_offsets, p.offsets, p.weights, threads=8)).all(), (what, layout, "STD")
            ctx.render("TEN_WM"); ctx.sync()
            m16 = oracle_c.blend_ten(cur, p.focused_offsets, p.offsets, p.weights, model=oracle_c.TEN_M16, threads=8)
            assert np.abs(ctx.download_views().astype(int) - m16.astype(int)).max() <= 1, (what, layout, "TEN_WM")
        ctx.set_output_layout("rgba")

    check(lf, hp, "released")
    ctx.set_params(hp_small)
    check(lf, hp_small, "released, smaller offsets")
    cur = lf.copy()
    for g in (0, n - 1, n // 2):                 # replaced images go through the staging plane into the copy
        cur[g] = other[g]
        ctx.upload_image(g, other[g])
    check(cur, hp_small, "released, images replaced")
    ctx.set_params(hp)
    check(cur, hp, "released, back to the first offsets")
    # what needs the RGBA planes is refused
    for call in (lambda: ctx.render("TEN_WM", all_focus=True), lambda: ctx.focus_map(), lambda: ctx.fill_synthetic(1), lambda: ctx.download_coords(0),
                 lambda: ctx.grid_device_ptr(), lambda: ctx.grid_modified()):
        with pytest.raises(gpu.LfiError, match="released"):
            call()
    check(cur, hp, "released, after the refused calls (ADVICE r4: lfi_grid_device_ptr used to cost the context its only copy)")
    ctx.set_variant("TEN_WM", "persist_m2_nt")   # a kernel that reads the RGBA planes
    with pytest.raises(gpu.LfiError, match="released"):
        ctx.render("TEN_WM")
    ctx.set_variant("TEN_WM", "auto")
    ctx.set_params(hp_large)
    with pytest.raises(gpu.LfiError, match="released"):
        ctx.render("TEN_WM")
    # starting over
    ctx.set_grid(cols, rows, W, H)
    ctx.upload_grid(cur)
    ctx.set_params(hp_large)
    ctx.render("STD", all_focus=False); ctx.sync()
    assert (ctx.download_views() == oracle_c.blend_std(cur, hp_large.focused_offsets, hp_large.offsets, hp_large.weights, threads=8)).all()
    ctx.close()
    # release, upload, a LARGER row window, release, upload (the one-image staging plane kept the first window's size and the second
    # upload overflowed it)
    ctx = gpu.Context(0)
    ctx.set_grid(cols, rows, W, H)
    for band in ((10, 14), (0, H), (11, 13), (0, H)):
        in_rows = gpu.input_rows(band, hp_small.focused_offsets, H)
        ctx.set_row_window(band[0], band[1], in_rows[0], in_rows[1])
        ctx.upload_grid(lf)
        ctx.set_params(hp_small)
        ctx.release_inputs()
        cur = lf.copy()
        for g in (1, n - 2):
            cur[g] = other[g]
            ctx.upload_image(g, other[g])
        ctx.render("STD"); ctx.sync()
        want = oracle_c.blend_std(cur, hp_small.focused_offsets, hp_small.offsets, hp_small.weights, threads=8)
        assert (ctx.download_views()[:, band[0]:band[1]] == want[:, band[0]:band[1]]).all(), band
    ctx.close()
    # the footprint at a BASELINE-like width (padding is relative to the offsets, not to the image count)
    ctx = gpu.Context(0)
    ctx.set_grid(8, 8, 1920, 64)
    ctx.fill_synthetic(SEED)
    ctx.set_params(gpu.build_params(8, 8, 1920, 64, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 8))
    before = ctx.memory_info()
    ctx.release_inputs()
    after = ctx.memory_info()
    assert after.grid_bytes + after.derived_bytes <= 1.1 * before.grid_bytes, (before, after)
    ctx.close()
