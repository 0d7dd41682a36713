"""CPU: the oracle against its golden fixtures, the two restatements against each other, fp16 helpers.

PARITY UNPINNED (see oracle/lfi_oracle.h): the reference has no golden vectors; these tests pin the oracle itself.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, SEED, SMALL_CASES


def _golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def test_f16_conversions_match_numpy(oracle_c):
    l = oracle_c.lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.standard_normal(2000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 5, 2000).astype(np.float32),
        np.array([0.0, -0.0, 1.0, 65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                  2.0 ** -14, 2.0 ** -14 * 0.99999, 5.96e-8, 1.19e-7, np.inf, -np.inf], dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
        want64 = vals.astype(np.float64).astype(np.float16).view(np.uint16)
    for v, w, w64 in zip(vals, want, want64):
        assert l.lfo_f32_to_f16(float(v)) == int(w), v
        assert l.lfo_f64_to_f16(float(v)) == int(w64), v
    # every half bit pattern: widening is exact and the truncating u8 conversion saturates
    for bits in range(0, 1 << 16, 7):
        h = np.array([bits], dtype=np.uint16).view(np.float16)[0]
        f = l.lfo_f16_to_f32(bits)
        if np.isnan(h):
            assert np.isnan(f)
            assert l.lfo_f16_to_u8_rz(bits) == 0
        else:
            assert f == float(h)
            assert l.lfo_f16_to_u8_rz(bits) == int(np.floor(np.clip(float(h), 0, 255)))
    # double → half must round once (a value that rounds differently through float)
    tricky = float(np.float64(1.0) + 2.0 ** -11 + 2.0 ** -30)  # just above the tie between 1 and 1+2^-10
    assert l.lfo_f64_to_f16(tricky) == 0x3c01


def test_known_answers_readme_example(oracle_c):
    # 8×8 @1920×1080, -t 0,0,1,1 -a 1.783 -f 0.23
    se = oracle_c.interpret_trajectory("0,0,1,1", 8, 8)
    assert se.tolist() == [0.0, 0.0, 7.0, 7.0]
    off, foc = oracle_c.offsets(se, 8, 8, 1920, 1080, 1.783, 0.23)
    assert np.abs(off).max(0)[0] == 840.0 and abs(np.abs(off).max(0)[1] - 471.116) < 1e-3
    assert np.abs(foc).max(0).tolist() == [193, 108]
    assert oracle_c.block_radius(1920, 1080).tolist() == [20, 10]
    assert oracle_c.block_radius(3840, 2160).tolist() == [38, 22]
    assert oracle_c.block_radius(64, 48).tolist() == [1, 1]
    w = oracle_c.weight_matrix_f16(se, 8, 8, 64, 3.0).view(np.float16).astype(np.float64)
    assert np.all(np.abs(w.sum(1) - 1.0) < 2e-3)
    assert w[0].argmax() == 0 and w[63].argmax() == 63  # view 0 sits on image (0,0), view 63 on (7,7)
    ids = oracle_c.focus_map_ids(se, 8, 8)
    assert len(ids) == 32 and set(ids[:4].tolist()) == {27, 28, 35, 36}  # the four images around the centre (3.5,3.5)
    assert len(oracle_c.focus_map_ids(oracle_c.interpret_trajectory("0,0,1,1", 3, 3), 3, 3)) == 9  # defect D4 guard


def test_subnormal_weights_occur_with_effect_7(oracle_c):
    se = oracle_c.interpret_trajectory("0.071,0.071,0.93,0.93", 15, 15)
    w = oracle_c.weight_matrix_f16(se, 15, 15, 8, 7.0)
    assert ((w & 0x7c00) == 0).any() and (w != 0).all() is not None  # fp16 subnormals are part of the input domain


@pytest.mark.parametrize("case", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
def test_oracles_reproduce_golden(case, oracle_c, oracle_np):
    name, cols, rows, W, H, V, traj, focus, aspect, effect = case
    g = _golden(name)
    se = oracle_c.interpret_trajectory(traj, cols, rows)
    w = oracle_c.weight_matrix_f16(se, cols, rows, V, effect)
    off, foc = oracle_c.offsets(se, cols, rows, W, H, aspect, focus)
    assert (w == g["weights"]).all() and (off == g["offsets"]).all() and (foc == g["focused"]).all()
    assert (oracle_c.focus_map_ids(se, cols, rows) == g["ids"]).all()
    lf = oracle_c.synthetic_lf(cols * rows, W, H, SEED)
    assert (lf == g["lf"]).all()
    assert (oracle_c.blend_std(lf, foc, off, w) == g["std"]).all()
    assert (oracle_c.blend_ten(lf, foc, off, w, model=oracle_c.TEN_M16) == g["ten_m16"]).all()
    assert (oracle_c.blend_ten(lf, foc, off, w, model=oracle_c.TEN_EXACT) == g["ten_exact"]).all()
    # the independent numpy restatement reaches the same bytes
    assert (oracle_np.blend_std(lf, foc, off, w) == g["std"]).all()
    assert (oracle_np.blend_ten(lf, foc, off, w, model=oracle_np.TEN_M16) == g["ten_m16"]).all()
    # all-focus leg
    rng = float(g["range"])
    map0 = oracle_c.focus_estimate(lf, off, g["ids"], focus, rng, g["radius"])
    assert (map0 == g["map0"]).all()
    assert (oracle_c.focus_filter(map0, g["radius"]) == g["map1"]).all()
    assert (oracle_c.blend_std(lf, foc, off, w, all_focus=True, map_plane=g["map1"], focus=focus, rng=rng) == g["af_std"]).all()
    assert (oracle_c.blend_ten(lf, foc, off, w, all_focus=True, map_plane=g["map1"], focus=focus, rng=rng) == g["af_ten_m16"]).all()
    # the reference's tensor kernel reads the unfiltered map (src/kernels.cu:430)
    assert (oracle_c.blend_ten(lf, foc, off, w, all_focus=True, map_plane=g["map0"], focus=focus, rng=rng) == g["af_ten_m16_map0"]).all()


def test_oracle_properties(oracle_c):
    """Domain properties the GPU tests reuse at full size: identity under one-hot weights, view-range and row-range
    independence, tolerance between the two tensor models and the exact blend."""
    cols = rows = 4
    W, H, V = 40, 12, 16
    lf = oracle_c.synthetic_lf(16, W, H, 7)
    se = oracle_c.interpret_trajectory("0,0,1,1", cols, rows)
    off, foc = oracle_c.offsets(se, cols, rows, W, H, 1.0, 0.2)
    onehot = np.zeros((V, 16), dtype=np.uint16)
    onehot[np.arange(V), np.arange(V)] = 0x3c00  # fp16 1.0
    for out in (oracle_c.blend_std(lf, foc, off, onehot), oracle_c.blend_ten(lf, foc, off, onehot)):
        for v in range(V):
            ys = np.clip(np.arange(H) + foc[v, 1], 0, H - 1)
            xs = np.clip(np.arange(W) + foc[v, 0], 0, W - 1)
            want = lf[v][ys][:, xs].copy()
            want[..., 3] = 255
            assert (out[v] == want).all()
    w = oracle_c.weight_matrix_f16(se, cols, rows, V, 3.0)
    full = oracle_c.blend_std(lf, foc, off, w)
    part = oracle_c.blend_std(lf, foc, off, w, v0=4, v1=9)
    assert (part[4:9] == full[4:9]).all() and (part[:4] == 0).all()
    threaded = oracle_c.blend_std(lf, foc, off, w, threads=3)
    assert (threaded == full).all()
    m16, pre16 = oracle_c.blend_ten(lf, foc, off, w, model=oracle_c.TEN_M16, return_prequant=True)
    ex, preex = oracle_c.blend_ten(lf, foc, off, w, model=oracle_c.TEN_EXACT, return_prequant=True)
    exact = oracle_c.blend_f64(lf, foc, off, w)
    assert np.abs(m16.astype(int) - ex.astype(int)).max() <= 1
    assert np.abs(pre16 - exact).max() / 255.0 <= 1e-3  # ≤1e-3 normalised before quantisation
    assert np.abs(preex - exact).max() <= 0.0625 + 1e-9  # one fp16 rounding: half an ulp of [128,256)


def test_edge_cases(oracle_c):
    # single view, start == end trajectory, offsets larger than the image (everything clamps), 1×1 grid
    se = oracle_c.interpret_trajectory("0.5,0.5,0.5,0.5", 3, 3)
    w = oracle_c.weight_matrix_f16(se, 3, 3, 1, 3.0)
    assert w.shape == (1, 9) and not np.isnan(w.view(np.float16)).any()
    w4 = oracle_c.weight_matrix_f16(se, 3, 3, 4, 3.0)
    assert (w4 == w4[0]).all()  # zero-length trajectory: every view has the same weights
    lf = oracle_c.synthetic_lf(9, 8, 6, 3)
    off, foc = oracle_c.offsets(se, 3, 3, 8, 6, 1.0, 5.0)  # |offset·focus| up to ±13 px on an 8×6 image
    out = oracle_c.blend_std(lf, foc, off, w)
    assert out.shape == (1, 6, 8, 4) and (out[..., 3] == 255).all()
    lf1 = oracle_c.synthetic_lf(1, 5, 5, 1)
    se1 = oracle_c.interpret_trajectory("0,0,0,0", 1, 1)
    w1 = oracle_c.weight_matrix_f16(se1, 1, 1, 2, 3.0)
    off1, foc1 = oracle_c.offsets(se1, 1, 1, 5, 5, 1.0, 0.0)
    assert (w1 == 0x3c00).all()
    assert (oracle_c.blend_ten(lf1, foc1, off1, w1)[0] == lf1[0]).all()
