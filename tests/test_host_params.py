"""CPU: the product's host parameterisation (csrc/host/params.cpp, through liblfi_host.so) against the oracle's
restatement of reference src/interpolator.cu:139-246, 318-337 — the bytes handed to the kernels must be identical."""
import numpy as np
import pytest

from conftest import SMALL_CASES

CASES = [(c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9]) for c in SMALL_CASES] + [
    (8, 8, 1920, 1080, 64, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0),        # README example
    (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.0, 2.0223, 7.0),  # focusMapCompare.sh row 1
    (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 7.0),            # 45-view quilt sweep
    (8, 8, 3840, 2160, 256, "0,0,1,1", 0.23, 1.783, 3.0),
    (5, 3, 100, 60, 7, "1,0,0,1", 0.5, 0.75, 2.5),                         # non-square grid, non-integer effect
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}_{c[2]}x{c[3]}_v{c[4]}" for c in CASES])
def test_host_params_equal_oracle(case, native, oracle_c):
    cols, rows, W, H, V, traj, focus, aspect, effect = case
    hp = native.build_params(cols, rows, W, H, traj, focus, 0.3, effect, aspect, V)
    se = oracle_c.interpret_trajectory(traj, cols, rows)
    off, foc = oracle_c.offsets(se, cols, rows, W, H, aspect, focus)
    assert (hp.offsets == off).all()
    assert (hp.focused_offsets == foc).all()
    assert (hp.weights == oracle_c.weight_matrix_f16(se, cols, rows, V, effect)).all()
    assert (hp.focus_map_ids == oracle_c.focus_map_ids(se, cols, rows)).all()
    assert (hp.block_radius == oracle_c.block_radius(W, H)).all()


def test_half_conversion_matches_numpy(native):
    lib = native.load_host_library()
    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.random(3000, dtype=np.float32) * np.float32(2.0) ** rng.integers(-30, 17, 3000).astype(np.float32),
                           np.array([0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.5, 65504, 65519.996, 65520, 3e38, np.inf], np.float32)])
    vals = np.concatenate([vals, -vals])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        assert lib.lfi_host_float_to_half(float(v)) == int(w), v
    for bits in range(0, 1 << 16, 3):
        h = np.array([bits], np.uint16).view(np.float16)[0]
        f = lib.lfi_host_half_to_float(bits)
        assert (np.isnan(f) and np.isnan(h)) or f == float(h)


def test_bad_trajectory_is_an_error(native):
    with pytest.raises(ValueError):
        native.build_params(8, 8, 64, 48, "0,0,1", 0.1, 0.0, 3.0, 1.0, 64)
    with pytest.raises(ValueError):
        native.build_params(8, 8, 64, 48, "a,b,c,d", 0.1, 0.0, 3.0, 1.0, 64)
    with pytest.raises(ValueError):
        native.build_params(8, 8, 64, 48, "0,0,1,1", 0.1, 0.0, 3.0, 1.0, 0)


def test_row_restriction_for_view_sharding(native):
    hp = native.build_params(8, 8, 64, 48, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 64)
    part = hp.rows(16, 24)
    assert part.weights.shape == (8, 64) and (part.weights == hp.weights[16:24]).all()
    assert part.offsets is hp.offsets
