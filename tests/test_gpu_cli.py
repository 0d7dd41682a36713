"""GPU (-m gpu): the command-line program end to end — PNG grid in, NN.png out — against the oracle, with the reference's
flags (-i -o -t -a -m -f -s -r; reference src/main.cpp:7-43) and error behaviour."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu


def _run(native, *args):
    return subprocess.run([native.build.CLI, *args], capture_output=True, text=True, timeout=300)


def _write_grid(oracle_c, d, cols, rows, W, H, seed):
    lf = oracle_c.synthetic_lf(cols * rows, W, H, seed)
    for col in range(cols):
        for row in range(rows):
            Image.fromarray(lf[col * rows + row], "RGBA").save(os.path.join(d, f"{row:02d}_{col:02d}.png"))
    return lf


@pytest.mark.parametrize("method", ["TEN_WM", "STD"])
def test_cli_renders_png_grid(method, gpu, oracle_c, tmp_path):
    cols = rows = 4
    W, H = 80, 24
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    lf = _write_grid(oracle_c, str(src), cols, rows, W, H, 11)
    res = _run(gpu, "-i", str(src), "-o", str(dst), "-t", "0.0,0.0,1.0,1.0", "-a", "1.783", "-m", method, "-f", "0.23", "-b", "3")
    assert res.returncode == 0, res.stderr
    assert "Average time of 3 runs" in res.stdout
    hp = gpu.build_params(cols, rows, W, H, "0.0,0.0,1.0,1.0", 0.23, 0.0, 3.0, 1.783, 64)
    if method == "STD":
        want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights)
    else:
        want = oracle_c.blend_ten(lf, hp.focused_offsets, hp.offsets, hp.weights)
    files = sorted(os.listdir(dst))
    assert files == [f"{i:02d}.png" for i in range(64)]           # always 64 views by default, as the reference
    for v in (0, 31, 63):
        got = np.array(Image.open(dst / f"{v:02d}.png"))
        assert np.abs(got.astype(int) - want[v].astype(int)).max() <= (0 if method == "STD" else 1)


def test_cli_all_focus_writes_maps(gpu, oracle_c, tmp_path):
    cols = rows = 4
    W, H = 64, 16
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    lf = _write_grid(oracle_c, str(src), cols, rows, W, H, 5)
    res = _run(gpu, "-i", str(src), "-o", str(dst), "-t", "0.071,0.071,0.93,0.93", "-a", "2.0223", "-m", "STD", "-f", "0.1", "-r",
               "0.3", "-s", "7", "-n", "4", "-b", "1")
    assert res.returncode == 0, res.stderr
    assert "Estimating focus map" in res.stdout
    assert sorted(os.listdir(dst)) == ["00.png", "01.png", "02.png", "03.png", "map0.png", "map1.png"]
    hp = gpu.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.1, 0.3, 7.0, 2.0223, 4)
    map0 = oracle_c.focus_estimate(lf, hp.offsets, hp.focus_map_ids, 0.1, 0.3, hp.block_radius)
    map1 = oracle_c.focus_filter(map0, hp.block_radius)
    assert (np.array(Image.open(dst / "map0.png")) == map0).all()
    assert (np.array(Image.open(dst / "map1.png")) == map1).all()
    want = oracle_c.blend_std(lf, hp.focused_offsets, hp.offsets, hp.weights, all_focus=True, map_plane=map1, focus=0.1, rng=0.3)
    for v in range(4):
        assert (np.array(Image.open(dst / f"{v:02d}.png")) == want[v]).all()


def test_cli_errors_and_help(gpu, tmp_path):
    res = _run(gpu, "-h")
    assert res.returncode == 0 and "-t - trajectory of the camera" in res.stdout
    res = _run(gpu, "-i", str(tmp_path))
    assert res.returncode != 0 and "Missing required parameters" in res.stderr
    res = _run(gpu, "-i", str(tmp_path / "nope"), "-o", str(tmp_path / "o"), "-t", "0,0,1,1", "-m", "STD")
    assert res.returncode != 0 and "does not exist" in res.stderr
    res = _run(gpu, "--synthetic", "3,3,32,8", "-o", str(tmp_path / "o"), "-t", "0,0,1,1", "-m", "NOPE", "-b", "1")
    assert res.returncode != 0 and "The specified interpolation method does not exist!" in res.stderr
    res = _run(gpu, "--synthetic", "3,3,256,256", "-o", str(tmp_path / "o1"), "-t", "0,0,1,1", "-m", "TEN_WM", "-n", "1", "-b", "2")
    assert res.returncode == 0 and os.listdir(tmp_path / "o1") == ["00.png"]   # BASELINE config 1's shape on the GPU path


def test_cli_quilt(gpu, oracle_c, tmp_path):
    """-q cols,rows writes quilt.png = the NN.png tiles montaged left to right, top to bottom (scripts/viewsToQuilt.sh)."""
    dst = tmp_path / "out"
    res = _run(gpu, "--synthetic", "4,4,48,20", "-o", str(dst), "-t", "0,0.5,1,0.5", "-m", "TEN_WM", "-f", "0.1", "-n", "6", "-q", "3,2",
               "-b", "1")
    assert res.returncode == 0, res.stderr
    quilt = np.array(Image.open(dst / "quilt.png"))
    assert quilt.shape == (2 * 20, 3 * 48, 4)
    for v in range(6):
        tile = np.array(Image.open(dst / f"{v:02d}.png"))
        ty, tx = divmod(v, 3)
        assert (quilt[ty * 20:(ty + 1) * 20, tx * 48:(tx + 1) * 48] == tile).all()
    res = _run(gpu, "--synthetic", "4,4,48,20", "-o", str(dst), "-t", "0,0,1,1", "-m", "STD", "-n", "4", "-q", "3,2", "-b", "1")
    assert res.returncode != 0 and "more tiles than rendered views" in res.stderr
