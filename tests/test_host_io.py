"""CPU: the steps either side of the hot path — PNG/PPM decode and encode (csrc/host/image_io.cpp) and the grid loader
(csrc/host/lfLoader.cpp, reference src/lfLoader.cpp:8-67) — against Pillow."""
import os

import numpy as np
import pytest
from PIL import Image


def _rand(h, w, c, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("mode,channels", [("RGBA", 4), ("RGB", 3), ("L", 1), ("LA", 2), ("P", 1)])
def test_png_decode_matches_pillow(mode, channels, native, tmp_path):
    arr = _rand(13, 37, channels, 3)
    if mode == "P":
        img = Image.fromarray(arr[..., 0], "P")
        img.putpalette(list(np.random.default_rng(1).integers(0, 256, 768, dtype=np.uint8)))
    else:
        img = Image.fromarray(arr if channels > 1 else arr[..., 0], mode)
    path = str(tmp_path / f"img_{mode}.png")
    img.save(path)
    got = native.load_image(path)
    want = np.array(Image.open(path).convert("RGBA"))
    assert got.shape == want.shape and (got == want).all()


def test_png_16bit_and_filters(native, tmp_path):
    # smooth gradients make Pillow's encoder pick Sub/Up/Average/Paeth filters
    y, x = np.mgrid[0:64, 0:96]
    arr = np.stack([(x * 2) % 256, (y * 3) % 256, ((x + y) * 5 // 3) % 256], -1).astype(np.uint8)
    path = str(tmp_path / "grad.png")
    Image.fromarray(arr, "RGB").save(path, optimize=True)
    assert (native.load_image(path)[..., :3] == arr).all()
    g16 = (np.arange(40 * 30, dtype=np.uint16).reshape(30, 40) * 53)
    path16 = str(tmp_path / "g16.png")
    Image.fromarray(g16, "I;16").save(path16)
    assert (native.load_image(path16)[..., 0] == (g16 >> 8)).all()


def test_png_encode_roundtrip_and_ppm(native, tmp_path):
    arr = _rand(21, 33, 4, 9)
    path = str(tmp_path / "out.png")
    native.write_png(path, arr)
    assert (np.array(Image.open(path)) == arr).all()
    assert (native.load_image(path) == arr).all()
    ppm = str(tmp_path / "x.ppm")
    Image.fromarray(arr[..., :3], "RGB").save(ppm)
    got = native.load_image(ppm)
    assert (got[..., :3] == arr[..., :3]).all() and (got[..., 3] == 255).all()


def test_bad_images_raise(native, tmp_path):
    p = tmp_path / "junk.png"
    p.write_bytes(b"not a png at all")
    with pytest.raises(RuntimeError, match="Cannot load image"):
        native.load_image(str(p))
    with pytest.raises(RuntimeError, match="Cannot load image"):
        native.load_image(str(tmp_path / "missing.png"))


def test_loader_grid_order_and_errors(native, tmp_path):
    # files are <row>_<col>.png (reference src/lfLoader.cpp:22-31); non-square 3 cols × 2 rows; g = col*rows + row
    cols, rows, W, H = 3, 2, 10, 6
    d = tmp_path / "lf"
    d.mkdir()
    imgs = {}
    for row in range(rows):
        for col in range(cols):
            a = _rand(H, W, 4, row * 10 + col)
            a[..., 3] = 255
            imgs[(col, row)] = a
            Image.fromarray(a, "RGBA").save(d / f"{row:02d}_{col:02d}.png")
    c, r, planes = native.load_grid(str(d))
    assert (c, r) == (cols, rows) and planes.shape == (6, H, W, 4)
    for col in range(cols):
        for row in range(rows):
            assert (planes[col * rows + row] == imgs[(col, row)]).all()
    with pytest.raises(RuntimeError, match="does not exist"):
        native.load_grid(str(tmp_path / "nope"))
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(RuntimeError, match="empty"):
        native.load_grid(str(empty))
    (d / "README.txt").write_text("x")
    with pytest.raises(RuntimeError, match="not named properly"):
        native.load_grid(str(d))
    os.remove(d / "README.txt")
    os.remove(d / "01_02.png")
    with pytest.raises(RuntimeError, match="missing"):
        native.load_grid(str(d))
