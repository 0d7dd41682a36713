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


# ---- against the reference's own codec (oracle/_ref/libref_codec.so = stb_image / stb_image_write built from the reference
# tree by `make -C oracle ref`): what the reference's LfLoader would have handed to its kernels, pixel for pixel -------------
def _ref_codec():
    from oracle import ref_codec
    if not ref_codec.available():
        pytest.skip("oracle/_ref/libref_codec.so not built (needs /root/reference)")
    return ref_codec


def _png_cases(tmp_path):
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:29, 0:45]
    grad = np.stack([(x * 5) % 256, (y * 7) % 256, ((x + 2 * y) * 3) % 256, 255 - (x * 4) % 256], -1).astype(np.uint8)
    cases = {}
    cases["rgba"] = Image.fromarray(grad, "RGBA")
    cases["rgb"] = Image.fromarray(grad[..., :3], "RGB")
    cases["grey"] = Image.fromarray(grad[..., 0], "L")
    cases["grey_alpha"] = Image.fromarray(grad[..., [0, 3]], "LA")
    pal = Image.fromarray((grad[..., 0] // 4).astype(np.uint8), "P")
    pal.putpalette(list(rng.integers(0, 256, 768, dtype=np.uint8)))
    cases["palette"] = pal
    cases["grey16"] = Image.fromarray((np.arange(29 * 45, dtype=np.uint16).reshape(29, 45) * 47), "I;16")
    cases["bilevel"] = Image.fromarray(((x + y) % 3 == 0).astype(np.uint8) * 255, "L").convert("1")
    paths = {}
    for name, img in cases.items():
        p = str(tmp_path / f"{name}.png")
        img.save(p)
        paths[name] = p
    # palette with per-entry alpha (tRNS) and RGB with a colour key (tRNS)
    p = str(tmp_path / "palette_trns.png")
    pal.save(p, transparency=bytes(rng.integers(0, 256, 64, dtype=np.uint8)))
    paths["palette_trns"] = p
    p = str(tmp_path / "rgb_colourkey.png")
    Image.fromarray(grad[..., :3] // 64 * 64, "RGB").save(p, transparency=(64, 128, 0))
    paths["rgb_colourkey"] = p
    return paths


def test_png_decode_matches_reference_codec(native, tmp_path):
    ref = _ref_codec()
    for name, path in _png_cases(tmp_path).items():
        want = ref.load_rgba(path)
        got = native.load_image(path)
        assert got.shape == want.shape, name
        assert (got == want).all(), (name, int((got != want).sum()))


def test_png_writers_round_trip_through_each_other(native, tmp_path):
    ref = _ref_codec()
    arr = _rand(31, 47, 4, 5)
    ours = str(tmp_path / "ours.png")
    native.write_png(ours, arr)
    assert (ref.load_rgba(ours) == arr).all()          # the reference's decoder reads our files
    theirs = str(tmp_path / "theirs.png")
    ref.write_png(theirs, arr)
    assert (native.load_image(theirs) == arr).all()    # we read what stbi_write_png (src/interpolator.cu:313) writes


def _adam7_png(arr):
    """An Adam7-interlaced 8-bit RGB(A) PNG of arr (Pillow cannot write one): seven reduced images, filter type 0."""
    import struct
    import zlib
    h, w, c = arr.shape
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = arr[y0::dy, x0::dx]
        if sub.size:
            raw += b"".join(b"\x00" + row.tobytes() for row in sub)

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("shape", [(37, 53, 4), (5, 3, 3), (1, 1, 4), (9, 2, 3)], ids=lambda s: "x".join(map(str, s)))
def test_interlaced_png_matches_reference_codec(shape, native, tmp_path):
    ref = _ref_codec()
    arr = _rand(*shape, 21)
    path = str(tmp_path / "adam7.png")
    with open(path, "wb") as f:
        f.write(_adam7_png(arr))
    want = ref.load_rgba(path)
    assert (want[..., :shape[2]] == arr).all()
    got = native.load_image(path)
    assert got.shape == want.shape and (got == want).all()


def _photo(h, w, seed):
    """Smooth content with edges and noise: exercises every AC band, both signs, chroma detail."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0) + 20 * rng.standard_normal((h, w))
    g = 128 + 90 * np.sin((x + y) / 13.0) + (((x // 16 + y // 16) % 2) * 60 - 30)
    b = 255 * ((x * 3 + y * 5) % 97) / 97.0
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(64, 96), (67, 101), (8, 8), (1, 1), (17, 3), (250, 333)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("kind", ["444", "422", "420", "grey", "420_restart", "444_progressive", "420_progressive", "q30_420", "q100_444"])
def test_jpeg_decode_matches_reference_codec(kind, size, native, tmp_path):
    """JPEG input (csrc/host/jpeg.cpp) against stb_image as the reference builds it: bit-exact RGBA."""
    ref = _ref_codec()
    h, w = size
    arr = _photo(h, w, 7 + h)
    img = Image.fromarray(arr, "RGB")
    opts = {"quality": 85}
    if kind == "grey":
        img = img.convert("L")
    if "444" in kind:
        opts["subsampling"] = 0
    if "422" in kind:
        opts["subsampling"] = 1
    if "420" in kind:
        opts["subsampling"] = 2
    if "restart" in kind:
        opts["restart_marker_blocks"] = 3
    if "progressive" in kind:
        opts["progressive"] = True
    if kind.startswith("q30"):
        opts["quality"] = 30
    if kind.startswith("q100"):
        opts["quality"] = 100
    path = str(tmp_path / f"{kind}.jpg")
    img.save(path, "JPEG", **opts)
    want = ref.load_rgba(path)
    got = native.load_image(path)
    assert got.shape == want.shape
    assert (got == want).all(), (int((got != want).sum()), int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_jpeg_colour_models_match_reference_codec(native, tmp_path):
    """Adobe-marked files: CMYK (four components), RGB kept as RGB (no YCbCr transform)."""
    ref = _ref_codec()
    arr = _photo(40, 56, 3)
    cases = {"cmyk": (Image.fromarray(arr, "RGB").convert("CMYK"), {"quality": 90})}
    try:
        probe = str(tmp_path / "probe.jpg")
        Image.fromarray(arr, "RGB").save(probe, "JPEG", keep_rgb=True)
        cases["rgb_kept"] = (Image.fromarray(arr, "RGB"), {"quality": 90, "keep_rgb": True})
    except (TypeError, OSError, ValueError):
        pass
    for name, (img, opts) in cases.items():
        path = str(tmp_path / f"{name}.jpg")
        img.save(path, "JPEG", **opts)
        want = ref.load_rgba(path)
        got = native.load_image(path)
        assert got.shape == want.shape and (got == want).all(), name


def test_jpeg_truncated_and_corrupt_files_do_not_crash(native, tmp_path):
    arr = _photo(48, 64, 5)
    good = str(tmp_path / "good.jpg")
    Image.fromarray(arr, "RGB").save(good, "JPEG", quality=80, subsampling=2, progressive=True)
    data = open(good, "rb").read()
    rng = np.random.default_rng(0)
    for i, cut in enumerate([3, 20, 100, len(data) // 3, len(data) // 2, len(data) - 2]):
        p = str(tmp_path / f"cut{i}.jpg")
        open(p, "wb").write(data[:cut])
        try:
            img = native.load_image(p)
            assert img.shape == (48, 64, 4)      # a truncated scan still yields an image, like the reference's decoder
        except RuntimeError as e:
            assert "Cannot load image" in str(e)
    for i in range(8):                           # random byte damage after the headers
        b = bytearray(data)
        for k in rng.integers(len(data) // 4, len(data), 6):
            b[k] = int(rng.integers(0, 256))
        p = str(tmp_path / f"bad{i}.jpg")
        open(p, "wb").write(bytes(b))
        try:
            native.load_image(p)
        except RuntimeError as e:
            assert "Cannot load image" in str(e)


def test_loader_reads_a_jpeg_grid(native, tmp_path):
    """LfLoader on a directory of column_row.jpg files (reference naming, src/lfLoader.cpp:22-31)."""
    ref = _ref_codec()
    for col in range(2):
        for row in range(3):
            Image.fromarray(_photo(24, 40, 10 * col + row), "RGB").save(str(tmp_path / f"{row}_{col}.jpg"), "JPEG", quality=92)
    cols, rows, lf = native.load_grid(str(tmp_path))
    assert (cols, rows) == (2, 3) and lf.shape == (6, 24, 40, 4)
    for col in range(2):
        for row in range(3):
            assert (lf[col * rows + row] == ref.load_rgba(str(tmp_path / f"{row}_{col}.jpg"))).all()
