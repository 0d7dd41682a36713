"""ctypes binding of oracle/_ref/libref_codec.so — the reference's OWN image codec (stb_image / stb_image_write, built from
/root/reference/src/libs by `make -C oracle ref`).  TEST INFRASTRUCTURE ONLY: the checker for csrc/host/image_io.cpp.

load_rgba(path) = what the reference's LfLoader::loadImage gets (stbi_load(…, STBI_rgb_alpha), src/lfLoader.cpp:36);
write_png(path, rgba) = what Interpolator::storeResults writes (stbi_write_png, src/interpolator.cu:313)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libref_codec.so")
_lib = None


def available(build: bool = True) -> bool:
    """True when the library exists (it is built on demand where the reference tree is present)."""
    if not os.path.exists(_SO) and build and os.path.isdir("/root/reference/src/libs"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=False, capture_output=True)
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref_codec.so is missing (make -C oracle ref needs /root/reference)")
        _lib = C.CDLL(_SO)
        _lib.stbi_load.restype = C.POINTER(C.c_uint8)
        _lib.stbi_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        _lib.stbi_image_free.argtypes = [C.c_void_p]
        _lib.stbi_failure_reason.restype = C.c_char_p
        _lib.stbi_write_png.restype = C.c_int
        _lib.stbi_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return _lib


def load_rgba(path: str) -> np.ndarray:
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = lib().stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(c), 4)
    if not p:
        raise RuntimeError(f"stbi_load({path}): {lib().stbi_failure_reason().decode()}")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib().stbi_image_free(p)


def write_png(path: str, rgba: np.ndarray) -> None:
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, ch = rgba.shape
    if not lib().stbi_write_png(path.encode(), w, h, ch, rgba.ctypes.data, w * ch):
        raise RuntimeError(f"stbi_write_png({path}) failed")
