import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_c():
    """The C oracle (test infrastructure), built on demand with gcc."""
    from oracle import lfi_oracle_c
    lfi_oracle_c.build()
    lfi_oracle_c.lib()
    return lfi_oracle_c


@pytest.fixture(scope="session")
def oracle_np():
    from oracle import lfi_oracle_np
    return lfi_oracle_np


@pytest.fixture(scope="session")
def native():
    """The product libraries; built in-tree if a source is newer (make is a no-op otherwise)."""
    import lfinterpolator_amd as L
    if not (os.path.exists(L.build.HIP_LIB) and os.path.exists(L.build.HOST_LIB)):
        L.build_all()
    return L


@pytest.fixture(scope="session")
def gpu(native, request):
    """A context factory on cuda:0.  A GPU without the built library is an error (never a fallback).  No GPU: an error too when
    the GPU tests were asked for (`-m gpu`: a broken HIP environment must not read as "skipped"); a skip only when the GPU tests
    were swept up by an unfiltered run on a machine without a device."""
    lib = native.load_hip_library()
    if lib.lfi_device_count() <= 0:
        if "gpu" in (request.config.getoption("-m") or "") and "not gpu" not in request.config.getoption("-m"):
            pytest.fail("-m gpu was requested but no HIP device is visible: " + lib.lfi_last_error(None).decode())
        pytest.skip("no HIP device on this machine")
    return native


# (name, cols, rows, W, H, views, trajectory, focus, aspect, effect)
SMALL_CASES = [
    ("g3x3_16x16_v8", 3, 3, 16, 16, 8, "0,0,1,1", 0.23, 1.783, 3.0),
    ("g8x8_32x24_v64", 8, 8, 32, 24, 64, "0.0,0.0,1.0,1.0", 0.23, 1.783, 3.0),
    ("g15x15_16x16_v8", 15, 15, 16, 16, 8, "0,0.5,1,0.5", 0.23, 1.783, 3.0),
    ("g4x4_33x17_v5", 4, 4, 33, 17, 5, "0.071,0.071,0.93,0.93", 0.43, 1.8266, 7.0),
]
SEED = 0x1F1F
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
