"""CPU: the C-ABI library loads and exports exactly what include/lfi.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest


def _declared_symbols(root):
    text = open(os.path.join(root, "include", "lfi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfi_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(native):
    from conftest import ROOT
    declared = _declared_symbols(ROOT)
    assert declared == sorted(native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(native):
    from conftest import ROOT
    lib = ctypes.CDLL(native.build.HIP_LIB)
    for name in _declared_symbols(ROOT):
        assert hasattr(lib, name), name
    lib.lfi_abi_version.restype = ctypes.c_int
    assert lib.lfi_abi_version() == 1


def test_header_compiles_as_plain_c(tmp_path):
    from conftest import ROOT
    src = tmp_path / "t.c"
    src.write_text('#include "lfi.h"\nint main(void){ lfi_params p; (void)p; return LFI_ABI_VERSION - 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                           str(tmp_path / "t.o")])


def test_code_object_is_gfx950(native):
    # the fat binary inside the shared library must carry gfx950 code objects and no other GPU target
    data = open(native.build.HIP_LIB, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets


def test_no_gpu_means_error_not_fallback(native):
    lib = native.load_hip_library()
    if lib.lfi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.LfiError, match="no CPU fallback"):
        native.Context(0)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under lfinterpolator_amd/ or include/ may reference it."""
    from conftest import ROOT
    bad = []
    for base in ("lfinterpolator_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"lfi_oracle|lfo_[a-z]|from oracle|import oracle|oracle/", text):
                        # comments that merely NAME the oracle are fine; includes / imports / calls are not
                        if re.search(r'#include\s+"[^"]*oracle|import\s+oracle|from\s+oracle|lfo_[a-z0-9_]+\s*\(', text):
                            bad.append(os.path.join(dirpath, f))
    assert not bad, bad
