"""In-tree build of the gfx950 libraries and the command-line binary (drives csrc/Makefile: hipcc + g++)."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
BIN_DIR = os.path.join(_PKG, "bin")
HIP_LIB = os.path.join(LIB_DIR, "liblfi_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "liblfi_host.so")
CLI = os.path.join(BIN_DIR, "lfInterpolator")


def build_all(force: bool = False, verbose: bool = False) -> None:
    """hipcc --offload-arch=gfx950 for the kernels, g++ for the host code; outputs stay inside the package."""
    cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building the lfinterpolator_amd native code failed (see output above)")
    for path in (HIP_LIB, HOST_LIB, CLI):
        if not os.path.exists(path):
            raise RuntimeError(f"build finished but {path} is missing")
