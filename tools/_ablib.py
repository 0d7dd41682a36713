"""LFI_AB_LIB=<path to a liblfi_hip.so>: make the tools load a differently built library (A/B of kernel versions on one box).
Measurement only — the product always loads lfinterpolator_amd/lib/liblfi_hip.so."""
import os
import lfinterpolator_amd.abi as _abi

if os.environ.get("LFI_AB_LIB"):
    _abi.HIP_LIB = os.path.abspath(os.environ["LFI_AB_LIB"])
    print("library:", _abi.HIP_LIB, flush=True)
