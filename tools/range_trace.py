"""Where focus_range_t's waves spend their clocks (measurement build -DFRT_TRACE=1, LFI_AB_LIB selects it): per loading wave the fetch issue, the
wait for its loads, widen + LDS stores, the barrier; per reducing wave the barrier, the reduction, the epilogue.  usage: LFI_AB_LIB=… python tools/range_trace.py"""
import os, sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import lfinterpolator_amd as L
import lfinterpolator_amd.abi as abi
import _ablib  # noqa
ctx = L.Context(0); ctx.set_grid(15, 15, 3840, 2160)
ctx.set_params(L.build_params(15, 15, 3840, 2160, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64))
ctx.fill_synthetic_scene(0x1F1F)
for _ in range(3):
    ctx.focus_map(); ctx.sync()
lib = ctypes.CDLL(abi.HIP_LIB)
buf = np.zeros(256 * 16 * 8, np.uint64)
assert lib.lfi_debug_frt_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(256, 16, 8).astype(np.float64)
NRED = int(os.environ.get("FRT_NRED", "8"))  # reducing waves of the build (12 - FRT_LWAVES)
red, ld = t[:, :NRED], t[:, NRED:12]
steps = red[:, :, 4].mean()
print(f"steps per workgroup {steps:.0f}; clocks per step (mean over 256 workgroups and the role's waves)")
print(f"  reducing waves: barrier wait {red[:, :, 0].mean() / steps:7.0f}  reduction {red[:, :, 1].mean() / steps:7.0f}  epilogue {red[:, :, 2].mean() / steps:7.0f}  | whole kernel {red[:, :, 3].mean() / steps:7.0f}")
print(f"  loading waves : fetch issue  {ld[:, :, 0].mean() / steps:7.0f}  wait for loads {ld[:, :, 1].mean() / steps:7.0f}  widen + LDS stores {ld[:, :, 2].mean() / steps:7.0f}  barrier {ld[:, :, 3].mean() / steps:7.0f}")
for wv in range(12 - NRED):
    print(f"    loader {wv}: fetch {ld[:, wv, 0].mean() / steps:6.0f} wait {ld[:, wv, 1].mean() / steps:6.0f} store {ld[:, wv, 2].mean() / steps:6.0f} barrier {ld[:, wv, 3].mean() / steps:6.0f}")
print("  reducing waves' barrier wait by wave:", " ".join(f"{red[:, wv, 0].mean() / steps:.0f}" for wv in range(NRED)))
print("  reducing waves' reduction by wave:   ", " ".join(f"{red[:, wv, 1].mean() / steps:.0f}" for wv in range(NRED)))
ctx.close()
