"""Where a blend_stdx workgroup's time goes: per unit slot, the clocks wave 0 spends waiting (vmcnt wait + barrier) and working, from a
measurement build (-DLFI_SX_TRACE=1, see blend_stdx.hpp; LFI_AB_LIB selects it).   usage: LFI_AB_LIB=… python tools/stdx_trace.py"""
import sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import lfinterpolator_amd as L
import lfinterpolator_amd.abi as abi
import _ablib  # noqa
cols, rows, W, H, V, traj, focus, aspect, effect = 15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0
ctx = L.Context(0)
ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
for _ in range(3): ctx.render("STD")
ctx.sync()
print(ctx.last_kernel_name(), f"{ctx.benchmark('STD', warmup=1, runs=4).back_to_back_ms:.3f} ms")
ctx.render("STD"); ctx.sync()
lib = ctypes.CDLL(abi.HIP_LIB)
buf = np.zeros(512 * 32, np.uint64)
assert lib.lfi_debug_sx_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(512, 32).astype(np.float64)
tiles = (W // 128) * H / 512
names = ["M3", "M2", "M1", "MC0", "C1", "C2", "C3"]
print(f"tiles per workgroup {tiles:.1f}; clocks per tile and workgroup (mean over workgroups; s_memtime ticks): total {t[:, 14].mean() / tiles:.0f}; queued sums per wave-0 tile {t[:, 15].mean() / tiles:.1f}")
for i, nme in enumerate(names):
    print(f"  {nme:4s} wait {t[:, 2 * i].mean() / tiles:8.0f}   work {t[:, 2 * i + 1].mean() / tiles:8.0f}")
print(f"  sum  wait {t[:, 0:14:2].sum(1).mean() / tiles:8.0f}   work {t[:, 1:14:2].sum(1).mean() / tiles:8.0f}")
m = lambda k: t[:, k].mean() / tiles
print(f"  wave 0's tiles: more than 64 queued {m(16):.3f}, none {m(17):.3f}, queue full {m(18):.4f}")
ctx.close()
