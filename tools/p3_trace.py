"""Where a blend_p3 workgroup's time goes: clocks wave 0 spends waiting (vmcnt wait + barrier), issuing the next fetch, in the k-loop, in the
one-pass drain wait and in the epilogue, from a measurement build (-DLFI_P3_TRACE=1; LFI_AB_LIB selects it).
usage: LFI_AB_LIB=… python tools/p3_trace.py [configs=2,4r,4,3,5]"""
import sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import lfinterpolator_amd as L
import lfinterpolator_amd.abi as abi
import _ablib  # noqa
CFG = {"2": (8, 8, 1920, 1080, 64, "0,0,1,1", 0.23, 1.783, 3.0), "3": (15, 15, 1920, 1080, 45, "0,0.5,1,0.5", 0.06, 2.276, 3.0),
       "4r": (8, 8, 3840, 2160, 32, "0,0,1,1", 0.23, 1.783, 3.0), "4": (8, 8, 3840, 2160, 256, "0,0,1,1", 0.23, 1.783, 3.0),
       "5": (15, 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 0.22, 1.783, 7.0)}
lib = None
for name in (sys.argv[1] if len(sys.argv) > 1 else "2,4r,4,3,5").split(","):
    cols, rows, W, H, V, traj, focus, aspect, effect = CFG[name]
    ctx = L.Context(0)
    ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_params(L.build_params(cols, rows, W, H, traj, focus, 0.0, effect, aspect, V))
    ctx.set_output_layout("planar")
    for _ in range(4): ctx.render("TEN_WM")
    ctx.sync()
    ms = ctx.benchmark("TEN_WM", warmup=2, runs=8).back_to_back_ms
    ctx.render("TEN_WM"); ctx.sync()
    lib = lib or ctypes.CDLL(abi.HIP_LIB)
    buf = np.zeros(1024 * 8, np.uint64)
    assert lib.lfi_debug_p3_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    t = buf.reshape(1024, 8).astype(np.float64)
    g = int(t[0, 7]); t = t[:g]
    u = t[:, 5].mean()
    print(f"config {name}: {ctx.last_kernel_name()} {ms:.4f} ms (traced build); {g} workgroups, {u:.1f} units each; clocks per unit (wave 0, mean): "
          f"wait {t[:, 0].mean() / u:.0f}, fetch issue + lookup {t[:, 1].mean() / u:.0f}, k-loop(s) {t[:, 2].mean() / u:.0f}, drain wait {t[:, 3].mean() / u:.0f}, "
          f"epilogue(s) {t[:, 4].mean() / u:.0f}; whole kernel {t[:, 6].mean() / u:.0f} per unit", flush=True)
    ctx.close()
