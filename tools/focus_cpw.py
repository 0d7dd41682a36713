"""Time the factored focus-map estimate for the LFI_FOCUS_CPW debug settings (subprocess per setting)."""
import os, subprocess, sys
for cpw in ("1", "2", "4", "8"):
    env = dict(os.environ, LFI_FOCUS_CPW=cpw)
    code = ("import sys, time; sys.path.insert(0, '.'); import lfinterpolator_amd as L\n"
            "ctx = L.Context(0); ctx.set_grid(8, 8, 1920, 1080); ctx.fill_synthetic(0x1F1F)\n"
            "ctx.set_params(L.build_params(8, 8, 1920, 1080, '0.071,0.071,0.93,0.93', 0.22, 0.17, 7.0, 1.783, 64)); ctx.sync()\n"
            "ctx.focus_map(); ctx.sync(); ts = []\n"
            "for _ in range(5):\n"
            "    t0 = time.perf_counter(); ctx.focus_map(); ctx.sync(); ts.append(time.perf_counter() - t0)\n"
            "print('cpw', %s, 'focus_map ms', ['%%.3f' %% (1e3 * t) for t in ts])\n" % cpw)
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
