"""do the 192 input and 192 output plane streams of config 2 collide on HBM channels / stacks?  A measurement build
(-DLFI_MEASUREMENT_BUILD) reads LFI_PLANAR_EXTRA_PITCH (bytes added to every plane row of the derived copy: the plane stride moves by
1080 × that, every row start by a multiple of it); the views' pitch follows the width, so the output planes are skewed by rendering
a narrower / wider image (W = 1920 ± 16·k changes the view plane stride by 1080·16·k bytes).  One process per setting.
usage: LFI_AB_LIB=<a library built with make HIPFLAGS+=-DLFI_MEASUREMENT_BUILD> LFI_PLANAR_EXTRA_PITCH=n python tools/plane_skew.py [W]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import _ablib  # noqa: F401
import lfinterpolator_amd as L
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H, V = 1080, 64
ctx = L.Context(0); ctx.set_grid(8, 8, W, H); ctx.fill_synthetic(0x1F1F)
hp = L.build_params(8, 8, W, H, "0.0,0.0,1.0,1.0", 0.23, 0.0, 3.0, 1.783, V)
ctx.set_params(hp); ctx.set_output_layout("planar"); ctx.prepare("TEN_WM")
for _ in range(300): ctx.render("TEN_WM")
ctx.sync()
res = []
for _ in range(5):
    ctx.timer_start()
    for _ in range(50): ctx.render("TEN_WM")
    res.append(ctx.timer_stop() / 50)
mem = ctx.memory_info()
mb = 3.0 * W * H * 128 / 1e6
t = sorted(res)[2]
print(f"extra pitch {os.environ.get('LFI_PLANAR_EXTRA_PITCH', '0'):>5s} W {W}: {t*1e3:.1f} us  {mb / t / 1e3:.0f} GB/s moved  (copy {mem.derived_bytes/1e6:.0f} MB)", flush=True)
ctx.close()
