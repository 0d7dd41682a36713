"""What do the three sizes of STD's rounding band cost?  The matrix core's accumulation error per addend budgeted at 2^-17 (asserted by the
probe test: LFI_FLAG_STD_MEASURED_BAND), at the defaults (up to 64 images: the analytic 2^-15; more: the measured 2^-17) and at the analytic 2^-15
(LFI_FLAG_STD_ANALYTIC_BAND).  Fixed-focus STD at BASELINE configs 2, 3, 5 and the all-focus STD render of config 5.
usage: python tools/std_band_cost.py"""
import sys
sys.path.insert(0, ".")
import tools._ablib  # noqa
import lfinterpolator_amd as L
CFG = {2: (8, 8, 1920, 1080, 64, 3.0), 3: (15, 15, 1920, 1080, 45, 7.0), 5: (15, 15, 3840, 2160, 64, 7.0)}
def timed(ctx, fn, n):
    for _ in range(3): fn()
    ctx.sync()
    r = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(n): fn()
        r.append(ctx.timer_stop() / n)
    return sorted(r)[1]
for key, (cols, rows, W, H, V, effect) in CFG.items():
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
    hp = L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17 if key == 5 else 0.0, effect, 1.783, V)
    res = {}
    SETTINGS = (("measured", L.LFI_FLAG_STD_MEASURED_BAND), ("default", 0), ("analytic", L.LFI_FLAG_STD_ANALYTIC_BAND))
    for name, flags in SETTINGS:
        ctx.set_params(hp, flags=flags)
        ctx.prepare("STD")
        res[name] = timed(ctx, lambda: ctx.render("STD"), 20 if key == 2 else 8)
        k = ctx.last_kernel_name()
    print(f"config {key} fixed focus STD ({k}): measured band {res['measured']:.4f} ms, default {res['default']:.4f} ms ({(res['default'] / res['measured'] - 1) * 100:+.1f} %), "
          f"analytic band {res['analytic']:.4f} ms ({(res['analytic'] / res['measured'] - 1) * 100:+.1f} %)", flush=True)
    if key == 5:
        ctx.fill_synthetic_scene(0x1F1F)
        ctx.set_params(hp); ctx.focus_map()
        for name, flags in SETTINGS:
            ctx.set_params(hp, flags=flags)
            res[name] = timed(ctx, lambda: ctx.render("STD", all_focus=True), 6)
            k = ctx.last_kernel_name()
        print(f"config 5 all-focus STD ({k}): measured band {res['measured']:.4f} ms, default {res['default']:.4f} ms ({(res['default'] / res['measured'] - 1) * 100:+.1f} %), "
              f"analytic band {res['analytic']:.4f} ms ({(res['analytic'] / res['measured'] - 1) * 100:+.1f} %)", flush=True)
    ctx.close()
