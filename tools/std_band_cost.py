"""What does sizing STD's rounding band with the ANALYTIC bound on the matrix core's accumulation error (LFI_FLAG_STD_ANALYTIC_BAND: N·2^-15)
cost against the bound measured on gfx950 (N·2^-17)?  Fixed-focus STD at BASELINE configs 2, 3, 5 and the all-focus STD render of config 5.
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
    for name, flags in (("measured", 0), ("analytic", L.LFI_FLAG_STD_ANALYTIC_BAND)):
        ctx.set_params(hp, flags=flags)
        ctx.prepare("STD")
        res[name] = timed(ctx, lambda: ctx.render("STD"), 20 if key == 2 else 8)
        k = ctx.last_kernel_name()
    print(f"config {key} fixed focus STD ({k}): measured band {res['measured']:.4f} ms, analytic band {res['analytic']:.4f} ms ({(res['analytic'] / res['measured'] - 1) * 100:+.1f} %)", flush=True)
    if key == 5:
        ctx.fill_synthetic_scene(0x1F1F)
        ctx.set_params(hp); ctx.focus_map()
        for name, flags in (("measured", 0), ("analytic", L.LFI_FLAG_STD_ANALYTIC_BAND)):
            ctx.set_params(hp, flags=flags)
            res[name] = timed(ctx, lambda: ctx.render("STD", all_focus=True), 6)
            k = ctx.last_kernel_name()
        print(f"config 5 all-focus STD ({k}): measured band {res['measured']:.4f} ms, analytic band {res['analytic']:.4f} ms ({(res['analytic'] / res['measured'] - 1) * 100:+.1f} %)", flush=True)
    ctx.close()
