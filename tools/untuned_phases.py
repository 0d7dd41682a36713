"""What do untuned per-image phases of the planar copy cost?  Config 5 / config 2, planar views: render at the focus the copy was tuned for
(lfi_prepare), then at neighbouring focus values WITHOUT re-tuning (a sweep: new offsets every step), then re-tuned.
usage: python tools/untuned_phases.py"""
import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
for name, cols, W, H, V, traj, effect in (("config 2", 8, 1920, 1080, 64, "0.0,0.0,1.0,1.0", 3.0), ("config 5", 15, 3840, 2160, 64, "0.071,0.071,0.93,0.93", 7.0)):
    ctx = L.Context(0); ctx.set_grid(cols, cols, W, H); ctx.fill_synthetic(0x1F1F)
    ctx.set_output_layout("planar")
    def t(n=8):
        for _ in range(2): ctx.render("TEN_WM")
        ctx.sync()
        r = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(n): ctx.render("TEN_WM")
            r.append(ctx.timer_stop() / n)
        return sorted(r)[1]
    hp = lambda f: L.build_params(cols, cols, W, H, traj, f, 0.0, effect, 1.783, V)
    ctx.set_params(hp(0.22)); ctx.prepare("TEN_WM")
    line = [f"tuned at 0.22: {t():.4f}"]
    for f in (0.221, 0.225, 0.23, 0.24):
        ctx.set_params(hp(f))
        # two timed launches only per value would re-tune on the third: time exactly two launches after one warm launch
        ctx.render("TEN_WM"); ctx.sync()
        ctx.timer_start(); ctx.render("TEN_WM"); ms = ctx.timer_stop()
        line.append(f"untuned at {f}: {ms:.4f}")
    ctx.prepare("TEN_WM")
    line.append(f"re-tuned at 0.24: {t():.4f}")
    print(name + ": " + "; ".join(line) + " ms", flush=True)
    ctx.close()
