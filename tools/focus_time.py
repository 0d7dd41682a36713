"""Time the focus-map kernels and the all-focus renders (config-5-like parameters from scripts/focusMapCompare.sh)."""
import sys, time
sys.path.insert(0, ".")
import lfinterpolator_amd as L
for (cols, rows, W, H) in ((8, 8, 1920, 1080), (15, 15, 1920, 1080), (15, 15, 3840, 2160)):
    ctx = L.Context(0); ctx.set_grid(cols, rows, W, H); ctx.fill_synthetic(0x1F1F)
    hp = L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, 64)
    ctx.set_params(hp); ctx.sync()
    msg = f"{cols}x{rows} @{W}x{H}: focus_map (estimate+filter)"
    for variant in ctx.list_variants("FOCUS"):
        ctx.set_variant("FOCUS", variant)
        ctx.focus_map(); ctx.sync()
        t0 = time.perf_counter(); ctx.focus_map(); ctx.sync(); t1 = time.perf_counter()
        msg += f" {variant} {1e3*(t1-t0):.2f} ms"
    ctx.set_variant("FOCUS", "auto")
    for method in ("TEN_WM", "STD"):
        st = ctx.benchmark(method, all_focus=True, warmup=1, runs=3)
        msg += f" | all-focus {method} {st.median_ms:.3f} ms"
        st = ctx.benchmark(method, all_focus=False, warmup=1, runs=3)
        msg += f" (fixed focus {st.median_ms:.3f} ms)"
    print(msg, flush=True)
    ctx.close()
