"""Launch each TEN_WM variant a few times at config 2 (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys
sys.path.insert(0, ".")
import lfinterpolator_amd as L
names = sys.argv[1].split(",") if len(sys.argv) > 1 else None
method = sys.argv[2] if len(sys.argv) > 2 else "TEN_WM"
ctx = L.Context(0)
ctx.set_grid(8, 8, 1920, 1080); ctx.fill_synthetic(0x1F1F); ctx.sync()
ctx.set_params(L.build_params(8, 8, 1920, 1080, "0,0,1,1", 0.23, 0.0, 3.0, 1.783, 64))
for name in (names or ctx.list_variants(method)):
    ctx.set_variant(method, name)
    for _ in range(3):
        ctx.render(method)
    ctx.sync()
ctx.close()
