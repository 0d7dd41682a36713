# Round 4: the structured scene's estimated focus map (map 0, what TEN_WM reads; pass 1 for map 1) through tools/probe_gather (reads only)
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - "${1:-0}" <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
hp = L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V)
ctx.set_params(hp); ctx.fill_synthetic_scene(0x1F1F); ctx.focus_map(); ctx.sync()
ctx.download_map(int(sys.argv[1]))[..., 0].copy().tofile("/tmp/map.bin")
np.asarray(hp.offsets, np.float32).tofile("/tmp/offsets.bin")
ctx.close()
PY
timeout -k 10 300 ./tools/probe_gather /tmp/map.bin /tmp/offsets.bin 0.22 0.17
