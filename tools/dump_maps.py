"""Config 5 on the structured scene: the two focus maps (and the parameters that made them) to gpurun_out/maps_c5.npz, for offline study of how
coherent the all-focus gathers are (round 4).  usage: python tools/dump_maps.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
cols = rows = 15; W, H, V = 3840, 2160, 64
ctx = L.Context(0); ctx.set_grid(cols, rows, W, H)
hp = L.build_params(cols, rows, W, H, "0.071,0.071,0.93,0.93", 0.22, 0.17, 7.0, 1.783, V)
ctx.set_params(hp)
ctx.fill_synthetic_scene(0x1F1F)
ctx.focus_map()
ctx.sync()
m0 = ctx.download_map(0)[..., 0].copy(); m1 = ctx.download_map(1)[..., 0].copy()
ctx.close()
np.savez_compressed("gpurun_out/maps_c5.npz", m0=m0, m1=m1, offsets=np.asarray(hp.offsets), focus=hp.focus, range=hp.range)
print("saved", m0.shape, len(np.unique(m0)), len(np.unique(m1)))
