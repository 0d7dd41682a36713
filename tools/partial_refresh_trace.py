"""Kernel trace of partial refreshes: one image replaced → how large are the planar_build / focus_pad launches that follow?  (run under rocprofv3 --kernel-trace)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import lfinterpolator_amd as L
cols=8; W,H,V=1920,1080,64
ctx=L.Context(0); ctx.set_grid(cols,cols,W,H); ctx.fill_synthetic(1)
hp=L.build_params(cols,cols,W,H,"0,0,1,1",0.23,0.17,3.0,1.783,V)
ctx.set_params(hp); ctx.set_output_layout("planar")
ctx.render("TEN_WM"); ctx.focus_map(); ctx.sync()
img=np.zeros((H,W,4),np.uint8)
for g in (3, int(hp.focus_map_ids[0])):
    ctx.upload_image(g, img)
    ctx.render("TEN_WM"); ctx.focus_map(); ctx.sync()
ctx.close()
