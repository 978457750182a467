#!/usr/bin/env python
"""Is the C3 / C5 throughput gap a launch-size (tail) effect or a per-ray effect?  Renders the four 2048x1024 quadrants of the
4096x2048 frame as separate launches (C5's ray density, C3's launch size) and compares their sum with the whole C5 frame."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
def params(W, H, ux=0, uy=0):
    return np.array([W, H, ux, uy, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
ms3, st3 = ctx.time_clouds(params(2048, 1024), 2048, (8, 0, 1, 128), warmup=2, iters=10)
ms5, st5 = ctx.time_clouds(params(4096, 2048), 4096, (8, 0, 1, 256), warmup=2, iters=10)
print("C3 frame %.3f ms (in-cloud %.4f)   C5 frame %.3f ms (in-cloud %.4f)   C5/4 = %.3f" % (
    ms3, st3["incloud_samples"] / st3["primary_samples"], ms5, st5["incloud_samples"] / st5["primary_samples"], ms5 / 4))
tot = 0.0
for uy in (0, 1024):
    for ux in (0, 2048):
        ms, st = ctx.time_clouds(params(4096, 2048, ux, uy), 2048, (8, 0, 1, 128), warmup=2, iters=10)
        tot += ms
        print("quadrant (%4d,%4d): %.3f ms  in-cloud %.4f" % (ux, uy, ms, st["incloud_samples"] / max(1, st["primary_samples"])))
print("sum of quadrants %.3f ms vs whole C5 frame %.3f ms" % (tot, ms5))
