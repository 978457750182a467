#!/usr/bin/env python
"""Kernel time vs number of light steps (splits primary march cost from light-march cost)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
for v in (3, 1, 0):
    ctx.set_variant(v)
    for ls in (0, 1, 2, 3, 4, 5, 6):
        ctx.set_march(128, ls)
        ms, st = ctx.time_clouds(p, W, (8, 0, 1, H // 8), warmup=2, iters=10)
        print("variant %d light_steps %d: %.3f ms" % (v, ls, ms), flush=True)
cov = p.copy()
ctx.set_variant(-1); ctx.set_march(128, 6)
for c in (1e-6, 0.1, 0.2, 0.3, 0.5):
    cov[26] = c
    ms, st = ctx.time_clouds(cov, W, (8, 0, 1, H // 8), warmup=2, iters=10)
    print("coverage %.2g: %.3f ms  incloud %.4f" % (c, ms, st["incloud_samples"] / max(1, st["primary_samples"])), flush=True)
