#!/usr/bin/env python
"""Latency of tiny launches: a single 8-row band at different image rows, and single tiles (critical path of one wavefront)."""
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
ctx.set_variant(-1)
for seg in (1, 4, 5):
    ctx.set_segments(seg)
    out = []
    for band in (0, 16, 32, 48, 64, 80, 96, 112, 127):
        ms, st = ctx.time_clouds(p, W, (8, band, 1, 1), warmup=1, iters=5)
        out.append("%d:%.3f(%.2f)" % (band, ms, st["incloud_samples"] / max(1, st["primary_samples"])))
    print("seg", seg, "single band (256 tiles):", " ".join(out), flush=True)
# single 8x8 tiles via update_position
for seg in (1, 5):
    ctx.set_segments(seg)
    out = []
    for (tx, ty) in ((128, 0), (30, 1), (128, 20), (60, 40), (128, 64), (200, 90), (10, 120), (250, 127)):
        q = p.copy(); q[2:4] = (tx * 8, ty * 8)
        ms, st = ctx.time_clouds(q, 8, (8, 0, 1, 1), warmup=1, iters=5)
        out.append("(%d,%d):%.3f ev%d" % (tx, ty, ms, st["incloud_samples"]))
    print("seg", seg, "single tile:", " ".join(out), flush=True)
