#!/usr/bin/env python
"""Kernel time vs launch size for the segment modes (to set the auto rule)."""
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
ctx.set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else -1)
for nb in (128, 64, 32, 16, 8, 4, 2, 1):
    bands = (8, 0, nb, H // 8 // nb)
    row = []
    for seg in (1, 2, 4, 5):
        ctx.set_segments(seg)
        for sched in (2, 5, 7):
            ctx.set_schedule(sched)
            ms, _ = ctx.time_clouds(p, W, bands, warmup=1, iters=6)
            row.append("g%d/s%d %.3f" % (seg, sched, ms))
    print("%5d tiles: %s" % (256 * 128 // nb, "  ".join(row)), flush=True)
