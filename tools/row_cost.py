#!/usr/bin/env python
"""Cost profile of the C3 frame by 64-row chunk (kernel ms for each chunk rendered alone)."""
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
for v in (0, 1):
    ctx.set_variant(v); ctx.set_schedule(2)
    out = []
    for c in range(16):
        ms, st = ctx.time_clouds(p, W, (8, c * 8, 1, 8), warmup=1, iters=5)
        out.append((ms, st["incloud_samples"] / max(1, st["primary_samples"])))
    print("variant", v, " ".join("%.3f" % m for m, _ in out), "sum %.3f" % sum(m for m, _ in out))
    print("  incloud", " ".join("%.3f" % f for _, f in out))
