#!/usr/bin/env python
"""Wall time of the LUT kernels through the device-form entry points (many launches, one sync)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
ctx = gvcd_amd.Context(0)
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
ctx.render_transmittance(256, 64)
for _ in range(20):
    ctx.render_sky_lut_device(s, 200, 100)
ctx.sync()
t0 = time.perf_counter()
N = 500
for _ in range(N):
    ctx.render_sky_lut_device(s, 200, 100)
ctx.sync()
print("sky LUT: %.1f us per launch (back to back, incl. launch overhead)" % ((time.perf_counter() - t0) / N * 1e6))
