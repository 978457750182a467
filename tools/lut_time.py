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

# parity of the two LUTs against the CPU oracle (worst fp16 ulp over a sun sweep): what a cheaper transcendental mix costs in exactness
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from conftest import ulp_diff, norm
tro = O.transmittance_lut(256, 64)
worst, nonzero = 0, 0.0
for th in list(np.linspace(-20, 200, 23)) + [45.0, 90.0]:
    for z in (0.0, 0.1, -0.4):
        sun = norm((np.cos(np.radians(th)), np.sin(np.radians(th)), z))
        d = ulp_diff(ctx.render_sky_lut(sun, 200, 100), O.sky_lut(sun, tro))
        worst = max(worst, int(d.max())); nonzero = max(nonzero, float((d > 0).mean()))
print("%s: transmittance LUT max ulp %d; sky LUT worst ulp %d over 75 suns, worst share of texels off by >= 1 ulp %.4f" % (
    os.path.basename(gvcd_amd.library_path()), int(ulp_diff(ctx.render_transmittance(256, 64), tro).max()), worst, nonzero))
