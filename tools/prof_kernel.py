#!/usr/bin/env python
"""Minimal driver for profiling: render N frames of a BASELINE config through the C ABI (no torch, no oracle)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd  # noqa: E402

CONFIGS = {"C2": (512, 256, 64, 4, (0.0, 1.0, 0.0)), "C3": (2048, 1024, 128, 6, (1.0, 1.0, 0.0)), "C5frame": (4096, 2048, 128, 6, (1.0, 1.0, 0.0))}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--sched", type=int, default=-1)
ap.add_argument("--early-out", type=float, default=0.0)
ap.add_argument("--coverage", type=float, default=0.2)
ap.add_argument("--time", action="store_true", help="print csky_time_clouds mean ms for every variant")
a = ap.parse_args()
W, H, ps, ls, sun = CONFIGS[a.config]
s = np.asarray(sun, np.float64)
s = (s / np.linalg.norm(s)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05,
              a.coverage, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.set_march(ps, ls)
ctx.set_early_out(a.early_out)
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
if a.time:
    L = gvcd_amd.lib()
    for v in range(L.csky_variant_count()):
        ctx.set_variant(v)
        for nb in (1, 8):
            ms, st = ctx.time_clouds(p, W, (8, 0, nb, H // 8 // nb), warmup=2, iters=a.frames)
            print("variant %d %-10s 1/%d frame: %.3f ms" % (v, L.csky_variant_name(v).decode(), nb, ms), flush=True)
else:
    ctx.set_variant(a.variant)
    ctx.set_schedule(a.sched)
    ms, st = ctx.time_clouds(p, W, (8, 0, 1, H // 8), warmup=1, iters=a.frames)
    print("variant %d: %.3f ms/launch" % (a.variant, ms))
ctx.close()
