#!/usr/bin/env python
"""Launch-tail experiment: cloud-kernel ms of the C3 frame, one frame at a time, per workgroup schedule
(5 static XCD rows, 7 heaviest-first feedback, 8 deadline feedback at CSKY_TAIL_BETA) and per share of the frame."""
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
scheds = [int(a) for a in sys.argv[1:]] or [5, 7, 8]
for share in (1, 2):
    row = []
    for sch in scheds:
        ctx.set_schedule(sch); ctx.set_segments(1)
        best = min(ctx.time_clouds(p, W, (8, 0, share, H // 8 // share), warmup=3, iters=20)[0] for _ in range(3))
        row.append("s%d %.3f" % (sch, best))
    print("beta %s  1/%d frame, whole rays: %s" % (os.environ.get("CSKY_TAIL_BETA", "default"), share, "  ".join(row)), flush=True)
ctx.close()
