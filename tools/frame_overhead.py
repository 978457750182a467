#!/usr/bin/env python
"""Per-frame cost of everything around the cloud kernel: steps of (sky LUT + frame set-up + march [+ feedback sort]) enqueued back to
back on one stream vs the cloud kernel alone, for the whole frame and for one rank's share at N = 2, 4, 8."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvcd_amd
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
stream = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    bands = (8, 0, n, H // 8 // n)
    out = torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda")
    def step():
        ctx.render_sky_lut_device(s, 200, 100, stream)
        ctx.render_clouds_device(p, W, bands, out.data_ptr(), W * 8, stream)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / K * 1e3
    k_ms, _ = ctx.time_clouds(p, W, bands, warmup=3, iters=50)
    print("1/%d frame: %.3f ms per step, cloud kernel alone %.3f ms, around it %.0f us" % (n, per, k_ms, (per - k_ms) * 1e3), flush=True)
