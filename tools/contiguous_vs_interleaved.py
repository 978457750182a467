#!/usr/bin/env python
"""One rank's share of the C3 frame at N = 8: interleaved 8-row bands (rank r: bands r, r+8, ...) vs a CONTIGUOUS eighth (rows 128 r .. 128 r + 127,
which would make the gathered layout the frame itself and save rank 0 the interleave).  ms per frame with 1 / 2 / 4 frames in flight, every rank."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvcd_amd
W, H, N = 2048, 1024, 8
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise()); ctx.render_transmittance(256, 64)
pool = [torch.cuda.Stream() for _ in range(4)]
outs = [torch.zeros((H // N, W, 4), dtype=torch.int16, device="cuda") for _ in range(4)]
for name, mk in (("interleaved", lambda r: (8, r, N, H // 8 // N)), ("contiguous ", lambda r: (H // N, r, 1, 1))):
    for ns in (1, 2, 4):
        ctx.set_frames_in_flight(ns)
        row = []
        for r in range(N):
            bands = mk(r)
            def step(k):
                i = k % ns
                ctx.render_sky_lut_device(s, 200, 100, pool[i].cuda_stream)
                ctx.render_clouds_device(p, W, bands, outs[i].data_ptr(), W * 8, pool[i].cuda_stream)
            for k in range(12):
                step(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(60):
                step(k)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / 60 * 1e3)
        print("%s x%d: %s   max %.3f mean %.3f" % (name, ns, " ".join("%.3f" % v for v in row), max(row), sum(row) / len(row)), flush=True)
