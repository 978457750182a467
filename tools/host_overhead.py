#!/usr/bin/env python
"""Host (CPU) time per frame spent in the two library calls of a frame, measured while the GPU queue is the bottleneck."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
ctx.set_frames_in_flight(2)
streams = [torch.cuda.Stream() for _ in range(2)]
for share in (1, 8):
    bands = (8, 0, share, H // 8 // share)
    outs = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(6):
        ctx.render_sky_lut_device(s, 200, 100, streams[k % 2].cuda_stream); ctx.render_clouds_device(p, W, bands, outs[k % 2].data_ptr(), W * 8, streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    K = 40
    t0 = time.perf_counter()
    for k in range(K):
        ctx.render_sky_lut_device(s, 200, 100, streams[k % 2].cuda_stream); ctx.render_clouds_device(p, W, bands, outs[k % 2].data_ptr(), W * 8, streams[k % 2].cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("1/%d frame: host enqueue %.1f us per frame; GPU %.3f ms per frame" % (share, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e3))
