#!/usr/bin/env python
"""A/B of alternative builds of libcloudsky (CSKY_LIBRARY): C3 cloud-kernel ms alone (best of 3 x 20 launches), ms per frame with two frames in flight,
and a hash of a 512x256 frame (layout / scheduling experiments must not change a single bit)."""
import hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvcd_amd
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
def P(w, h):
    return np.array([w, h, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
h = hashlib.sha256(ctx.render_clouds(P(512, 256)).tobytes()).hexdigest()[:12]
solo = min(ctx.time_clouds(P(W, H), W, (8, 0, 1, H // 8), warmup=2, iters=20)[0] for _ in range(3))
ctx.set_frames_in_flight(2)
streams = [torch.cuda.Stream() for _ in range(2)]
outs = [torch.zeros((H, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
def step(k):
    i = k % 2
    ctx.render_sky_lut_device(s, 200, 100, streams[i].cuda_stream)
    ctx.render_clouds_device(P(W, H), W, (8, 0, 1, H // 8), outs[i].data_ptr(), W * 8, streams[i].cuda_stream)
best = 1e9
for rep in range(3):
    for k in range(10):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        step(k)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
print("%-28s frame hash %s   kernel alone %.3f ms   two frames in flight %.3f ms/frame" % (os.path.basename(gvcd_amd.library_path()), h, solo, best), flush=True)
ctx.close()
