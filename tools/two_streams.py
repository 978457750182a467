#!/usr/bin/env python
"""Throughput with several frames in flight: consecutive renders alternate between caller streams (the library's rings + events
keep them independent), so the tail of frame k overlaps the head of frame k+1.  Whole frame and one rank's share at N = 2, 4, 8."""
import os, sys, time
if os.environ.get("CSKY_TOOLS_NO_QUEUE_ENV") != "1":   # =1: leave it to libcloudsky's load-time default (api.cpp::csky_runtime_defaults)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # frame streams must land on different hardware queues (see bench.py)
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


def run(share, nstreams, K=100):
    bands = (8, 0, share, H // 8 // share)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(nstreams)]

    def step(k):
        i = k % nstreams
        ctx.render_sky_lut_device(s, 200, 100, streams[i].cuda_stream)
        ctx.render_clouds_device(p, W, bands, outs[i].data_ptr(), W * 8, streams[i].cuda_stream)

    for k in range(12):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / K * 1e3
    a = outs[0].view(torch.float16).float()
    same = all(bool(((o.view(torch.float16).float() - a).abs() <= 5e-4 + 2e-3 * a.abs()).all().item()) for o in outs)
    print("1/%d frame, %d stream(s): %.3f ms per frame, frames agree: %s" % (share, nstreams, per, same), flush=True)


for sched in ([int(a) for a in sys.argv[1:]] or [-1]):
    ctx.set_schedule(sched)
    print("schedule %d" % sched)
    for share in (1, 2, 4, 8):
        for nstreams in (1, 2, 3):
            run(share, nstreams)
