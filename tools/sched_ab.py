#!/usr/bin/env python
"""Whole C3 frame under each workgroup schedule: ms per frame one at a time and with two frames in flight (plain and persistent forms)."""
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
ctx.set_noise(*gvcd_amd.assets.load_default_noise()); ctx.render_transmittance(256, 64)
pool = [torch.cuda.Stream() for _ in range(2)]
outs = [torch.zeros((H, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
bands = (8, 0, 1, H // 8)
ctx.set_segments(1)
for rep in range(2):
    for sched in (5, 7):   # (8 = slab rows heaviest first was measured here in round 3 and removed: profiles/r03/tail_row_lpt_ab.txt)
        row = []
        for ns in (1, 2):
            ctx.set_schedule(sched); ctx.set_frames_in_flight(ns)
            def step(k):
                i = k % ns
                ctx.render_sky_lut_device(s, 200, 100, pool[i].cuda_stream)
                ctx.render_clouds_device(p, W, bands, outs[i].data_ptr(), W * 8, pool[i].cuda_stream)
            for k in range(12):
                step(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(100):
                step(k)
            torch.cuda.synchronize()
            row.append("x%d %.3f" % (ns, (time.perf_counter() - t0) / 100 * 1e3))
        print("%s sched %d: %s" % (os.environ.get("TAG", ""), sched, "  ".join(row)), flush=True)
