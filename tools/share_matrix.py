#!/usr/bin/env python
"""One rank's share of the frame (1/4, 1/8): ms per frame for ray segments x schedule x frames in flight.
LUT=rows (default for shares < 1: the rank's rows of the sky LUT, as bench.py does at N > 1) | whole (every rank the whole LUT, rounds 1-3) |
none (no per-frame LUT at all: what the LUT costs a share)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
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
ctx.render_sky_lut(s, 200, 100, readback=False)
LUT = os.environ.get("LUT", "rows")
lut_out = torch.zeros(200 * 100 * 8, dtype=torch.uint8, device="cuda")
NS = [int(a) for a in os.environ.get("NS", "1,2").split(",")]   # frames in flight to try (the rings are four deep)
pool = [torch.cuda.Stream() for _ in range(max(NS))]
for share in [int(a) for a in sys.argv[1:]] or (2, 4, 8):
    bands = (8, 0, share, H // 8 // share)
    outs = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(max(NS))]
    for seg in (0, 1, 2, 4):                                   # 0 = the library's automatic choice
        row = []
        for sched in ((-1,) if seg == 0 else (5, 7)):
            for ns in NS:
                ctx.set_segments(seg); ctx.set_schedule(sched); ctx.set_frames_in_flight(ns)
                def step(k):
                    i = k % ns
                    if LUT == "rows" and share > 1:     # the rank's rows of the LUT, into the buffer that rides with its bands (bench.py at N > 1)
                        ctx.render_sky_lut_rows_device(s, 0, share, lut_out.data_ptr(), lut_out.numel(), 200, 100, pool[i].cuda_stream)
                    elif LUT != "none":
                        ctx.render_sky_lut_device(s, 200, 100, pool[i].cuda_stream)
                    ctx.render_clouds_device(p, W, bands, outs[i].data_ptr(), W * 8, pool[i].cuda_stream)
                for k in range(12):
                    step(k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(60):
                    step(k)
                torch.cuda.synchronize()
                row.append("s%d/x%d %.3f" % (sched, ns, (time.perf_counter() - t0) / 60 * 1e3))
        print("%sLUT %s, 1/%d frame, seg %d: %s" % (os.environ.get("TAG", ""), LUT if share > 1 or LUT == "none" else "whole", share, seg, "  ".join(row)), flush=True)
