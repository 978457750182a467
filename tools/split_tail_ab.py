#!/usr/bin/env python
"""Round 6, VERDICT r5 item 7 (latency mode): C3 frames strictly ONE AT A TIME (sky LUT + set-up + march per frame, nothing overlapped), ms per frame and a
frame hash, for whatever launch policy the environment selects (CSKY_SPLIT_PCT / CSKY_SPLIT_PRIO = the two-launch experiment of api.cpp::clouds_dev;
argv[1] = schedule mode, -1 = automatic)."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvcd_amd
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
P = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else -1
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.set_frames_in_flight(1)
ctx.set_schedule(mode)
st = torch.cuda.Stream()
out = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
def step():
    ctx.render_sky_lut_device(s, 200, 100, st.cuda_stream)
    ctx.render_clouds_device(P, W, (8, 0, 1, H // 8), out.data_ptr(), W * 8, st.cuda_stream)
for _ in range(80):
    step()
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / 100 * 1e3)
h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
print("sched %2d  split %3s %% prio %s   one frame at a time %.3f ms (runs %s)   frame hash %s" % (mode, os.environ.get("CSKY_SPLIT_PCT", "0"), os.environ.get("CSKY_SPLIT_PRIO", "0"),
      min(best), " ".join("%.3f" % b for b in best), h), flush=True)
ctx.close()
