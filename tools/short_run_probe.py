#!/usr/bin/env python
"""short_run_probe.py [idle_seconds] -- why the driver's bench line reads 4-5 % slower than the builder's: per-frame completion times of a 20-step timed region.

The driver runs `bench.py --gpus 1 --steps 20 --warmup 5`; the builder's numbers are 200-step runs.  The timed region starts with an EMPTY two-deep pipeline
(barrier + synchronize in front of it, as the contract asks): the first frame has the chip to itself and completes after ~2.3 ms (a solo march + its serial
prologue: sky LUT -> frame set-up -> march), the second fills its tail, and from the fourth frame on one frame completes every ~1.64 ms.  One fill of ~0.7-1 ms
is 0.2 % of 200 steps and 2-3 % of 20; the first two or three frames after an idle period also run a few per cent slow.  Steady state is the same on both boxes."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gvcd_amd
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0); ctx.set_noise(*gvcd_amd.assets.load_default_noise()); ctx.render_transmittance(256, 64)
ctx.set_frames_in_flight(2)
st = [torch.cuda.Stream() for _ in range(2)]
out = [torch.zeros((H, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
def step(k, evs=None):
    i = k % 2
    ctx.render_sky_lut_device(s, 200, 100, st[i].cuda_stream)
    ctx.render_clouds_device(p, W, (8, 0, 1, H // 8), out[i].data_ptr(), W * 8, st[i].cuda_stream)
    if evs is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(st[i]); evs.append(e)
for rep in range(3):
    for k in range(5): step(k)
    torch.cuda.synchronize()
    if len(sys.argv) > 1: time.sleep(float(sys.argv[1]))
    e0 = torch.cuda.Event(enable_timing=True); e0.record(st[0]); 
    t0 = time.perf_counter(); evs = []
    for k in range(20): step(k, evs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ts = [e0.elapsed_time(e) for e in evs]
    print("wall %.3f ms/frame; frame completion times (ms): %s" % ((t1 - t0) / 20 * 1e3, " ".join("%.2f" % t for t in ts)))
    print("   deltas: " + " ".join("%.2f" % (b - a) for a, b in zip([0] + ts[:-1], ts)))
