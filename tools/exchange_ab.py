#!/usr/bin/env python
"""exchange_ab.py [quick] -- A/B of the light-march packet exchange (csky_set_exchange 0 vs 2; csrc/exchange.h) on one MI355X.

For each mode: hashes of a 512x256 frame, the C3 frame and a rank's 1/8 share (the three must be identical across modes: who runs a
packet must not change a bit), then ms per frame for: the C3 frame one at a time (cloud kernel alone, events), two frames in flight,
and one rank's 1/2, 1/4, 1/8 share with 1 / 2 / 4 frames in flight; the exchange's counters next to each number
(published, run by helpers, waited sweeps, idle scans, TIMED-OUT SPINS: must be 0)."""
import hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gvcd_amd
quick = "quick" in sys.argv[1:]
MODES = [int(a) for a in os.environ.get("XMODES", "0,2").split(",")]
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
def P(w, h):
    return np.array([w, h, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
pool = [torch.cuda.Stream() for _ in range(4)]
def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
hashes = {}
for xm in MODES:
    ctx.set_exchange(xm)
    ctx.set_frames_in_flight(1); ctx.set_segments(0); ctx.set_schedule(-1)
    ctx.exchange_counters(1)                                   # counting ON for the hash frames only: it costs an atomic per packet on one word
    ctx.set_segments(1)
    h_small = sha(ctx.render_clouds(P(512, 256)))
    ctx.set_segments(0)
    h_c3 = sha(ctx.render_clouds(P(W, H)))
    bands = (8, 3, 8, H // 8 // 8)
    out = torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda")
    ctx.set_segments(1)                                        # the same whole-ray arithmetic in both modes (ray segments re-associate the compositing sums)
    ctx.render_clouds_device(P(W, H), W, bands, out.data_ptr(), W * 8, pool[0].cuda_stream)
    torch.cuda.synchronize()
    h_share = sha(out.cpu().numpy())
    ctx.set_segments(0)
    hashes[xm] = (h_small, h_c3, h_share)
    print("exchange %d: frame hashes 512x256 %s  C3 %s  1/8 share (whole rays) %s   counters %s" % (xm, h_small, h_c3, h_share, ctx.exchange_counters(3)), flush=True)
    # ---- C3 one frame at a time: cloud kernel alone
    solo = min(ctx.time_clouds(P(W, H), W, (8, 0, 1, H // 8), warmup=2, iters=10 if quick else 20)[0] for _ in range(2 if quick else 3))
    c_solo = ''
    # ---- C3 two frames in flight
    ctx.set_frames_in_flight(2)
    outs = [torch.zeros((H, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
    def step(k):
        i = k % 2
        ctx.render_sky_lut_device(s, 200, 100, pool[i].cuda_stream)
        ctx.render_clouds_device(P(W, H), W, (8, 0, 1, H // 8), outs[i].data_ptr(), W * 8, pool[i].cuda_stream)
    best = 1e9
    for rep in range(2 if quick else 3):
        for k in range(10):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(60):
            step(k)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
    c_two = ''
    ctx.exchange_counters(1)
    ctx.set_frames_in_flight(1)
    ctx.time_clouds(P(W, H), W, (8, 0, 1, H // 8), warmup=0, iters=4)
    c_solo = ctx.exchange_counters(1)
    ctx.set_frames_in_flight(2)
    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    c_two = ctx.exchange_counters(3)
    print("exchange %d: C3 kernel alone %.3f ms   two frames in flight %.3f ms/frame   counters of 4 frames: alone %s  two in flight %s" % (xm, solo, best, c_solo, c_two), flush=True)
    # ---- rank shares
    for share in ((8,) if quick else (2, 4, 8)):
        bands = (8, 0, share, H // 8 // share)
        so = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(4)]
        row = []
        for ns in (1, 2, 4):
            ctx.set_frames_in_flight(ns)
            def sstep(k):
                i = k % ns
                ctx.render_sky_lut_device(s, 200, 100, pool[i].cuda_stream)
                ctx.render_clouds_device(P(W, H), W, bands, so[i].data_ptr(), W * 8, pool[i].cuda_stream)
            b = 1e9
            for rep in range(2):
                for k in range(12):
                    sstep(k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(60):
                    sstep(k)
                torch.cuda.synchronize()
                b = min(b, (time.perf_counter() - t0) / 60 * 1e3)
            row.append("x%d %.3f" % (ns, b))
        ctx.exchange_counters(1)
        for k in range(4):
            sstep(k)
        torch.cuda.synchronize()
        print("exchange %d: 1/%d share (automatic policy): %s ms/frame   counters of 4 frames at x4: %s" % (xm, share, "  ".join(row), ctx.exchange_counters(3)), flush=True)
    ctx.set_frames_in_flight(1)
if len(MODES) > 1:
    same = all(hashes[m] == hashes[MODES[0]] for m in MODES)
    print("frames byte-identical across exchange modes: %s" % ("YES" if same else "NO  <-- %s" % hashes), flush=True)
ctx.close()
