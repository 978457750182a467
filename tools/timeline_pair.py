#!/usr/bin/env python
"""Occupancy with TWO frames in flight (the headline regime), from per-wavefront timestamps of K consecutive launches on two streams.
Needs the analysis build:  make -C godot-volumetric-cloud-demo-v2_amd/csrc timeline
  CSKY_LIBRARY=.../libcloudsky_timeline.so [CSKY_PERSISTENT=0|1] python tools/timeline_pair.py [K=12] [1/N of the frame]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
out = os.path.join(ROOT, "gpurun_out", "timeline.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
os.environ["CSKY_TIMELINE"] = out
os.environ["CSKY_TIMELINE_PAIR"] = str(K)
W, H = 2048, 1024
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
ctx.time_clouds(p, W, (8, 0, nb, H // 8 // nb), warmup=1, iters=2)
raw = np.fromfile(out + ".pair", dtype=np.uint64)
K, per = int(raw[0]), int(raw[1])
raw = raw[2:].reshape(K, per)
tick = 1e-5                                                    # 100 MHz -> ms
frames = []
for k in range(K):
    d = raw[k, 2:].reshape(-1, 4)
    d = d[d[:, 1] > 0]
    frames.append((d[:, 0].astype(np.int64), d[:, 1].astype(np.int64), (d[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf))
base = min(f[0].min() for f in frames)
st = np.array([(f[0].min() - base) * tick for f in frames]); en = np.array([(f[1].max() - base) * tick for f in frames])
print("frame: first wavefront start / last wavefront end (ms), tiles")
for k in range(K):
    print("  %2d  %.3f  %.3f   span %.3f   %d" % (k, st[k], en[k], en[k] - st[k], len(frames[k][0])))
per_frame = np.diff(en[2:-1]).mean()
print("steady state: one frame completes every %.3f ms" % per_frame)
a, b = en[2], en[K - 2]                                        # window between completions of frame 2 and frame K-2: whole periods
t0 = np.concatenate([f[0] for f in frames]) - base; t1 = np.concatenate([f[1] for f in frames]) - base
xc = np.concatenate([f[2] for f in frames])
A, B = a / tick, b / tick
busy = (np.clip(t1, A, B) - np.clip(t0, A, B)).sum() * tick
print("window %.3f .. %.3f ms: mean active wavefronts %.0f of 7168 slots (%.1f %%)" % (a, b, busy / (b - a), 100 * busy / (b - a) / 7168))
for x in range(8):
    m = xc == x
    bx = (np.clip(t1[m], A, B) - np.clip(t0[m], A, B)).sum() * tick
    print("  XCD %d: mean active %.0f of 896" % (x, bx / (b - a)))
edges = np.linspace(A, B, 61)
act = [(np.clip(t1, edges[i], edges[i + 1]) - np.clip(t0, edges[i], edges[i + 1])).sum() / (edges[i + 1] - edges[i]) for i in range(60)]
print("active wavefronts over the window (60 bins):")
print(" ".join("%4d" % v for v in act))
dur = (t1 - t0) * tick
print("tile duration: mean %.3f ms, busy wavefront-ms per frame %.0f" % (dur.mean(), dur.sum() / K))
