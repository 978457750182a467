#!/usr/bin/env python
"""Occupancy over time of one cloud-kernel launch, from per-wavefront start/end timestamps.  Needs the analysis build:
    make -C godot-volumetric-cloud-demo-v2_amd/csrc timeline
    CSKY_LIBRARY=$PWD/godot-volumetric-cloud-demo-v2_amd/libcloudsky_timeline.so python tools/timeline.py [1/N of the frame] [schedule]
(csky_time_clouds of that build dumps [t0, t1, xcc<<32|hw_id, workgroup<<32|samples] per wavefront to $CSKY_TIMELINE).
Prints the active-wavefront profile, per-XCD finish times and how wavefront duration tracks in-cloud samples."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
out = os.path.join(ROOT, "gpurun_out", "timeline.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["CSKY_TIMELINE"] = out
W, H = 2048, 1024
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sched = int(sys.argv[2]) if len(sys.argv) > 2 else -1
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.render_sky_lut(s, 200, 100, readback=False)
ctx.set_schedule(sched)
ms, st = ctx.time_clouds(p, W, (8, 0, nb, H // 8 // nb), warmup=2, iters=5)
d = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
d = d[d[:, 1] > 0]
t0 = d[:, 0].astype(np.int64); t1 = d[:, 1].astype(np.int64)
base = t0.min(); t0 -= base; t1 -= base
tick = 1e-5            # wall_clock64: 100 MHz -> ms per tick
xcc = (d[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
ev = (d[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
end = t1.max()
print("launch %.3f ms by events; %d wavefronts; span %.3f ms; mean wave duration %.3f ms (min %.3f max %.3f)" % (
    ms, len(d), end * tick, (t1 - t0).mean() * tick, (t1 - t0).min() * tick, (t1 - t0).max() * tick))
edges = np.linspace(0, end, 41)
act = np.zeros(40)
for i in range(40):
    a, b = edges[i], edges[i + 1]
    act[i] = (np.clip(t1, a, b) - np.clip(t0, a, b)).sum() / (b - a)
print("active wavefronts per 1/40 of the span (capacity 8192):")
print(" ".join("%4d" % v for v in act))
full = act.max()
print("time-integral of occupancy = %.1f %% of (peak occupancy x span)" % (100 * act.mean() / full))
for x in range(8):
    m = xcc == x
    if m.any():
        print("XCD %d: %5d waves, events %8d, busy wave-ms %.1f, last end %.3f ms" % (x, m.sum(), ev[m].sum(), (t1[m] - t0[m]).sum() * tick, t1[m].max() * tick))
dur = (t1 - t0) * tick
order = np.argsort(t0)
last = order[-len(order) // 10:]
print("last dispatch at %.3f ms (%.0f %% of the span); the 10 %% of wavefronts that start last: start >= %.3f ms, duration mean %.3f max %.3f ms" % (
    t0.max() * tick, 100.0 * t0.max() / end, t0[last].min() * tick, dur[last].mean(), dur[last].max()))
for q in (0.5, 0.75, 0.9, 0.99):
    print("  %2.0f %% of the wavefronts have finished by %.3f ms" % (100 * q, np.quantile(t1, q) * tick))
c = np.corrcoef(dur, ev)[0, 1]
print("corr(wave duration, in-cloud events) = %.3f; duration ~ %.4f + %.6f * events ms" % (c, *np.polyfit(ev, dur, 1)[::-1]))
