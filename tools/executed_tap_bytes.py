#!/usr/bin/env python
"""executed_tap_bytes.py [W H] [--out FILE] -- the texture bytes the cloud kernel's lanes actually REQUEST per frame (host trace, no GPU).

SURVEY 8(d)'s contractual figure prices every sample at 80 B (4 + 8 + 8 RGBA8 texels) against the HBM peak and comes out above 1: the taps are
served by L1 / L2 / Infinity Cache, AND the exact rejects skip most of them.  This tool counts the second part exactly: the kernel cores
(cloud_core.h, compiled for the host by tools/stage_trace) walk every ray of the frame with the bench's parameters and record how far each
density() evaluation gets; a lane that reaches a tap requests that tap's cell -- 16 B (weather xy cell of r and b), 32 B (shape xyz cells of r
and the fBm numerator), 16 B (detail xyz cell).  Primary samples fetch lazily (a cell only when its stage is reached); light samples fetch all
three cells of an in-window sample together (sample_density_eager; LOD 5 of the detail volume is one texel and needs no tap).
Output: JSON with the counts, the executed bytes and the ratio to the algorithmic 80 B/sample; bench.py quotes it next to `hbm_algorithmic`."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gvcd_amd  # noqa: E402
import pmc_collect  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="*", type=int, default=[2048, 1024])
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "executed_tap_bytes_C3.json"))
a = ap.parse_args()
W, H = a.size
steps, ls = 128, 6
so = os.path.join(HERE, "stage_trace", "libstage_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", so, os.path.join(HERE, "stage_trace", "stage_trace.cpp")])
L = C.CDLL(so)
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
large, small, weather = gvcd_amd.assets.load_default_noise()
lc, sc = gvcd_amd.assets.build_mips(large, 8), gvcd_amd.assets.build_mips(small, 6)
P = lambda x: x.ctypes.data_as(C.c_void_p)
st = np.zeros((H, W, steps), np.uint8)
hist = np.zeros((7, 5), np.uint64)
win = np.zeros(2, np.float32)
L.stage_trace(P(lc), P(sc), P(weather), P(p), steps, ls, W, H, P(st), P(hist), P(win))
valid = st != 255                                               # samples of rays above the horizon
prim = {"samples": int(valid.sum()), "weather": int((valid & (st >= 1)).sum()), "shape": int((valid & (st >= 2)).sum()), "detail": int((valid & (st >= 3)).sum()),
        "in_cloud": int((valid & (st >= 4)).sum())}
prim_bytes = 16 * prim["weather"] + 32 * prim["shape"] + 16 * prim["detail"]
light = {"samples": int(hist.sum()), "per_step": []}
light_bytes = 0
for j in range(7):
    r = hist[j].astype(np.int64)
    in_window = int(r[1:].sum())                                # reached the weather tap = inside the height window: the eager form fetches every cell then
    detail_tap = j < 5                                          # LOD = j for the cone samples, 5 (one texel, no tap) for j = 5 and the distant sample
    b = in_window * (16 + 32 + (16 if detail_tap else 0))
    light["per_step"].append({"j": j, "samples": int(r.sum()), "in_window": in_window, "bytes": b})
    light_bytes += b
algo = 80 * (prim["samples"] + light["samples"])
out = {"config": "C3 %dx%d, %d x %d steps, sun 45 degrees, default textures, wind frozen" % (W, H, steps, ls), "source_hash": pmc_collect.source_hash(),
       "cell_bytes": {"weather": 16, "shape": 32, "detail": 16}, "height_window": [float(win[0]), float(win[1])],
       "primary": dict(prim, bytes=prim_bytes), "light": dict(light, bytes=light_bytes),
       "executed_tap_bytes": prim_bytes + light_bytes, "algorithmic_bytes_80_per_sample": algo, "executed_over_algorithmic": (prim_bytes + light_bytes) / algo,
       "note": "lane-level requests of the kernel's own reject logic (host walk of cloud_core.h over every ray); what L1 / L2 / Infinity Cache then serve is the counters' business"}
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("executed_tap_bytes", "algorithmic_bytes_80_per_sample", "executed_over_algorithmic")}))
print("primary", out["primary"]); print("light bytes", light_bytes)
