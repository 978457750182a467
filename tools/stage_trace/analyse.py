#!/usr/bin/env python
"""Where do density() evaluations stop, and how full are the wavefronts at each stage?  (host analysis, no GPU)
Runs tools/stage_trace/stage_trace.cpp (the kernel cores compiled with g++) over the headline view at reduced resolution."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import gvcd_amd  # noqa: E402

so = os.path.join(HERE, "libstage_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", so, os.path.join(HERE, "stage_trace.cpp")])
L = C.CDLL(so)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
steps, ls = 128, 6
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
large, small, weather = gvcd_amd.assets.load_default_noise()
lc, sc = gvcd_amd.assets.build_mips(large, 8), gvcd_amd.assets.build_mips(small, 6)
P = lambda a: a.ctypes.data_as(C.c_void_p)
st = np.zeros((H, W, steps), np.uint8)
hist = np.zeros((7, 5), np.uint64)
win = np.zeros(2, np.float32)
L.stage_trace(P(lc), P(sc), P(weather), P(p), steps, ls, W, H, P(st), P(hist), P(win))
print("height window", win)
above = st[..., 0] != 255
tiles = st.reshape(H // 8, 8, W // 8, 8, steps).transpose(0, 2, 4, 1, 3).reshape(H // 8, W // 8, steps, 64)   # [ty][tx][step][lane]
ta = above.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(H // 8, W // 8, 64).any(-1)
t = tiles[ta]                                                                                               # wavefronts with live rays
valid = t != 255
names = ["window test", "weather tap + gradient", "shape tap", "detail tap + remap", "t > 0"]
print("primary samples: %d wavefronts x %d steps" % (t.shape[0], steps))
for k in range(1, 5):
    reach = valid & (t >= k)                                    # lanes that execute stage k's code
    any_ = reach.any(-1)
    print("  reach %-24s lanes %6.2f %%   wave-steps %6.2f %%   lane utilisation when executed %5.1f %%" % (
        names[k], 100 * reach.sum() / valid.sum(), 100 * any_.mean(), 100 * reach.sum() / max(1, any_.sum() * 64)))
print("light-march samples by stage reached (rows j = 0..5, distant):")
for j in range(7):
    r = hist[j].astype(np.float64)
    c = np.cumsum(r[::-1])[::-1] / max(1.0, r.sum())
    print("  j=%d  n=%9d   reach weather %5.1f %%  shape %5.1f %%  detail %5.1f %%  t>0 %5.1f %%" % (j, int(r.sum()), 100 * c[1], 100 * c[2], 100 * c[3], 100 * c[4]))
