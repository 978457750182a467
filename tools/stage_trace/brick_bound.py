#!/usr/bin/env python
"""How much of exact reject (2) could a conservative per-brick (max r, min fbm) bound prove WITHOUT the shape tap?  (host analysis, no GPU)"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
import gvcd_amd  # noqa: E402
so = os.path.join(HERE, "libbrick_bound.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", so, os.path.join(HERE, "brick_bound.cpp")])
L = C.CDLL(so)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 256)
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
large, small, weather = gvcd_amd.assets.load_default_noise()
lc, sc = gvcd_amd.assets.build_mips(large, 8), gvcd_amd.assets.build_mips(small, 6)
P = lambda a: a.ctypes.data_as(C.c_void_p)
sizes = np.array([1, 2, 4, 8, 16], np.int32)
for cov in (0.2, 0.35, 0.5):
    p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, cov, 0.0], np.float32)
    out = np.zeros((len(sizes), 4), np.uint64)
    L.brick_bound(P(lc), P(sc), P(weather), P(p), 128, W, H, P(sizes), len(sizes), P(out))
    print("coverage %.2f, %dx%d: %d primary samples reach the shape tap, %.1f %% of them are rejected there (exact reject 2)" % (cov, W, H, out[0, 0], 100.0 * out[0, 1] / out[0, 0]))
    for k, B in enumerate(sizes):
        print("   bricks of %2d^3 texels (+1 apron; table %6.1f KiB at 4 B/brick): the bound proves %5.1f %% of the shape-tap samples empty = %5.1f %% of reject (2); false rejects %d" % (
            B, (128 // B) ** 3 * 4 / 1024.0, 100.0 * out[k, 2] / out[k, 0], 100.0 * out[k, 2] / max(1, out[k, 1]), out[k, 3]))
