#!/usr/bin/env python
"""Could sharing the light march inside a workgroup shorten one GPU's share of a split frame?  (host analysis, no GPU)

A 1/8 share of the C3 frame is 4 096 wavefronts on 1 024 SIMDs (4 per SIMD, all resident at once), so the launch ends when its slowest
wavefront ends.  A lone wavefront takes ~0.162 ms + 6.1 us per flush of 64 in-cloud samples (profiles/r02/share_matrix.txt).  The proposal: a
per-workgroup queue, so that a heavy tile's flushes are marched by its three sibling wavefronts too.  That only helps if the siblings are
LIGHTER.  This script traces the headline frame on the CPU (tools/stage_trace: the kernel cores compiled for the host), counts the in-cloud
samples of every 8x8 tile and compares, per share, the heaviest wavefront with the heaviest workgroup under PERFECT four-way sharing."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import gvcd_amd  # noqa: E402
HERE = os.path.join(ROOT, "tools", "stage_trace")
so = os.path.join(HERE, "libstage_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", so, os.path.join(HERE, "stage_trace.cpp")])
L = C.CDLL(so)
W, H, steps, ls = 2048, 1024, 128, 6
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
large, small, weather = gvcd_amd.assets.load_default_noise()
lc, sc = gvcd_amd.assets.build_mips(large, 8), gvcd_amd.assets.build_mips(small, 6)
P = lambda a: a.ctypes.data_as(C.c_void_p)
st = np.zeros((H, W, steps), np.uint8); hist = np.zeros((7, 5), np.uint64); win = np.zeros(2, np.float32)
L.stage_trace(P(lc), P(sc), P(weather), P(p), steps, ls, W, H, P(st), P(hist), P(win))
ev = (st == 4).reshape(H // 8, 8, W // 8, 8, steps).sum(axis=(1, 3, 4))          # in-cloud samples per tile [128][256]
live = (st[..., 0] != 255).reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
a, b = 0.162, 0.0061
F = np.ceil(ev / 64.0)
print("C3 frame, sun (1,1,0)/sqrt2: in-cloud samples per tile mean %.0f, max %d; flushes per tile mean %.1f, max %d" % (ev.mean(), ev.max(), F.mean(), F.max()))
wgF = F.reshape(128, 64, 4)
print("per workgroup (4 adjacent tiles): mean of its four tiles' flushes: mean %.1f, max %.1f; the heaviest workgroup's four tiles: %s" % (
    wgF.mean(-1).mean(), wgF.mean(-1).max(), wgF.reshape(-1, 4)[wgF.mean(-1).argmax()].astype(int).tolist()))
for N in (1, 2, 4, 8, 16):
    rows = np.arange(0, 128, N)
    t = np.where(live[rows], a + b * F[rows], 0.01)
    wg = t.reshape(len(rows), 64, 4)
    solo = wg.max()
    shared = np.maximum(wg.sum(-1) / 4.0, a).max()
    print("1/%-2d share: %5d wavefronts, mean lone-wavefront time %.3f ms; heaviest wavefront %.3f ms; heaviest workgroup with perfect 4-way sharing of its light "
          "marches %.3f ms (gain %.1f %%)" % (N, t.size, t.mean(), solo, shared, 100 * (1 - shared / solo)))
print("=> neighbouring tiles are as heavy as each other (clouds are larger than 32 pixels): the workgroup that decides the launch time has four heavy tiles and")
print("   nothing to share.  Balancing would have to cross workgroups (CUs), i.e. go through global memory between XCDs with non-coherent L2s: not built.")
