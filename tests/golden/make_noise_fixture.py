#!/usr/bin/env python
"""Renders tests/golden/noise_fixture.npz with the numpy restatement of the noise generators (oracle/noise_restatement.py), NOT with the
library: SHA-256 of the 128^3 RGBA stand-in shape volume (seed 1 = the default assets' volume) and of the generated 32^3 RGB detail volume,
one interior 16^3 block and one block across the wrap-around faces of the shape volume, and a second seed's hashes.  About a minute.
    python tests/golden/make_noise_fixture.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import noise_restatement as NR  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


out = {}
for seed in (1, 7):
    vol = NR.shape_volume(seed, 128)
    out["shape_sha256_seed%d" % seed] = sha(vol)
    if seed == 1:
        out["shape_block_z8_y72_x40"] = vol[8:24, 72:88, 40:56].copy()          # interior block
        out["shape_block_z120_y112_x0"] = vol[120:128, 112:128, 0:16].copy()    # touches three faces of the period
        out["shape_channel_means"] = vol.reshape(-1, 4).mean(0)
    det = NR.detail_volume(seed, 32)
    out["detail_sha256_seed%d" % seed] = sha(det)
    if seed == 1:
        out["detail_block_z0_y8_x16"] = det[0:16, 8:24, 16:32].copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "noise_fixture.npz"), **out)
for k, v in out.items():
    print(k, v if isinstance(v, str) else getattr(v, "shape", v))
