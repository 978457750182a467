#!/usr/bin/env python
"""Golden vectors for csky_decode_bc7: random BC7 blocks of every mode, decoded by Pillow's "bcn" decoder (an implementation independent of
this repository).  Writes tests/golden/bc7_vectors.npz: blocks [8, N, 16] uint8 (mode m in row m) and pixels [8, N, 16, 4] uint8.
Run: python tests/golden/make_bc7_vectors.py  (needs Pillow)."""
import os
import numpy as np
from PIL import Image

N = 96
rng = np.random.default_rng(20260927)
blocks = rng.integers(0, 256, size=(8, N, 16), dtype=np.uint8)
for m in range(8):
    blocks[m, :, 0] = (blocks[m, :, 0] & ~np.uint8((1 << (m + 1)) - 1)) | np.uint8(1 << m)     # m zero bits, then a one
pixels = np.zeros((8, N, 16, 4), np.uint8)
for m in range(8):
    for i in range(N):
        im = Image.frombytes("RGBA", (4, 4), blocks[m, i].tobytes(), "bcn", (7,))
        pixels[m, i] = np.asarray(im).reshape(16, 4)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bc7_vectors.npz")
np.savez_compressed(out, blocks=blocks, pixels=pixels)
print("wrote", out, os.path.getsize(out), "bytes")
