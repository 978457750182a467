#!/usr/bin/env python
"""Writes tests/golden/clouds_64x32_deg45_rgba16f.bin: the 64x32 default-config cloud frame (sun (1,1,0)/sqrt2, 128 x 6 steps) of the
independent numpy restatement (oracle/numpy_restatement.py -> clouds_np.npz, key 'deg45') as raw little-endian RGBA16F rows, so that the
plain-C tests (tests/c_abi_check.c, tests/gdext_mock_host.c) can compare a frame rendered through the C ABI / the GDExtension shim
without Python.  Data only: 64 * 32 * 4 halfs = 16 384 bytes."""
import os
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(here, "clouds_np.npz"))
a = np.ascontiguousarray(g["deg45"], np.uint16)
assert a.shape == (32, 64, 4)
a.astype("<u2").tofile(os.path.join(here, "clouds_64x32_deg45_rgba16f.bin"))
print("wrote", a.nbytes, "bytes")
