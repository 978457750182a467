"""noise_restatement.py -- numpy-fp32 restatement of the stand-in noise generators (TEST INFRASTRUCTURE, never imported by the product).

What it restates: csrc/noise_core.h `shape_voxel` (the 128^3 RGBA stand-in for the missing cloud_sky/perlworlnoise.tga,
perlworlnoise.tga.import:24-27: 128 slices, R = Perlin-Worley, G/B/A = inverted Worley fBm, the channel roles clouds.glsl:118,122 fix) and
`detail_voxel` (a generated 32^3 RGB volume in the role of cloud_sky/worlnoise.bmp, README.md:30 TODO 3 "generate the noise on the GPU").
Those generators are this build's own definition (the original asset is absent from the reference checkout, .MISSING_LARGE_BLOBS), so there
is no reference output to pin them to; what this file gives is an INDEPENDENT second implementation -- array arithmetic in numpy, written
from the formulas, not from the C control flow -- so that the GPU bake and the host generator are no longer only compared with each other
(noise_core.h compiled for gfx950 against noise_core.h compiled for x86).  Every operation is IEEE single precision
(+, -, *, /, sqrt, floor) or 32-bit integer hashing, in the order the formulas give, so the result must agree to the BYTE.

tests/golden/make_noise_fixture.py renders the fixture (SHA-256 of both volumes + one 16^3 block) with this file; tests/test_noise_oracle.py
checks the host generator against it on CPU, tests/test_gpu_round4.py the HIP bake."""
import numpy as np

F = np.float32
U = np.uint32


def _hash_u32(x):
    """lowbias32 finaliser on uint32 arrays (wrap-around arithmetic)."""
    x = x.astype(U, copy=True)
    x ^= x >> U(16); x *= U(0x7FEB352D); x ^= x >> U(15); x *= U(0x846CA68B); x ^= x >> U(16)
    return x


def _hash_cell(x, y, z, salt):
    """hash of the lattice cell (x, y, z) (already wrapped to the period) under `salt`."""
    inner = _hash_u32(z.astype(U) * U(0xCB1AB31F) ^ U(salt))
    mid = _hash_u32(y.astype(U) * U(0xD8163841) ^ inner)
    return _hash_u32(x.astype(U) * U(0x8DA6B343) ^ mid)


def _u01(h):
    return (h >> U(8)).astype(F) * F(1.0 / 16777216.0)          # 24 bits / 2^24: exact


def worley(x, y, z, freq, salt):
    """1 - distance to the nearest feature point of a jittered lattice with `freq` cells per unit, tileable, clamped at 0."""
    px, py, pz = x * F(freq), y * F(freq), z * F(freq)
    cx, cy, cz = np.floor(px).astype(np.int64), np.floor(py).astype(np.int64), np.floor(pz).astype(np.int64)
    best = np.full(px.shape, F(1e9))
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                gx, gy, gz = cx + dx, cy + dy, cz + dz
                h = _hash_cell(np.mod(gx, freq), np.mod(gy, freq), np.mod(gz, freq), salt)
                fx = gx.astype(F) + _u01(h)
                fy = gy.astype(F) + _u01(_hash_u32(h + U(0x9E3779B9)))
                fz = gz.astype(F) + _u01(_hash_u32(h + U(0x3C6EF372)))
                ex, ey, ez = fx - px, fy - py, fz - pz
                d2 = (ex * ex + ey * ey) + ez * ez
                best = np.minimum(best, d2)
    v = F(1.0) - np.sqrt(best)
    return np.maximum(v, F(0.0))


def worley_fbm(x, y, z, freq, salt):
    return (worley(x, y, z, freq, salt) * F(0.625) + worley(x, y, z, freq * 2, (salt + 1) & 0xFFFFFFFF) * F(0.25)) + worley(x, y, z, freq * 4, (salt + 2) & 0xFFFFFFFF) * F(0.125)


_GRAD = np.array([[1, 1, 0], [-1, 1, 0], [1, -1, 0], [-1, -1, 0], [1, 0, 1], [-1, 0, 1], [1, 0, -1], [-1, 0, -1], [0, 1, 1], [0, -1, 1], [0, 1, -1], [0, -1, -1]], np.float32)


def _grad(h, x, y, z):
    """one of the 12 edge gradients dotted with (x, y, z): always a sum or difference of TWO of the coordinates"""
    k = (h % U(12)).astype(np.int64)
    g = _GRAD[k]
    # the two non-zero terms, added in the order x, y, z (a term with coefficient 0 does not take part: +-a +- b is one rounding)
    first = np.where(g[..., 0] != 0, g[..., 0] * x, g[..., 1] * y)
    second = np.where(g[..., 2] != 0, g[..., 2] * z, g[..., 1] * y)
    return (first + second).astype(F)


def _fade(t):
    return t * t * t * (t * (t * F(6.0) - F(15.0)) + F(10.0))


def _lerp(a, b, t):
    return a + (b - a) * t


def perlin(x, y, z, freq, salt):
    px, py, pz = x * F(freq), y * F(freq), z * F(freq)
    ix, iy, iz = np.floor(px).astype(np.int64), np.floor(py).astype(np.int64), np.floor(pz).astype(np.int64)
    fx, fy, fz = px - ix.astype(F), py - iy.astype(F), pz - iz.astype(F)
    u, v, w = _fade(fx), _fade(fy), _fade(fz)
    c = {}
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                h = _hash_cell(np.mod(ix + dx, freq), np.mod(iy + dy, freq), np.mod(iz + dz, freq), salt)
                c[dz, dy, dx] = _grad(h, fx - F(dx), fy - F(dy), fz - F(dz))
    lo = _lerp(_lerp(c[0, 0, 0], c[0, 0, 1], u), _lerp(c[0, 1, 0], c[0, 1, 1], u), v)
    hi = _lerp(_lerp(c[1, 0, 0], c[1, 0, 1], u), _lerp(c[1, 1, 0], c[1, 1, 1], u), v)
    return _lerp(lo, hi, w)


def perlin_fbm(x, y, z, freq, octaves, salt):
    amp, norm = F(1.0), F(0.0)
    total = np.zeros(x.shape, F)
    for o in range(octaves):
        total = total + amp * perlin(x, y, z, freq << o, (salt + 17 * o) & 0xFFFFFFFF)
        norm = F(norm + amp)
        amp = F(amp * F(0.5))
    return total / norm


def _clamp01(v):
    return np.minimum(np.maximum(v, F(0.0)), F(1.0))


def _unorm8(v):
    return (_clamp01(v) * F(255.0) + F(0.5)).astype(np.int32).astype(np.uint8)


def _centres(n, x0, x1, y0, y1, z0, z1):
    inv = F(1.0) / F(n)
    zz, yy, xx = np.meshgrid(np.arange(z0, z1), np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
    return (xx.astype(F) + F(0.5)) * inv, (yy.astype(F) + F(0.5)) * inv, (zz.astype(F) + F(0.5)) * inv


def shape_block(seed, n, x0, x1, y0, y1, z0, z1):
    """[z1-z0, y1-y0, x1-x0, 4] uint8: voxels of the stand-in shape volume (R Perlin-Worley, G/B/A Worley fBm at base frequency 4 / 8 / 16)."""
    with np.errstate(over="ignore"):
        u, v, w = _centres(n, x0, x1, y0, y1, z0, z1)
        s = (seed * 101) & 0xFFFFFFFF
        g = worley_fbm(u, v, w, 4, (s + 11) & 0xFFFFFFFF)
        b = worley_fbm(u, v, w, 8, (s + 23) & 0xFFFFFFFF)
        a = worley_fbm(u, v, w, 16, (s + 37) & 0xFFFFFFFF)
        pf = perlin_fbm(u, v, w, 4, 5, (s + 53) & 0xFFFFFFFF)
        p01 = _clamp01(pf * F(0.9) + F(0.5))
        nmin = g * F(0.55)
        pw = nmin + ((p01 - F(0.0)) / (F(1.0) - F(0.0))) * (F(1.0) - nmin)      # remap(p01, 0, 1, 0.55 g, 1): dilate towards the Worley cells
        r = _clamp01((pw - F(0.38)) * F(1.75) + F(0.32))
        return np.stack([_unorm8(r), _unorm8(g), _unorm8(b), _unorm8(a)], -1)


def detail_block(seed, n, x0, x1, y0, y1, z0, z1):
    """[.., 3] uint8: voxels of the generated detail volume (three inverted-Worley fBm channels, base frequencies 2 / 5 / 7)."""
    with np.errstate(over="ignore"):
        u, v, w = _centres(n, x0, x1, y0, y1, z0, z1)
        out = []
        for c, (freq, centre, gain) in enumerate(((2, 0.4928, 1.0), (5, 0.4795, 0.95), (7, 0.4801, 1.2))):
            salt = (seed * 211 + 7 + c) & 0xFFFFFFFF
            d = ((F(1.0) - worley(u, v, w, freq, salt)) * F(0.625) + (F(1.0) - worley(u, v, w, freq * 2, (salt + 10) & 0xFFFFFFFF)) * F(0.25)) + \
                (F(1.0) - worley(u, v, w, freq * 4, (salt + 20) & 0xFFFFFFFF)) * F(0.125)
            out.append(_unorm8(((F(1.0) - d) - F(centre)) * F(gain) + F(0.712)))
        return np.stack(out, -1)


def shape_volume(seed=1, n=128, slab=8):
    """the whole [n, n, n, 4] volume, z slabs at a time (about a minute for 128^3)"""
    return np.concatenate([shape_block(seed, n, 0, n, 0, n, z, min(n, z + slab)) for z in range(0, n, slab)], 0)


def detail_volume(seed=1, n=32):
    return detail_block(seed, n, 0, n, 0, n, 0, n)
