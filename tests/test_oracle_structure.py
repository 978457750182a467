"""Known-answer / structural properties derivable from the GLSL text alone."""
import ctypes as C

import numpy as np

from conftest import norm


def test_half_conversion_roundtrip(oracle):
    L = oracle.lib()
    hs = np.arange(0, 0x7c00, dtype=np.uint16)                         # every finite non-negative half
    f = hs.view(np.float16).astype(np.float32)
    back = np.array([L.csko_f2h(float(v)) for v in f[::37]], np.uint16)
    assert (back == hs[::37]).all()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(4000) * 10 ** rng.uniform(-8, 5, 4000)).astype(np.float32)
    mine = np.array([L.csko_f2h(float(v)) for v in x], np.uint16)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)                     # numpy rounds to nearest even
    assert (mine == ref).all()
    assert all(L.csko_h2f(int(h)) == float(np.uint16(h).view(np.float16)) for h in hs[::91])


def test_transmittance_known_values_and_monotonicity(o_trans):
    t = o_trans.astype(np.float32)
    # SURVEY §8c self-derived fp32 probes (scratch numpy, not reference output)
    np.testing.assert_allclose(t[0, 255], [0.9031, 0.8673, 0.8308, 0.7499], atol=6e-4)
    np.testing.assert_allclose(t[0, 192], [0.8177, 0.7547, 0.6929, 0.5655], atol=6e-4)
    np.testing.assert_allclose(t[32, 128], [0.9834, 0.9808, 0.9911, 0.9947], atol=6e-4)
    assert (t[0, :65] == 0).all() and (t[63, 255] == 1).all()
    assert np.isfinite(t).all() and t.min() >= 0 and t.max() <= 1
    # more atmosphere above you -> less transmittance: monotone in altitude (rows) for sun above the horizon,
    # and monotone in sun cosine (columns) at ground level
    up = t[:, 160:, :]
    assert (np.diff(up, axis=0) >= -1e-3).all()
    assert (np.diff(t[0], axis=0) >= -1e-3).all()


def test_sky_lut_known_values(o_skies):
    z = o_skies["zenith"].astype(np.float32)
    assert 0.1 < z[..., :3].min() and 17.0 < z[..., :3].max() < 18.5
    np.testing.assert_allclose(z[99, 0, :3], [3.75, 5.49, 8.87], rtol=5e-3)
    np.testing.assert_allclose(z[50, 0, :3], [16.4, 16.6, 16.4], rtol=5e-3)
    d = o_skies["demo"].astype(np.float32)
    np.testing.assert_allclose(d[50, 0, :3], [96.4, 33.6, 2.24], rtol=5e-3)
    assert (z[..., 3] == 1).all() and np.isfinite(d).all()


def test_hash_is_identically_zero(oracle):
    """clouds.glsl:60-64,145: the start jitter hash(pos*10) vanishes for every shell entry point in fp32."""
    L = oracle.lib()
    p = oracle.default_params(256, 128, (0, 1, 0))
    d = np.zeros(3, np.float32)
    for px, py in [(1, 1), (128, 64), (255, 127), (17, 93), (200, 5), (3, 120)]:
        L.csko_pixel_dir(p.ctypes.data_as(C.c_void_p), px, py, d.ctypes.data_as(C.c_void_p))
        assert d[1] > 0
        # entry point on the 6 001 500 m sphere seen from (0, 6e6, 0)
        b = 2 * d[1] * 6.0e6
        c = 6.0e6 ** 2 - 6001500.0 ** 2
        t = (-b + np.sqrt(b * b - 4 * c)) / 2
        pos = np.array([0, 6.0e6, 0]) + d.astype(np.float64) * t
        assert L.csko_hash_probe(float(pos[0]), float(pos[1]), float(pos[2])) == 0.0


def test_pixel_directions(oracle):
    L = oracle.lib()
    p = oracle.default_params(256, 128, (0, 1, 0))
    d = np.zeros(3, np.float32)
    L.csko_pixel_dir(p.ctypes.data_as(C.c_void_p), 128, 64, d.ctypes.data_as(C.c_void_p))
    assert d.tolist() == [0.0, 1.0, 0.0]                                     # centre pixel looks at the zenith
    for px, py in [(0, 5), (7, 0), (0, 0), (255, 0), (0, 127)]:              # row 0 / column 0: n.z == 0 -> dir.y == 0
        L.csko_pixel_dir(p.ctypes.data_as(C.c_void_p), px, py, d.ctypes.data_as(C.c_void_p))
        assert d[1] == 0.0
    for px, py in [(13, 77), (200, 31)]:
        L.csko_pixel_dir(p.ctypes.data_as(C.c_void_p), px, py, d.ctypes.data_as(C.c_void_p))
        assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1) < 1e-6 and d[1] > 0


def test_cloud_image_structure(oracle, otex, o_skies):
    img, st = oracle.clouds(otex, oracle.default_params(64, 32, (0, 1, 0)), o_skies["zenith"], return_stats=True)
    f = img.astype(np.float32)
    assert (f[0] == 0).all() and (f[:, 0] == 0).all()                        # horizon row/column -> vec4(0)
    assert np.isfinite(f).all() and f[..., 3].min() >= 0 and f[..., 3].max() <= 1 and f[..., :3].min() >= 0
    assert st["rays"] == 64 * 32 and st["rays_marched"] == 63 * 31
    assert 0.05 < st["incloud_samples"] / st["primary_samples"] < 0.4
    # alpha == 0 <=> no radiance was ever accumulated
    assert ((f[..., 3] == 0) == (f[..., :3].sum(-1) == 0)).all()


def test_coverage_to_zero_gives_empty_sky(oracle, otex, o_skies):
    img = oracle.clouds(otex, oracle.default_params(32, 16, (0, 1, 0), coverage=1e-6), o_skies["zenith"])
    assert (img.astype(np.float32) == 0).all()
    # coverage exactly 0 divides by zero in remap (clouds.glsl:124): defined as density 0, never NaN
    img = oracle.clouds(otex, oracle.default_params(32, 16, (0, 1, 0), coverage=0.0), o_skies["zenith"])
    assert (img.astype(np.float32) == 0).all()


def test_more_coverage_more_alpha(oracle, otex, o_skies):
    a = [oracle.clouds(otex, oracle.default_params(48, 24, (0, 1, 0), coverage=c), o_skies["zenith"])[..., 3].astype(np.float32).mean()
         for c in (0.1, 0.2, 0.4)]
    assert a[0] < a[1] < a[2]


def test_tile_rendering_equals_full_frame(oracle, otex, o_skies):
    """clouds.glsl:260: rendering tile by tile through update_position gives the full-frame image bit for bit."""
    p = oracle.default_params(64, 32, (1, 1, 0))
    full = oracle.clouds(otex, p, o_skies["deg45"]).view(np.uint16)
    tiles = np.zeros_like(full)
    for ty in range(0, 32, 16):
        for tx in range(0, 64, 16):
            q = p.copy(); q[2:4] = (tx, ty)
            tiles[ty:ty + 16, tx:tx + 16] = oracle.clouds(otex, q, o_skies["deg45"], rect=(0, 0, 16, 16)).view(np.uint16)
    assert (tiles == full).all()


def test_density_rejects_are_exact(oracle, otex):
    """SURVEY A.7 exact empty-space test: g <= 1 - coverage*weather.b  ==>  density() == 0 (0 false rejects)."""
    import ctypes as C
    L = oracle.lib()
    p = oracle.default_params(64, 32, (0, 1, 0))
    rng = np.random.default_rng(3)
    zero = 0
    for _ in range(3000):
        pos = np.array([rng.uniform(-2e4, 2e4), 6.0e6 + rng.uniform(1500, 4000), rng.uniform(-2e4, 2e4)], np.float32)
        w = np.array([rng.uniform(0.55, 0.95), 0.0, rng.uniform(0.05, 1.0)], np.float32)
        d = L.csko_density_probe(C.byref(otex.c), p.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), 0.0)
        assert d >= 0 and np.isfinite(d)
        zero += d == 0
    assert zero > 1000
