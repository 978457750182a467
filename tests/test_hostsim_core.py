"""The kernel cores (csrc/cloud_core.h, lut_core.h: the per-lane code the HIP kernels instantiate), compiled for the
host by tests/hostsim, checked against the oracle on a CPU: kernel maths, texture baking and band addressing.
This is a unit test of device code, not a render path: libcloudsky itself has no CPU implementation."""
import ctypes as C

import numpy as np

from conftest import SUNS, cloud_close, norm, ulp_diff


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def hs_clouds(hostsim, pkg, noise, params, sky, tile_w, bands, primary=128, light=6, eps=0.0, window=True, ret_window=False, lds_path=False):
    large, small, weather = noise
    lc, sc = pkg.assets.build_mips(large, 8), pkg.assets.build_mips(small, 6)
    rows = bands[0] * bands[3]
    out = np.zeros((rows, tile_w, 4), np.uint16)
    ic = C.c_uint64()
    win = np.zeros(2, np.float32)
    p = np.ascontiguousarray(params, np.float32)
    s = np.ascontiguousarray(sky).view(np.uint16)
    hostsim.hostsim_clouds(P(lc), P(sc), P(weather), P(p), primary, light, C.c_float(eps), P(s), s.shape[1], s.shape[0], tile_w,
                           bands[0], bands[1], bands[2], bands[3], P(out), C.byref(ic), int(window), P(win), int(lds_path))
    if ret_window:
        return out.view(np.float16), ic.value, win
    return out.view(np.float16), ic.value


def test_lut_cores_bit_exact(hostsim, o_trans, o_skies):
    t = np.zeros((64, 256, 4), np.uint16)
    hostsim.hostsim_transmittance(256, 64, P(t))
    assert (t == o_trans.view(np.uint16)).all()
    for k, sun in SUNS.items():
        s = np.zeros((100, 200, 4), np.uint16)
        sn = norm(sun)
        hostsim.hostsim_sky(200, 100, P(sn), P(o_trans.view(np.uint16)), 256, 64, P(s))
        assert (s == o_skies[k].view(np.uint16)).all(), k


def test_cloud_core_matches_oracle(hostsim, pkg, oracle, noise, otex, o_skies):
    for k, sun in SUNS.items():
        p = oracle.default_params(64, 32, sun)
        ref, st = oracle.clouds(otex, p, o_skies[k], return_stats=True)
        img, ic = hs_clouds(hostsim, pkg, noise, p, o_skies[k], 64, (8, 0, 1, 4))
        ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
        assert ok, (k, info)
        assert ic == st["incloud_samples"], k           # both exact rejects agree with the oracle's t > 0 decisions


def test_cloud_core_windy_tile(hostsim, pkg, oracle, noise, otex):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "clouds_np.npz"))
    pw, sk = g["windy_params"], g["windy_sky"].view(np.float16)
    ref = oracle.clouds(otex, pw, sk, rect=(0, 0, 48, 24), primary_steps=64, light_steps=4)
    img, _ = hs_clouds(hostsim, pkg, noise, pw, sk, 48, (8, 0, 1, 3), primary=64, light=4)
    ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
    assert ok, info


def test_band_addressing(hostsim, pkg, oracle, noise, o_skies):
    """csky_bands: rank r of N renders bands r, r+N, ...; the compact outputs interleave back to the full frame."""
    p = oracle.default_params(32, 48, (1, 1, 0))
    full, _ = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 32, (8, 0, 1, 6))
    world = 4
    parts = []
    for r in range(world):
        b = pkg.tiling.bands_for_rank(48, r, world)
        img, _ = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 32, b)
        pad = np.zeros((pkg.tiling.max_bands(48, world) * 8, 32, 4), np.float16)
        pad[: img.shape[0]] = img
        parts.append(pad)
    re = pkg.tiling.interleave(np.stack(parts, 0), 48, world)
    assert (re.view(np.uint16) == full.view(np.uint16)).all()


def test_early_out_bounded(hostsim, pkg, oracle, noise, o_skies):
    p = oracle.default_params(48, 24, (0, 1, 0), coverage=0.5)
    a, _ = hs_clouds(hostsim, pkg, noise, p, o_skies["zenith"], 48, (8, 0, 1, 3), eps=0.0)
    b, _ = hs_clouds(hostsim, pkg, noise, p, o_skies["zenith"], 48, (8, 0, 1, 3), eps=1e-3)
    d = np.abs(a.astype(np.float32) - b.astype(np.float32))
    assert d[..., 3].max() <= 1.5e-3 and d[..., :3].max() <= 1.5e-3 * max(1.0, float(a.astype(np.float32)[..., :3].max()))


def test_height_window_reject_is_exact(hostsim, pkg, oracle, noise, o_skies):
    """The height-window reject (bake.h height_window) must never change a pixel: with and without it the kernel core
    renders bit-identical images and in-cloud counts, for several coverages; and the window is non-trivial at the default."""
    for cov in (0.05, 0.2, 0.5, 0.95, 1.0, 1.5, 0.0, -0.5):
        p = oracle.default_params(48, 24, (1, 1, 0), coverage=cov)
        a, ia, win = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 48, (8, 0, 1, 3), window=True, ret_window=True)
        b, ib = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 48, (8, 0, 1, 3), window=False)
        assert (a.view(np.uint16) == b.view(np.uint16)).all() and ia == ib, cov
        if cov == 0.2:
            assert 0.03 < win[0] < 0.2 and 0.6 < win[1] < 0.95, win        # default map: clouds only between ~6 % and ~80 % height
        if cov > 1.0:
            assert win[0] == -1.0 and win[1] == 2.0                        # coverage*weather.b may exceed 1: shortcut disabled
        if cov <= 0.0:
            assert ia == 0


def test_radius_only_grows_along_a_ray_precondition_of_the_early_march_end(hostsim, oracle):
    """kernels.hip::march_compact ends a wavefront's march once every live ray is at or above the top of the height window.  That is exact
    iff no ray ever comes back below a height fraction it has reached.  Every above-horizon ray of the headline frame (2048x1024, 128 steps),
    of the 4096x2048 frame and of a small frame, walked with the kernel's own ray set-up and fp32 position updates: the height fraction
    never decreases, no sample falls back below the window top (default 0.788, and other levels), and the smallest radius gain of any step
    (a grazing ray's first step) is 14.6 m at 128 steps (1.7 m at the 1024-step maximum) against ~0.5 m of fp32 position noise."""
    out = np.zeros(4, np.float64)
    for (W, H, steps) in ((2048, 1024, 128), (4096, 2048, 128), (96, 48, 64), (512, 256, 64), (512, 256, 1024)):   # 1024 = csky_set_march's maximum
        p = np.ascontiguousarray(oracle.default_params(W, H, (1, 1, 0)), np.float32)
        for hi in (0.788, 0.5, 0.95, 0.1):
            if W > 2048 and hi != 0.788:
                continue
            hostsim.hostsim_march_monotonic(P(p), steps, W, H, C.c_float(hi), P(out))
            rays, nonmono, back, gain = out
            assert rays == (W - 1) * (H - 1) and nonmono == 0 and back == 0, (W, H, hi, out.tolist())
            assert gain > (10.0 if steps <= 128 else 1.0), (W, H, gain)     # 14.6 m at 128 steps, 1.7 m at 1024


def test_lds_detail_tap_path_matches(hostsim, pkg, oracle, noise, o_skies):
    """The detail tap of the "lds" kernel variant (eight unpacked fp16 reads, a*(1-f) + b*f) agrees with the pre-differenced
    oct-packed gather (a + f*(b-a)) to rounding: same in-cloud decisions, images within 1 fp16 ulp."""
    p = oracle.default_params(48, 24, (1, 1, 0))
    a, ia = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 48, (8, 0, 1, 3))
    b, ib = hs_clouds(hostsim, pkg, noise, p, o_skies["deg45"], 48, (8, 0, 1, 3), lds_path=True)
    assert ia == ib and ulp_diff(a, b).max() <= 1


def test_shape_cell_ranks_and_exact_coefficients(hostsim, pkg, oracle, noise, otex, o_skies):
    """The polynomial-cell layouts (csky_common.h): every rank of the shape cell (x / xy / xyz) reproduces the oracle, and all
    finite-difference coefficients of the shipped textures are exact in fp16."""
    import os
    from conftest import ROOT
    large, small, weather = noise
    lc, sc = pkg.assets.build_mips(large, 8), pkg.assets.build_mips(small, 6)
    p = oracle.default_params(64, 32, SUNS["deg45"])
    ref, st = oracle.clouds(otex, p, o_skies["deg45"], return_stats=True)
    for rank, name in ((3, "libhostsim.so"), (1, "libhostsim_p1.so"), (2, "libhostsim_p2.so")):
        hs = C.CDLL(os.path.join(ROOT, "tests", "hostsim", name))
        assert hs.hostsim_shape_poly() == rank
        hs.hostsim_inexact_coeffs.restype = C.c_uint64
        assert hs.hostsim_inexact_coeffs(P(lc), P(sc), P(weather)) == 0, rank
        img, ic = hs_clouds(hs, pkg, noise, p, o_skies["deg45"], 64, (8, 0, 1, 4))
        ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
        assert ok, (rank, info)
        assert ic == st["incloud_samples"], rank
    # white noise does leave the exact range (second differences of the fbm numerator beyond 2048), and the bake says so
    rnd = np.random.default_rng(5).integers(0, 256, (128, 128, 128, 4), dtype=np.uint8)
    assert hs.hostsim_inexact_coeffs(P(pkg.assets.build_mips(rnd, 8)), P(sc), P(weather)) > 0


def test_white_noise_textures(hostsim, pkg, oracle, o_trans):
    """Adversarial inputs: white-noise volumes and weather map (largest possible texel-to-texel differences, some finite
    differences beyond the exact fp16 range).  The filtered result still matches the oracle and every t > 0 decision agrees."""
    rng = np.random.default_rng(11)
    noise = (rng.integers(0, 256, (128, 128, 128, 4), dtype=np.uint8), rng.integers(0, 256, (32, 32, 32, 3), dtype=np.uint8),
             rng.integers(0, 256, (512, 512, 3), dtype=np.uint8))
    otex = oracle.OracleTextures(*noise)
    sk = oracle.sky_lut(norm((1, 1, 0)), o_trans)
    for cov in (0.2, 0.6):
        p = oracle.default_params(64, 32, (1, 1, 0), coverage=cov)
        ref, st = oracle.clouds(otex, p, sk, return_stats=True)
        img, ic = hs_clouds(hostsim, pkg, noise, p, sk, 64, (8, 0, 1, 4))
        ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
        assert ok, (cov, info)
        assert ic == st["incloud_samples"] and ic > 0, cov


def test_fuzz_parameters_vs_oracle(hostsim, pkg, oracle, noise, otex, o_trans):
    """Random push-constant blocks, march lengths and ragged offset tiles (conftest.fuzz_case) through the kernel cores."""
    from conftest import fuzz_case
    for seed in range(4):
        p, sun, (tw, th), primary, light = fuzz_case(seed)
        sk = oracle.sky_lut(sun, o_trans)
        ref, st = oracle.clouds(otex, p, sk, rect=(0, 0, tw, th), primary_steps=primary, light_steps=light, return_stats=True)
        rows = (th + 7) // 8
        img, ic = hs_clouds(hostsim, pkg, noise, p, sk, tw, (8, 0, 1, rows), primary=primary, light=light)
        ok, info = cloud_close(img[:th], ref, frac=0.9995, atol=1e-3, rtol=2e-3)
        assert ok, (seed, info)


def test_stratus_only_weather_map(hostsim, pkg, oracle, noise, o_skies):
    """A weather map whose cloud-type channel stays below 0.5 everywhere (stratus .. stratocumulus): the height gradient takes its
    'all low' frame-wide specialisation (FrameConsts.ct_mode == 2); the shipped map exercises 'all high', white noise the mixed path."""
    large, small, weather = noise
    w2 = weather.copy(); w2[..., 0] = w2[..., 0] // 2
    assert w2[..., 0].max() <= 127
    otex = oracle.OracleTextures(large, small, w2)
    for cov in (0.3, 0.6):
        p = oracle.default_params(64, 32, SUNS["deg45"], coverage=cov)
        ref, st = oracle.clouds(otex, p, o_skies["deg45"], return_stats=True)
        for window in (True, False):                               # False: general (select) form, no height window
            img, ic = hs_clouds(hostsim, pkg, (large, small, w2), p, o_skies["deg45"], 64, (8, 0, 1, 4), window=window)
            ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
            assert ok, (cov, window, info)
            assert ic == st["incloud_samples"], (cov, window)
        assert st["incloud_samples"] > 0, cov


def test_eager_fetch_density_is_the_same_function(hostsim, pkg, oracle, noise, o_skies):
    """The light march evaluates density through sample_density_eager() (all three gathers of a sample issued before any of the
    arithmetic).  On the host both forms must return bit-identical values for random points of the cloud shell at every LOD pair."""
    rng = np.random.default_rng(21)
    large, small, weather = noise
    lc, sc = pkg.assets.build_mips(large, 8), pkg.assets.build_mips(small, 6)
    n = 20000
    d = rng.normal(size=(n, 3)); d[:, 1] = np.abs(d[:, 1]) + 0.05; d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = rng.uniform(6001400.0, 6004100.0, n)                                   # a little beyond both shells
    pos = (d * r[:, None]).astype(np.float32)
    j = rng.integers(0, 7, n)
    lods = np.stack([np.where(j == 6, 3, np.maximum(j - 2, 0)), np.where(j == 6, 5, j)], 1).astype(np.int32)   # the march's LOD pairs
    out = np.zeros(2 * n, np.float32)
    p = np.ascontiguousarray(oracle.default_params(64, 32, SUNS["deg45"], coverage=0.6), np.float32)
    sk = np.ascontiguousarray(o_skies["deg45"]).view(np.uint16)
    hostsim.hostsim_density_forms(P(lc), P(sc), P(weather), P(p), P(sk), sk.shape[1], sk.shape[0], n, P(pos), P(lods), P(out))
    lazy, eager = out[0::2], out[1::2]
    assert (lazy.view(np.uint32) == eager.view(np.uint32)).all()
    assert (lazy > 0).mean() > 0.02 and (lazy == 0).mean() > 0.05            # both outcomes are exercised


def test_frame_constants_from_twelve_texels_equal_those_from_the_whole_lut(hostsim, oracle, o_trans):
    """What a rank of an N-way frame split does (csky_render_sky_lut_rows_device: no whole LUT in memory): the frame set-up filters the LUT at three
    places (clouds.glsl:163-167) and gets the <= 12 texels involved handed over tap-major, as frame_setup_taps_kernel parks them in LDS.  The
    constants must be those of the whole-LUT set-up to the byte, for lights towards every octant, straight up / down (atan2(0, 0), clamped rows),
    below the horizon and unnormalised; and the texels asked for are inside the LUT."""
    import ctypes as C
    sk = oracle.sky_lut(norm((1, 1, 0)), o_trans, 200, 100).view(np.uint16)
    lights = [(1, 1, 0), (0, 1, 0), (0, -1, 0), (-1, 0.05, 0.3), (0.3, -0.2, -1), (-0.998773, 0.0495291, 2.69869e-07), (5, 3, -4), (0, 0.999, 1e-4), (1e-3, 0.2, 0)]
    for l in lights:
        p = oracle.default_params(256, 128, l)
        p = np.ascontiguousarray(p, np.float32)
        a = np.zeros(1024, np.uint8); b = np.zeros(1024, np.uint8); t = (C.c_int * 12)()
        n = hostsim.hostsim_frame_setup_two_ways(p.ctypes.data_as(C.c_void_p), sk.ctypes.data_as(C.c_void_p), 200, 100, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), t)
        assert 0 < n <= 1024 and (a[:n] == b[:n]).all(), l
        assert all(0 <= v < 200 * 100 for v in t), (l, list(t))
