"""C oracle vs the committed fixtures of the INDEPENDENT numpy-fp32 restatement (tests/golden/*.npz, generated
by oracle/numpy_restatement.py).  Parity unpinned w.r.t. a real Godot frame -- see DESIGN.md §3."""
import os

import numpy as np

from conftest import GOLDEN, SUNS, cloud_close, norm, ulp_diff


def test_transmittance_lut_vs_numpy_fixture(o_trans):
    g = np.load(os.path.join(GOLDEN, "transmittance_lut_np.npz"))["lut"].view(np.float16)
    d = ulp_diff(o_trans, g)
    # numpy's SIMD exp/log/pow and glibc's differ in the last fp32 bit; after fp16 rounding that is <= 1 half ulp
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_sky_lut_vs_numpy_fixture(o_skies):
    g = np.load(os.path.join(GOLDEN, "sky_lut_np.npz"))
    for k in SUNS:
        d = ulp_diff(o_skies[k], g[k].view(np.float16))
        assert d.max() <= 1 and (d > 0).mean() < 0.03, k


def test_clouds_vs_numpy_fixture(oracle, otex, o_skies):
    g = np.load(os.path.join(GOLDEN, "clouds_np.npz"))
    for k, sun in SUNS.items():
        img, st = oracle.clouds(otex, oracle.default_params(64, 32, sun), o_skies[k], return_stats=True)
        ref = g[k].view(np.float16)
        ok, info = cloud_close(img, ref, frac=0.9995, atol=1e-3, rtol=2e-3)
        assert ok, (k, info)
        assert abs(int(st["incloud_samples"]) - int(g[k + "_incloud"])) <= 2, k    # same t > 0 decisions (clouds.glsl:184)


def test_windy_offset_tile_reduced_steps_vs_numpy_fixture(oracle, otex):
    """Every push-constant field non-default (wind offsets, time, weather_pos, light colour/energy), a tile at
    update_position != 0 rendered from a gl_GlobalInvocationID offset, and 64 x 4 steps."""
    g = np.load(os.path.join(GOLDEN, "clouds_np.npz"))
    pw = g["windy_params"]
    img = oracle.clouds(otex, pw, g["windy_sky"].view(np.float16), rect=(8, 4, 48, 24), primary_steps=64, light_steps=4)
    ok, info = cloud_close(img, g["windy"].view(np.float16), frac=0.9995, atol=1e-3, rtol=2e-3)
    assert ok, info


def test_oracle_sky_for_windy_sun(oracle, o_trans):
    g = np.load(os.path.join(GOLDEN, "clouds_np.npz"))
    sk = oracle.sky_lut(g["windy_params"][16:19], o_trans)
    assert ulp_diff(sk, g["windy_sky"].view(np.float16)).max() <= 1
