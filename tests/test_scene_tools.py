"""The statistics the demo-scene fit is read with (tools/screenshot_stats.py, tools/demo_scene.py: VERDICT r4 item 1) on synthetic pictures whose
answers are known: a wrong horizon row, cover or spectrum there would make profiles/r05/demo_scene_fit.txt say something about the tool, not about
the scene.  CPU only; the renders themselves are GPU work (tools/demo_scene.py on the GPU box)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _picture(h=300, w=400, horizon=0.8, blob=16, cover=0.5, seed=0, dusk=False):
    """blue sky with white round 'clouds' of radius ~`blob` covering ~`cover` of it, a pale haze band, dark ground below row horizon * h"""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 3))
    hr = int(horizon * h)
    img[:hr] = (0.05, 0.08, 0.25) if dusk else (0.20, 0.35, 0.75)
    yy, xx = np.mgrid[:hr, :w]
    mask = np.zeros((hr, w), bool)
    n = int(-np.log(1.0 - cover) * hr * w / (np.pi * blob * blob))         # Poisson discs: cover = 1 - exp(-n pi r^2 / area)
    for cx, cy, r in zip(rng.uniform(0, w, n), rng.uniform(0, hr, n), rng.uniform(0.7, 1.3, n) * blob):
        mask |= (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
    img[:hr][mask] = (0.55, 0.45, 0.25) if dusk else (0.85, 0.85, 0.88)
    img[hr - 6:hr] = (0.30, 0.25, 0.22) if dusk else (0.75, 0.72, 0.68)
    img[hr:] = (0.02, 0.02, 0.03)
    return img, float(mask[: hr - int(0.03 * h)].mean())


def test_horizon_cover_and_glow_on_a_synthetic_sky():
    from screenshot_stats import stats
    img, cover = _picture()
    img[50:80, 290:320] = 1.0                                   # 900 of 120 000 pixels: the brightest half percent is inside this 'sun glow'
    s = stats(img)
    assert abs(s["horizon_row"] - 0.8) < 0.02
    assert abs(s["cloud_cover"] - cover) < 0.03
    assert abs(s["glow"]["x"] - 305 / 400) < 0.05 and abs(s["glow"]["y"] - 65 / 300) < 0.05
    assert s["mean_rgb_above_horizon"][2] > s["mean_rgb_above_horizon"][0]      # blue sky, white clouds


def test_horizon_detector_is_not_drawn_to_cloud_edges_in_a_dusk_picture():
    """Round 4's detector (each column's own strongest edge, then the median row) put the horizon of the scene as committed at 0.612 of the height: in a
    dim picture the cloud edges are stronger than the horizon.  The row-wise column median must find the horizon, which spans every column."""
    from screenshot_stats import stats
    img, _ = _picture(horizon=0.9, blob=30, cover=0.5, seed=3, dusk=True)
    s = stats(img)
    assert s["horizon_row"] is not None and abs(s["horizon_row"] - 0.9) < 0.02, s["horizon_row"]


def test_mask_spectrum_tells_billows_from_blobs():
    from screenshot_stats import stats
    coarse = stats(_picture(blob=40, seed=1)[0])["cloud_mask_spectrum"]
    fine = stats(_picture(blob=5, seed=1)[0])["cloud_mask_spectrum"]
    assert abs(sum(coarse["power_share"].values()) - 1.0) < 1e-9
    assert fine["mean_cycles_per_width"] > 2.0 * coarse["mean_cycles_per_width"]
    assert coarse["power_share"]["1-4"] > fine["power_share"]["1-4"] and fine["power_share"]["16-64"] > coarse["power_share"]["16-64"]
    # the unit is cycles per image WIDTH: the same picture at half the resolution reads the same
    half = stats(_picture(blob=40, seed=1)[0][::2, ::2])["cloud_mask_spectrum"]
    assert abs(half["mean_cycles_per_width"] - coarse["mean_cycles_per_width"]) < 0.25 * coarse["mean_cycles_per_width"]


def test_camera_helpers_and_tonemap(pkg):
    D = pytest.importorskip("demo_scene")
    fwd = -D.CAM[:, 2]
    assert np.allclose(D.cam_ray(0.5, 0.5), fwd / np.linalg.norm(fwd), atol=1e-6)           # the centre pixel looks down the camera's -z (cloud-demo.tscn:18)
    assert np.allclose(D.CAM.T @ D.CAM, np.eye(3), atol=1e-4)                               # the .tscn basis is orthonormal
    cam = D.look(fwd, 19.8)
    assert np.allclose(cam.T @ cam, np.eye(3), atol=1e-5) and abs(np.degrees(np.arcsin(-cam[1, 2])) - 19.8) < 1e-3 and abs(cam[1, 0]) < 1e-6   # no roll
    # a pixel's ray, projected back, lands on that pixel
    t = np.tan(np.radians(D.FOV / 2))
    for u, v in ((0.633, 0.671), (0.958, 0.713), (0.1, 0.2)):
        d = D.cam_ray(u, v, cam)
        x, y, z = cam[:, 0] @ d, cam[:, 1] @ d, -(cam[:, 2] @ d)
        assert abs((x / z / (t * D.W / D.H) + 1) / 2 - u) < 1e-6 and abs((1 - y / z / t) / 2 - v) < 1e-6
    # the horizon row the Sunset fit derives its pitch from is where a horizontal ray lands
    hz = D.cam_ray(0.5, 0.5 + np.tan(np.radians(19.8)) / (2 * t), cam)
    assert abs(hz[1]) < 1e-6
    x = np.linspace(0, 6, 50)[:, None] * np.ones((1, 3))
    y = D.aces(x)
    assert y[0].max() == 0.0 and (np.diff(y[:, 0]) >= 0).all() and abs(y[-1, 0] - 1.0) < 1e-9 and 0.5 < D.aces(np.full((1, 3), 0.5))[0, 0] < 0.9
    # miss(): a render with the screenshot's own statistics is a perfect fit
    from screenshot_stats import stats
    s = stats(_picture()[0])
    m = D.miss(s, s)
    assert m["cover"] == 0 and m["rgb_worst"] == 0 and m["spectrum_l1"] == 0 and m["score"] == 0
