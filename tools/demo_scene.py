#!/usr/bin/env python
"""demo_scene.py -- the reference's demo scene end to end through the HIP path, as a PNG (GPU box).

cloud_sky/cloud-demo.tscn: Camera3D transform (:18), DirectionalLight3D transform (:21), Environment tonemap_mode = 3 (ACES), tonemap_white = 3.53
(:9-10); Camera3D's default vertical fov 75 degrees; the screenshots are 1147x629.  Pipeline: transmittance LUT -> sky LUT -> clouds (2048x1024,
128 x 6) -> csky_composite_view (clouds.gdshader sky() per screen pixel) -> ACES tonemap + sRGB -> PNG under profiles/r04/, with the three coarse
statistics of tools/screenshot_stats.py next to the same statistics of the reference's own screenshots (profiles/r04/reference_screenshot_stats.json).

Two renders: (a) the scene exactly as committed (sun 2.8 degrees above the horizon BEHIND the camera: clouds lit from behind the viewer, no glow in view);
(b) the same camera with a high sun (elevation 55 degrees, to the viewer's right) and cloud_coverage 0.35, the kind of setting screenshots/Clouds.png shows
(its sun / wind / coverage are not recorded anywhere).  Qualitative: horizon where the camera puts it, sky above it, clouds in the sky and not on the
ground, glow on the sun's side -- what a shared misreading of an axis or row order would break."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gvcd_amd  # noqa: E402
from screenshot_stats import stats  # noqa: E402

# Transform3D(...) of a .tscn lists the basis row by row; its COLUMNS are the node's x, y, z axes
CAM = np.array([[0.105461, -0.534173, -0.838771], [-0.00147199, 0.84339, -0.5373], [0.994422, 0.0578988, 0.0881584]], np.float32)        # cloud-demo.tscn:18
SUN_T = np.array([[-0.0492487, -0.00526289, -0.998773], [-0.993118, -0.106134, 0.0495291], [-0.106264, 0.994338, 2.69869e-07]], np.float32)  # :21
SCENE_SUN = SUN_T[:, 2] / np.linalg.norm(SUN_T[:, 2])           # cloud_sky.gd:76-77: light.basis * (0, 0, 1) = towards the sun
W, H, FOV = 1147, 629, 75.0


def aces(x, white=3.53):
    """Godot 4's tonemap_mode 3 (the fitted ACES curve, RRT + ODT approximation) with its white point, then the sRGB transfer function."""
    def fit(v):
        m1 = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        m2 = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        v = v @ m1.T
        v = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
        return v @ m2.T
    exposure_bias = 1.8
    y = fit(np.maximum(x, 0.0) * exposure_bias) / fit(np.full((1, 3), white * exposure_bias))
    y = np.clip(y, 0.0, 1.0)
    return np.where(y <= 0.0031308, 12.92 * y, 1.055 * y ** (1 / 2.4) - 0.055)


def render(ctx, sun, coverage, name):
    sun = np.asarray(sun, np.float32) / np.linalg.norm(sun)
    ctx.render_sky_lut(sun, 200, 100)
    sky = ctx.read_sky_lut()
    p = np.array([2048, 1024, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, sun[0], sun[1], sun[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, coverage, 0.0], np.float32)   # clouds_sky.tres:11-17, wind frozen
    cl = ctx.render_clouds(p)
    view = ctx.composite_view(cl, cl, sky, sky, sun, CAM, FOV, 0.0, 2.0, W, H).astype(np.float32)[..., :3]
    img = aces(view.reshape(-1, 3)).reshape(H, W, 3)
    from PIL import Image
    out = os.path.join(ROOT, "profiles", "r04", name + ".png")
    Image.fromarray((img * 255.0 + 0.5).astype(np.uint8)).save(out)
    s = stats(img)
    s.update({"sun": [float(v) for v in sun], "cloud_coverage": coverage, "hemisphere_alpha_mean": float(cl[..., 3].astype(np.float32).mean()), "png": os.path.relpath(out, ROOT)})
    _IMAGES[name] = (img * 255.0 + 0.5).astype(np.uint8)
    return s


_IMAGES = {}


def main():
    os.makedirs(os.path.join(ROOT, "profiles", "r04"), exist_ok=True)
    ctx = gvcd_amd.Context(0)
    ctx.set_noise(*gvcd_amd.assets.load_default_noise())
    ctx.render_transmittance(256, 64)
    fwd = -CAM[:, 2]
    right = CAM[:, 0]
    el = np.radians(55.0)
    high = np.cos(el) * (right * np.array([1, 0, 1])) / max(1e-6, np.linalg.norm(right * np.array([1, 0, 1]))) + np.array([0, np.sin(el), 0])
    res = {"camera_forward": [float(v) for v in fwd], "camera_pitch_degrees": float(np.degrees(np.arcsin(fwd[1]))), "fov_y_degrees": FOV,
           "expected_horizon_row": float(0.5 + np.tan(np.arcsin(fwd[1])) / (2.0 * np.tan(np.radians(FOV / 2)))),
           "scene_as_committed": render(ctx, SCENE_SUN, 0.2, "demo_scene_as_committed"),
           "high_sun_coverage_0.35": render(ctx, high, 0.35, "demo_scene_high_sun")}
    # (c) render (b) again from the inputs as compress/mode=2 of the *.import files would leave them (this library's BC7 encoder in the importer's place,
    # DESIGN.md 3): what the texture compression the reference runs with does to the picture on screen
    large, small, weather = gvcd_amd.assets.load_default_noise()
    (lq, sq, wq), tex = gvcd_amd.assets.vram_compressed_chains(ctx, large, small, weather)
    ctx.set_noise_mips(lq, sq, wq)
    res["high_sun_coverage_0.35_bc7_inputs"] = render(ctx, high, 0.35, "demo_scene_high_sun_bc7_inputs")
    a, b = _IMAGES["demo_scene_high_sun"].astype(np.float64), _IMAGES["demo_scene_high_sun_bc7_inputs"].astype(np.float64)
    d = np.abs(a - b)
    res["high_sun_coverage_0.35_bc7_inputs"]["vs_uncompressed_inputs_8bit_srgb"] = {
        "psnr_db": float(10 * np.log10(255.0 ** 2 / max(1e-12, (d ** 2).mean()))), "mean_abs_levels": float(d.mean()), "max_abs_levels": float(d.max()),
        "pixels_off_by_more_than_2_levels": float((d.max(-1) > 2).mean()), "texture_round_trip_psnr_db": tex}
    ref = os.path.join(ROOT, "profiles", "r04", "reference_screenshot_stats.json")
    if os.path.exists(ref):
        res["reference_screenshots"] = json.load(open(ref))
    json.dump(res, open(os.path.join(ROOT, "profiles", "r04", "demo_scene_stats.json"), "w"), indent=1)
    v = res["high_sun_coverage_0.35_bc7_inputs"]["vs_uncompressed_inputs_8bit_srgb"]
    print("BC7-compressed inputs vs uncompressed, 8-bit sRGB picture: PSNR %.1f dB, mean |d| %.2f levels, max %d, %.1f %% of the pixels off by more than 2 levels" % (
        v["psnr_db"], v["mean_abs_levels"], v["max_abs_levels"], 100 * v["pixels_off_by_more_than_2_levels"]))
    for k in ("scene_as_committed", "high_sun_coverage_0.35", "high_sun_coverage_0.35_bc7_inputs"):
        s = res[k]
        print("%-24s horizon row %s (camera geometry: %.3f)   glow at (%.2f, %.2f) luminance %.2f   cloud cover %.2f   -> %s" % (
            k, "%.3f" % s["horizon_row"] if s["horizon_row"] else "none", res["expected_horizon_row"], s["glow"]["x"], s["glow"]["y"], s["glow"]["mean_luminance"], s["cloud_cover"], s["png"]))
    for k, s in (res.get("reference_screenshots") or {}).items():
        print("%-24s horizon row %s   glow at (%.2f, %.2f) luminance %.2f   cloud cover %.2f" % (k, "%.3f" % s["horizon_row"] if s["horizon_row"] else "none", s["glow"]["x"], s["glow"]["y"], s["glow"]["mean_luminance"], s["cloud_cover"]))
    ctx.close()


if __name__ == "__main__":
    main()
