#!/usr/bin/env python
"""demo_scene.py -- the reference's demo scene end to end through the HIP path, put next to the only images the reference holds (GPU box).

cloud_sky/cloud-demo.tscn: Camera3D transform (:18), DirectionalLight3D transform (:21), Environment tonemap_mode = 3 (ACES), tonemap_white = 3.53
(:9-10); Camera3D's default vertical fov 75 degrees; the screenshots are 1147x629.  Pipeline: transmittance LUT -> sky LUT -> clouds (2048x1024,
128 x 6) -> csky_composite_view (clouds.gdshader sky() per screen pixel) -> ACES tonemap + sRGB -> PNG under profiles/r05/.

What the screenshots were taken with (sun, wind, coverage, even the camera of Sunset.png) is recorded nowhere, and the shape volume they show
(cloud_sky/perlworlnoise.tga) is missing; so this is a FIT, not a comparison: for each screenshot the tool puts the sun where the picture's
brightest half percent sits (screenshots/Clouds.png (0.63, 0.67), screenshots/Sunset.png (0.96, 0.71)), sweeps cloud_coverage and
the knobs of the stand-in shape generator (csky_shape_noise_params), and prints, per render, the statistics of tools/screenshot_stats.py beside the
screenshot's: cloud cover, mean RGB above the horizon, glow position / luminance, horizon row, and the radially averaged spectrum of the cloud mask.
It then says which statistic lands within 0.1 (cover) / 20 % (mean RGB) and which does not for ANY setting, and what single linear gain (an unknown
light energy / exposure) would close the brightness gap.  profiles/r05/demo_scene_fit.txt is the table, demo_scene_fit_{clouds,sunset}.png the best
rows (and *_default_noise.png the best row of the unmodified benchmark volume)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gvcd_amd  # noqa: E402
from screenshot_stats import stats  # noqa: E402

OUT = os.path.join(ROOT, "profiles", "r05")
# Transform3D(...) of a .tscn lists the basis row by row; its COLUMNS are the node's x, y, z axes
CAM = np.array([[0.105461, -0.534173, -0.838771], [-0.00147199, 0.84339, -0.5373], [0.994422, 0.0578988, 0.0881584]], np.float32)        # cloud-demo.tscn:18
SUN_T = np.array([[-0.0492487, -0.00526289, -0.998773], [-0.993118, -0.106134, 0.0495291], [-0.106264, 0.994338, 2.69869e-07]], np.float32)  # :21
SCENE_SUN = SUN_T[:, 2] / np.linalg.norm(SUN_T[:, 2])           # cloud_sky.gd:76-77: light.basis * (0, 0, 1) = towards the sun
W, H, FOV = 1147, 629, 75.0
# the stand-in generator's settings the sweep visits (csky_shape_noise_params; {} = the benchmark volume)
NOISES = {"default": {}, "perlin8": dict(perlin_freq=8, perlin_octaves=4), "perlin8_dilate0.8": dict(perlin_freq=8, perlin_octaves=4, dilate=0.8),
          "offset0.42": dict(offset=0.42), "contrast2.5": dict(contrast=2.5), "worley8": dict(worley_freq=8)}


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


def cam_ray(u, v, cam=CAM):
    """World direction through the screen point (u, v) (fractions of width / height, v down) of a camera with basis `cam`, vertical fov FOV."""
    t = np.tan(np.radians(FOV / 2))
    d = cam[:, 0] * ((2 * u - 1) * t * W / H) + cam[:, 1] * ((1 - 2 * v) * t) - cam[:, 2]
    return d / np.linalg.norm(d)


def look(heading, pitch_degrees):
    """Basis of a camera looking along `heading` (its horizontal part) pitched up by pitch_degrees, no roll."""
    p = np.radians(pitch_degrees)
    h = np.array([heading[0], 0.0, heading[2]]); h /= np.linalg.norm(h)
    f = np.cos(p) * h + np.array([0.0, np.sin(p), 0.0])
    r = np.cross(f, [0.0, 1.0, 0.0]); r /= np.linalg.norm(r)
    return np.stack([r, np.cross(r, f), -f], 1).astype(np.float32)


class Scene:
    def __init__(self, ctx):
        self.ctx = ctx
        _, self.small, self.weather = gvcd_amd.assets.load_default_noise()
        self.bound = None
        ctx.render_transmittance(256, 64)

    def bind(self, noise):
        if self.bound != noise:
            self.ctx.set_noise(self.ctx.generate_shape_noise(1, 128, **NOISES[noise]), self.small, self.weather)   # the GPU bake (byte-identical to the host generator)
            self.bound = noise

    def linear(self, cam, sun, coverage, noise):
        """The scene's linear-light picture [H, W, 3] before tonemapping, and the hemisphere's mean alpha."""
        self.bind(noise)
        sun = np.asarray(sun, np.float32) / np.linalg.norm(sun)
        self.ctx.render_sky_lut(sun, 200, 100)
        sky = self.ctx.read_sky_lut()
        p = np.array([2048, 1024, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, sun[0], sun[1], sun[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, coverage, 0.0], np.float32)   # clouds_sky.tres:11-17, wind frozen
        cl = self.ctx.render_clouds(p)
        view = self.ctx.composite_view(cl, cl, sky, sky, sun, cam, FOV, 0.0, 2.0, W, H).astype(np.float32)[..., :3]
        return view, float(cl[..., 3].astype(np.float32).mean())

    def render(self, cam, sun, coverage, noise):
        view, alpha = self.linear(cam, sun, coverage, noise)
        img = aces(view.reshape(-1, 3)).reshape(H, W, 3)
        s = stats(img)
        s.update({"sun": [float(v) for v in np.asarray(sun) / np.linalg.norm(sun)], "sun_elevation_degrees": float(np.degrees(np.arcsin(sun[1] / np.linalg.norm(sun)))),
                  "cloud_coverage": coverage, "noise": noise, "hemisphere_alpha_mean": alpha})
        return s, img, view


def save(img, name):
    from PIL import Image
    out = os.path.join(OUT, name + ".png")
    Image.fromarray((img * 255.0 + 0.5).astype(np.uint8)).save(out)
    return os.path.relpath(out, ROOT)


def miss(s, ref):
    """How far a render's statistics are from a screenshot's: cover difference, worst channel ratio of the mean RGB, spectrum distance."""
    rgb = np.array(s["mean_rgb_above_horizon"]) / np.array(ref["mean_rgb_above_horizon"])
    a, b = s.get("cloud_mask_spectrum"), ref.get("cloud_mask_spectrum")
    spec = float(sum(abs(a["power_share"][k] - b["power_share"][k]) for k in b["power_share"])) if a and b else 1.0
    return {"cover": s["cloud_cover"] - ref["cloud_cover"], "rgb_ratio": [float(v) for v in rgb], "rgb_worst": float(np.abs(rgb - 1.0).max()), "spectrum_l1": spec,
            "glow": float(np.hypot(s["glow"]["x"] - ref["glow"]["x"], s["glow"]["y"] - ref["glow"]["y"])),
            "score": abs(s["cloud_cover"] - ref["cloud_cover"]) / 0.1 + float(np.abs(rgb - 1.0).max()) / 0.2 + spec / 0.3}


def exposure_fit(view, ref):
    """The single gain on the LINEAR picture (an unknown light energy / tonemap exposure: neither is recorded for the screenshots) that brings the
    tonemapped mean RGB above the horizon closest to the screenshot's, and the channel ratios left after it: if they are near 1 the gap is exposure,
    if they are not it is colour (sun height, the atmosphere, the lighting terms)."""
    best = None
    for g in np.exp(np.linspace(np.log(0.5), np.log(4.0), 25)):
        s = stats(aces((view * g).reshape(-1, 3)).reshape(H, W, 3))
        r = np.array(s["mean_rgb_above_horizon"]) / np.array(ref["mean_rgb_above_horizon"])
        e = float(np.abs(np.log(r)).sum())
        if best is None or e < best[0]:
            best = (e, float(g), [float(v) for v in r], s["cloud_cover"])
    return {"gain": best[1], "rgb_ratio_after": best[2], "cloud_cover_after": best[3]}


def row(tag, s, m):
    sp = s.get("cloud_mask_spectrum")
    return "%-46s cover %.2f (%+.2f)  rgb %.2f %.2f %.2f (x%.2f %.2f %.2f)  glow (%.2f, %.2f) L %.2f  horizon %s  spectrum %s  alpha %.2f" % (
        tag, s["cloud_cover"], m["cover"], *s["mean_rgb_above_horizon"], *m["rgb_ratio"], s["glow"]["x"], s["glow"]["y"], s["glow"]["mean_luminance"],
        "%.3f" % s["horizon_row"] if s["horizon_row"] else "none ",
        "%.2f/%.2f/%.2f k=%4.1f" % (sp["power_share"]["1-4"], sp["power_share"]["4-16"], sp["power_share"]["16-64"], sp["mean_cycles_per_width"]) if sp else "-", s["hemisphere_alpha_mean"])


def ref_row(tag, s):
    sp = s["cloud_mask_spectrum"]
    return "%-46s cover %.2f          rgb %.2f %.2f %.2f                     glow (%.2f, %.2f) L %.2f  horizon %s  spectrum %.2f/%.2f/%.2f k=%4.1f" % (
        tag, s["cloud_cover"], *s["mean_rgb_above_horizon"], s["glow"]["x"], s["glow"]["y"], s["glow"]["mean_luminance"], "%.3f" % s["horizon_row"] if s["horizon_row"] else "none ",
        sp["power_share"]["1-4"], sp["power_share"]["4-16"], sp["power_share"]["16-64"], sp["mean_cycles_per_width"])


def fit(sc, name, ref, cam, suns, coverages, noises, lines, res):
    lines.append("== fit to %s   (camera pitch %.1f degrees, expected horizon row %.3f)" % (name, np.degrees(np.arcsin(-cam[1, 2])), 0.5 + np.tan(np.arcsin(-cam[1, 2])) / (2.0 * np.tan(np.radians(FOV / 2)))))
    lines.append(ref_row("reference " + name, ref))
    rows = []
    for sun_tag, sun in suns:
        for noise in noises:
            for c in coverages:
                s, img, view = sc.render(cam, sun, c, noise)
                m = miss(s, ref)
                tag = "%s %s c=%.2f" % (sun_tag, noise, c)
                rows.append((m["score"], tag, s, m, img, view))
                lines.append(row(tag, s, m))
    out = {}
    for kind, sel in (("best", rows), ("best_default_noise", [r for r in rows if r[2]["noise"] == "default"])):
        if not sel:
            continue
        _, tag, s, m, img, view = min(sel, key=lambda r: r[0])
        key = name.split("/")[-1].replace(".png", "").replace(" ", "_").lower()
        png = save(img, "demo_scene_fit_%s%s" % (key, "" if kind == "best" else "_default_noise"))
        ex = exposure_fit(view, ref)
        out[kind] = {"setting": tag, "stats": s, "miss": m, "exposure_fit": ex, "png": png}
        lines.append("-> %s: %s   cover %+.2f, worst channel %.0f %% off, spectrum L1 %.2f   [%s]" % (kind, tag, m["cover"], 100 * m["rgb_worst"], m["spectrum_l1"], png))
        lines.append("   one linear gain of %.2f on the picture leaves channel ratios %.2f %.2f %.2f and cover %.2f" % (ex["gain"], *ex["rgb_ratio_after"], ex["cloud_cover_after"]))
    within_cover = sorted(set(r[1] for r in rows if abs(r[3]["cover"]) <= 0.1))
    within_rgb = sorted(set(r[1] for r in rows if r[3]["rgb_worst"] <= 0.2))
    both = sorted(set(within_cover) & set(within_rgb))
    lines.append("   settings with cover within 0.1: %d of %d; mean RGB within 20 %%: %d; both: %d%s" % (len(within_cover), len(rows), len(within_rgb), len(both), (" (" + "; ".join(both[:6]) + ")") if both else ""))
    out["rows"] = [{"setting": t, "stats": s, "miss": m} for _, t, s, m, _, _ in rows]
    out["within"] = {"cover": within_cover, "rgb": within_rgb, "both": both}
    res[name] = out
    lines.append("")


def main():
    os.makedirs(OUT, exist_ok=True)
    refs = json.load(open(os.path.join(OUT, "reference_screenshot_stats.json")))
    ctx = gvcd_amd.Context(0)
    sc = Scene(ctx)
    lines, res = [], {"fov_y_degrees": FOV}
    # (a) the scene exactly as committed: sun 2.8 degrees up BEHIND the camera; what the detector says about its horizon
    s, img, _ = sc.render(CAM, SCENE_SUN, 0.2, "default")
    exp_row = 0.5 + np.tan(np.arcsin(-CAM[1, 2])) / (2.0 * np.tan(np.radians(FOV / 2)))
    res["scene_as_committed"] = dict(s, png=save(img, "demo_scene_as_committed"), expected_horizon_row=float(exp_row))
    lines.append("scene as committed (cloud-demo.tscn:18,21; sun %.1f degrees up behind the camera): horizon row %s, camera geometry %.3f, screenshots/Clouds.png %.3f" % (
        s["sun_elevation_degrees"], "%.3f" % s["horizon_row"] if s["horizon_row"] else "none", exp_row, refs["screenshots/Clouds.png"]["horizon_row"]))
    lines.append("")
    # (b) screenshots/Clouds.png: the committed camera, the sun through the picture's brightest spot, then the same azimuth at other heights
    sun_a = cam_ray(0.633, 0.671)
    az = np.arctan2(sun_a[2], sun_a[0])
    at = lambda el: np.array([np.cos(np.radians(el)) * np.cos(az), np.sin(np.radians(el)), np.cos(np.radians(el)) * np.sin(az)])   # noqa: E731
    fit(sc, "screenshots/Clouds.png", refs["screenshots/Clouds.png"], CAM, [("sun@glow(%.1fdeg)" % np.degrees(np.arcsin(sun_a[1])), sun_a)], (0.2, 0.25, 0.3, 0.4, 0.5), list(NOISES), lines, res)
    fit(sc, "screenshots/Clouds.png sun height", refs["screenshots/Clouds.png"], CAM, [("sun@%ddeg" % e, at(e)) for e in (30, 50, 75)], (0.2, 0.3), ("default", "perlin8_dilate0.8"), lines, res)
    # (c) screenshots/Sunset.png: its horizon sits at row 0.734, i.e. another pitch than the committed camera's; heading kept, sun through the glow
    # at the right edge and, because the sky of the screenshot is still blue, a little higher just outside the frame
    pitch = np.degrees(np.arctan((refs["screenshots/Sunset.png"]["horizon_row"] - 0.5) * 2 * np.tan(np.radians(FOV / 2))))
    cam2 = look(-CAM[:, 2], pitch)
    suns = [("sun@(%.2f,%.2f;%.1fdeg)" % (u, v, np.degrees(np.arcsin(cam_ray(u, v, cam2)[1]))), cam_ray(u, v, cam2)) for u, v in ((0.958, 0.713), (1.03, 0.66), (1.05, 0.60), (1.08, 0.50))]
    fit(sc, "screenshots/Sunset.png", refs["screenshots/Sunset.png"], cam2, suns, (0.08, 0.12, 0.16, 0.2), ("default", "perlin8"), lines, res)
    ctx.close()
    json.dump(res, open(os.path.join(OUT, "demo_scene_fit.json"), "w"), indent=1)
    open(os.path.join(OUT, "demo_scene_fit.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
