#!/usr/bin/env python
"""screenshot_stats.py -- three coarse statistics of a tonemapped sky image, shared by tools/demo_scene.py (our renders) and the reference's
screenshots (screenshots/*.png: the only image evidence the reference holds; tonemapped, camera-projected, sun / wind / coverage unknown).

    horizon_row        fraction of the image height at which the strongest horizontal luminance edge of the lower half sits (column median)
    glow               centroid (x, y as fractions) and mean luminance of the brightest 0.5 % of the pixels: where the sun's glow is, if in view
    cloud_cover        above the horizon: share of pixels that are cloud rather than clear sky (saturation (max - min) / max below 0.30 at
                       luminance above 0.25: white / grey versus blue)

Run with no arguments IN THE BUILD CONTAINER it reads /root/reference/screenshots/*.png and writes profiles/r04/reference_screenshot_stats.json
(data derived from the reference's images; the images themselves are not copied).  Qualitative: what it can catch is a shared misreading of an
axis or row order (clouds below the horizon, a mirrored sun, the hemisphere upside down), not a per-pixel difference."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(rgb):
    """rgb: float array [h, w, 3] in 0..1 (display-referred)."""
    rgb = np.asarray(rgb, np.float64)[..., :3]
    h, w, _ = rgb.shape
    lum = rgb @ np.array([0.2126, 0.7152, 0.0722])
    k = 3                                                      # vertical box blur, 7 rows: the edge, not the noise
    c = np.cumsum(np.vstack([np.zeros((1, w)), lum]), 0)
    blur = (c[2 * k + 1:] - c[:-(2 * k + 1)]) / (2 * k + 1)     # rows k .. h-k-1
    g = np.abs(blur[2:] - blur[:-2])                            # rows k+1 .. h-k-2
    lo = h // 2 - (k + 1)
    rows = np.argmax(g[lo:], 0) + lo + k + 1
    horizon = float(np.median(rows) / h)
    edge_strength = float(np.median(np.max(g[lo:], 0)))
    thr = np.quantile(lum, 0.995)
    yy, xx = np.nonzero(lum >= thr)
    glow = {"x": float(xx.mean() / w), "y": float(yy.mean() / h), "mean_luminance": float(lum[lum >= thr].mean()), "spread": float(np.hypot(xx.std() / w, yy.std() / h))}
    hr = int(horizon * h) if edge_strength > 0.01 else h
    sky = rgb[: max(1, hr - int(0.03 * h))]                     # leave the haze band at the horizon out
    mx, mn = sky.max(-1), sky.min(-1)
    sat = (mx - mn) / np.maximum(mx, 1e-6)
    cloud = (sat < 0.30) & (sky @ np.array([0.2126, 0.7152, 0.0722]) > 0.25)
    return {"size": [w, h], "horizon_row": horizon if edge_strength > 0.01 else None, "horizon_edge_strength": edge_strength, "glow": glow,
            "cloud_cover": float(cloud.mean()), "mean_rgb_above_horizon": [float(v) for v in sky.reshape(-1, 3).mean(0)]}


if __name__ == "__main__":
    from PIL import Image
    src = "/root/reference/screenshots"
    if len(sys.argv) > 1:
        for p in sys.argv[1:]:
            print(p, json.dumps(stats(np.asarray(Image.open(p).convert("RGB"), np.float64) / 255.0)))
    elif os.path.isdir(src):
        out = {}
        for n in sorted(os.listdir(src)):
            if n.endswith(".png"):
                out["screenshots/" + n] = stats(np.asarray(Image.open(os.path.join(src, n)).convert("RGB"), np.float64) / 255.0)
        os.makedirs(os.path.join(ROOT, "profiles", "r04"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "profiles", "r04", "reference_screenshot_stats.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
    else:
        print("no screenshots here: pass image files")
