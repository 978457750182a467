#!/usr/bin/env python
"""bc7_sensitivity.py -- what compress/mode=2 of the reference's *.import files does to a frame (GPU box).

The reference's samplers return DECODED BC7 blocks (weather.bmp.import:19-20, worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20); this
build marches the uncompressed bytes by default.  The engine's encoder cannot be reproduced, so its exact texels are unknown; this tool puts a
number on the SIZE of the effect with the library's own encoder (csky_encode_bc7, modes 6 / 1 / 5) in the importer's place: the C3 frame from the
uncompressed inputs next to the frame from the same inputs after encode -> decode of every mip level (assets.vram_compressed_chains), per texture and
all three together.  A weaker encoder than the engine's overstates the effect, so read the q0 figures as an upper estimate of its order of magnitude:
it is what "bit-identical to the oracle" has to be weighed against when the question is "identical to the reference".
Round 5, the error bar (no independent BC7 encoder exists in this image): a second, stronger setting of the library's encoder
(quality 1) and the FORMAT'S FLOOR -- the shape chain through the best unquantised line fits of the format's subset shapes, which no BC7 encoder
can beat (tools/bc7_ideal_bound.py) -- bracket what the engine's `high_quality=true` encoder can do to the frame."""
import json, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd
from gvcd_amd import assets

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 1024)
s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
large, small, weather = assets.load_default_noise()
ctx = gvcd_amd.Context(0)
(lq, sq, wq), tex = assets.vram_compressed_chains(ctx, large, small, weather)
(lq1, sq1, wq1), tex1 = assets.vram_compressed_chains(ctx, large, small, weather, quality=1)     # the second encoder (csky_encode_bc7_quality 1)
lraw, sraw = assets.build_mips(large, 8), assets.build_mips(small, 6)
# the third row is not an encoder but the format's floor: the shape chain passed through the best UNQUANTISED line fits the format's subset shapes allow
# (tools/bc7_ideal_bound.py --write tools/_cache/shape_chain_ideal.npz, minutes of numpy, made where no GPU is waiting): no BC7 encoder does better
ideal, ideal_psnr = None, None
_ip = os.path.join(ROOT, "tools", "_cache", "shape_chain_ideal.npz")
if os.path.exists(_ip):
    _z = np.load(_ip)
    if str(_z["shape_sha256"]) == assets.sha256(large):
        ideal, ideal_psnr = _z["chain"], float(_z["psnr_level0"])


def frame(lc, sc, w):
    c = gvcd_amd.Context(0)
    try:
        c.set_noise_mips(lc, sc, w)
        c.render_transmittance(256, 64); c.render_sky_lut(s, 200, 100, readback=False)
        img = c.render_clouds(p).astype(np.float32)
        return img, c.cloud_stats()["incloud_samples"], c.noise_inexact_coeffs()
    finally:
        c.close()


def compare(a, b):
    d = np.abs(a - b)
    peak = float(np.abs(a[..., :3]).max())
    mse = float((d[..., :3] ** 2).mean())
    ulp = np.maximum(np.abs(a), 6.1e-5) * 2.0 ** -10                          # one fp16 step at the value's magnitude
    return {"psnr_rgb_db_vs_peak": float("inf") if mse == 0 else 10 * np.log10(peak * peak / mse), "max_abs_rgb": float(d[..., :3].max()),
            "mean_abs_rgb": float(d[..., :3].mean()), "mean_rel_rgb": float((d[..., :3] / np.maximum(np.abs(a[..., :3]), 1e-3)).mean()),
            "alpha_max_abs": float(d[..., 3].max()), "alpha_mean_abs": float(d[..., 3].mean()),
            "values_within_1_fp16_step": float((d <= ulp).mean()), "values_within_16_steps": float((d <= 16 * ulp).mean())}


base, n0, _ = frame(lraw, sraw, weather)
out = {"frame": "%dx%d, 128 x 6 steps, sun (1,1,0), default parameters" % (W, H), "texture_round_trip_psnr_db": {"quality_0": tex, "quality_1": tex1, "shape_ideal_bound_level0": ideal_psnr},
       "uncompressed_vs_box_chain_through_set_noise_mips": "the uncompressed frame is rendered through the same csky_set_noise_mips path", "cases": {}}
cases = [("q0 weather map only", (lraw, sraw, wq)), ("q0 detail volume only", (lraw, sq, weather)), ("q0 shape volume only", (lq, sraw, weather)), ("q0 all three", (lq, sq, wq)),
         ("q1 shape volume only", (lq1, sraw, weather)), ("q1 all three", (lq1, sq1, wq1))]
if ideal is not None:
    cases += [("format floor: shape volume only", (ideal, sraw, weather)), ("format floor shape + q1 others", (ideal, sq1, wq1))]
for name, (lc, sc, w) in cases:
    img, n, inexact = frame(lc, sc, w)
    r = compare(base, img)
    r["in_cloud_samples"] = [int(n0), int(n)]; r["cells_beyond_fp16_pairs"] = int(inexact)
    out["cases"][name] = r
    print("%-32s PSNR %.1f dB  mean |d| %.2e (%.2f %% of the value)  max |d| %.3g  within 1 / 16 fp16 steps: %.3f / %.3f  in-cloud samples %d -> %d"
          % (name, r["psnr_rgb_db_vs_peak"], r["mean_abs_rgb"], 100 * r["mean_rel_rgb"], r["max_abs_rgb"], r["values_within_1_fp16_step"], r["values_within_16_steps"], n0, n), flush=True)
print("texture round trips, level 0, dB: shape %.2f (q0) / %.2f (q1) / %s (format floor: no encoder exceeds it), detail %.2f / %.2f, weather %.2f / %.2f" % (
    tex["large_psnr_level0"], tex1["large_psnr_level0"], "%.2f" % ideal_psnr if ideal_psnr else "not computed", tex["small_psnr_level0"], tex1["small_psnr_level0"], tex["weather_psnr"], tex1["weather_psnr"]))
print(json.dumps(out))
