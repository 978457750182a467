#!/usr/bin/env python
"""bc7_sensitivity.py -- what compress/mode=2 of the reference's *.import files does to a frame (GPU box).

The reference's samplers return DECODED BC7 blocks (weather.bmp.import:19-20, worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20); this
build marches the uncompressed bytes by default.  The engine's encoder cannot be reproduced, so its exact texels are unknown; this tool puts a
number on the SIZE of the effect with the library's own encoder (csky_encode_bc7, modes 6 / 1 / 5) in the importer's place: the C3 frame from the
uncompressed inputs next to the frame from the same inputs after encode -> decode of every mip level (assets.vram_compressed_chains), per texture and
all three together.  A weaker encoder than the engine's overstates the effect, so read the figures as an upper estimate of its order of magnitude:
it is what "bit-identical to the oracle" has to be weighed against when the question is "identical to the reference"."""
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
lraw, sraw = assets.build_mips(large, 8), assets.build_mips(small, 6)


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
out = {"frame": "%dx%d, 128 x 6 steps, sun (1,1,0), default parameters" % (W, H), "texture_round_trip_psnr_db": tex,
       "uncompressed_vs_box_chain_through_set_noise_mips": "the uncompressed frame is rendered through the same csky_set_noise_mips path", "cases": {}}
for name, (lc, sc, w) in (("weather map only", (lraw, sraw, wq)), ("detail volume only", (lraw, sq, weather)), ("shape volume only", (lq, sraw, weather)), ("all three", (lq, sq, wq))):
    img, n, inexact = frame(lc, sc, w)
    r = compare(base, img)
    r["in_cloud_samples"] = [int(n0), int(n)]; r["cells_beyond_fp16_pairs"] = int(inexact)
    out["cases"][name] = r
    print("%-20s PSNR %.1f dB  mean |d| %.2e (%.2f %% of the value)  max |d| %.3g  within 1 / 16 fp16 steps: %.3f / %.3f  in-cloud samples %d -> %d"
          % (name, r["psnr_rgb_db_vs_peak"], r["mean_abs_rgb"], 100 * r["mean_rel_rgb"], r["max_abs_rgb"], r["values_within_1_fp16_step"], r["values_within_16_steps"], n0, n), flush=True)
print(json.dumps(out))
