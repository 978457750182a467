#!/usr/bin/env python
"""Whole-frame parity sweep, HIP (C ABI) vs the CPU oracle, at BASELINE's C3 size (2048x1024, 128 x 6 steps) over the push-constant
block's degrees of freedom: cloud coverage and density (cloud_sky.gd:22-26), the three wind-integrated positions (cloud_sky.gd:176-187),
light energy / colour (cloud_sky.gd:76-79), ground colour, sun elevation from below the horizon to the zenith.  The default-parameter
frames are tools/parity_stats.py's; the small-frame fuzz is tests/test_gpu_parity.py::test_fuzz_parameters_vs_oracle -- this is the
same question at the headline size (GPU box; the oracle renders on all granted host cores, ~5 s per frame)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gvcd_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402
from parity_metrics import cloud_tight, ulp16  # noqa: E402
from bench import usable_cores  # noqa: E402

W, H, PRIM, LIGHT = 2048, 1024, 128, 6


def params(sun, coverage=0.2, density=0.05, cloud_pos=(0, 0), detailed_pos=(0, 0), weather_pos=(0, 0), energy=1.0, colour=(1, 1, 1), ground=None, t=0.0):
    p = O.default_params(W, H, sun, coverage=coverage, density=density)
    p[4:6], p[6:8], p[8:10] = cloud_pos, detailed_pos, weather_pos
    p[20:23] = colour
    p[19] = energy
    p[23] = t
    if ground is not None:
        p[12:15] = ground
    return p


# name, sun, kwargs
CASES = [
    ("coverage 0.35", (1, 1, 0), dict(coverage=0.35)),
    ("coverage 0.5 density 0.1", (1, 1, 0), dict(coverage=0.5, density=0.1)),
    ("coverage 0.8 (overcast)", (0.3, 1, 0.2), dict(coverage=0.8)),
    ("coverage 0.05 (nearly clear)", (1, 1, 0), dict(coverage=0.05)),
    ("density 0.01 (thin)", (1, 1, 0), dict(density=0.01)),
    ("wind: 10 min at wind_speed 1", (1, 1, 0), dict(cloud_pos=(600.0, 0.0), detailed_pos=(600.0, 0.0), weather_pos=(0.6, 0.0), t=600.0)),
    ("wind: diagonal, 3 h", (-0.5, 0.4, 0.7), dict(cloud_pos=(7636.75, 7636.75), detailed_pos=(7636.75, 7636.75), weather_pos=(7.63675, 7.63675), coverage=0.3, t=10800.0)),
    ("sun 2 degrees above the horizon", (np.cos(np.radians(2.0)), np.sin(np.radians(2.0)), 0.0), dict(coverage=0.3)),
    ("sun 5 degrees BELOW the horizon", (0.0, -np.sin(np.radians(5.0)), np.cos(np.radians(5.0))), dict(coverage=0.3)),
    ("light energy 3, warm colour, green ground", (0.2, 0.6, -0.7), dict(energy=3.0, colour=(1.0, 0.7, 0.4), ground=(0.1, 0.5, 0.1), coverage=0.25)),
]


def main():
    large, small, weather = gvcd_amd.assets.load_default_noise()
    ctx = gvcd_amd.Context(0)
    ctx.set_noise(large, small, weather)
    ctx.render_transmittance(256, 64)
    ctx.set_march(PRIM, LIGHT)
    tex = O.OracleTextures(large, small, weather)
    tr = O.transmittance_lut(256, 64)
    cores = max(1, min(O.max_threads(), usable_cores()))
    worst = dict(beyond2_pixels=0, max_err=0.0, within1=1.0, psnr=1e9)
    all_ok = True
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, sun, kw in CASES:
        if only not in name:
            continue
        p = params(sun, **kw)
        s = p[16:19].copy()
        ctx.render_sky_lut(s, 200, 100)
        img = ctx.render_clouds(p)
        st = ctx.cloud_stats()
        t0 = time.perf_counter()
        ref, st_o = O.clouds(tex, p, O.sky_lut(s, tr, 200, 100), primary_steps=PRIM, light_steps=LIGHT, nthreads=cores, return_stats=True)
        dt = time.perf_counter() - t0
        sparse = kw.get("coverage", 0.2) < 0.1            # mostly values below 1e-3: counted above parity_metrics.ABS_FLOOR
        ok, d = cloud_tight(img, ref, sparse=sparse)
        d["gate"] = "sparse (2 ulp and 2^-18 absolute)" if sparse else "2 ulp"
        all_ok &= bool(ok)
        d.update(case=name, tight_gate=bool(ok), oracle_s=round(dt, 2), incloud_gpu=int(st["incloud_samples"]), incloud_oracle=int(st_o["incloud_samples"]),
                 primary_gpu=int(st["primary_samples"]), primary_oracle=int(st_o["primary_samples"]))
        # where the values beyond 2 ulp-equivalents sit: per channel, per magnitude of the reference value, and their absolute error
        a, b = img.astype(np.float64), ref.astype(np.float64)
        err = np.abs(a - b)
        bad = err / ulp16(b) > 2.0
        d["beyond2_by_channel"] = [int(bad[..., c].sum()) for c in range(4)]
        edges = [0.0, 2.0 ** -14, 1e-3, 1e-2, 1e-1, 1e9]
        d["beyond2_by_magnitude"] = {"%g..%g" % (lo, hi): int((bad & (np.abs(b) >= lo) & (np.abs(b) < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])}
        d["beyond2_max_abs_err"] = float(err[bad].max()) if bad.any() else 0.0
        d["beyond2_abs_err_quantiles"] = [float(q) for q in np.quantile(err[bad], [0.5, 0.9, 0.99])] if bad.any() else []
        worst["beyond2_pixels"] = max(worst["beyond2_pixels"], d["beyond2_pixels"])
        worst["max_err"] = max(worst["max_err"], d["max_err"])
        worst["within1"] = min(worst["within1"], d["within1"])
        worst["psnr"] = min(worst["psnr"], d["psnr"])
        print(json.dumps(d), flush=True)
    print(json.dumps(dict(summary="worst over %d whole C3 frames" % len(CASES), all_pass_tight_gate=all_ok, cores=cores, **worst)), flush=True)
    ctx.close()
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
