"""Parity metrics shared by the -m gpu tests and tools/parity_stats.py (test infrastructure, not product).

The cloud path stores RGBA16F, so the natural unit of disagreement is the fp16 ulp of the reference value:
ulp16(v) = 2^(floor(log2 |v|) - 10) for normal halfs (|v| >= 2^-14) and 2^-24 below.  `cloud_ulp_stats` expresses
|test - ref| in those units per value (an "ulp-equivalent": it also covers sign changes and values that straddle a
binade, where bit-pattern distance would mislead)."""
import numpy as np


def ulp16(v):
    """Spacing of fp16 numbers at |v| (float64 array)."""
    a = np.abs(np.asarray(v, np.float64))
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -14)))
    return 2.0 ** (e - 10)


# fp16 resolves values below 1e-3 more finely (ulp <= 1e-6, 6e-8 below 2^-14) than the march's fp32 arithmetic can deliver them: every
# in-cloud sample adds (1 - dt) (1 - alpha) to alpha and T (r - r dt) / t to L (clouds.glsl:207,209) with dt = exp(-density * step) -- for a
# thin cloud dt is within 1e-3 of 1, so `1 - dt` carries dt's rounding, a few fp32 ulps OF 1.0 (6e-8 each), whatever its own size; and the
# hardware exp2/log2 the kernel uses (as a GPU's GLSL compiler does) differ from the oracle's libm by an ulp per call.  2^-18 = 64 fp32 ulps
# of 1.0 is the floor below which a disagreement is that cancellation, not the algorithm.  Only frames that are mostly tiny values (a nearly
# clear sky: profiles/r03/parity_sweep.txt, coverage 0.05) are gated with it -- `cloud_tight(..., sparse=True)`; all other frames keep the
# plain 2-ulp count.
ABS_FLOOR = 2.0 ** -18


def cloud_ulp_stats(test, ref):
    """test, ref: float16 [h, w, 4] frames.  Returns a dict of what the tightened gates are written against."""
    a, b = np.asarray(test, np.float32).astype(np.float64), np.asarray(ref, np.float32).astype(np.float64)
    finite = bool(np.isfinite(a).all())
    err = np.abs(a - b)
    u = err / ulp16(b)
    mse = float(((a[..., :3] - b[..., :3]) ** 2).mean())
    peak = max(float(b[..., :3].max()), 1e-6)
    psnr = float(10 * np.log10(peak * peak / max(mse, 1e-30)))
    bad_px = (u > 2.0).any(-1)                                  # pixels with any channel beyond 2 ulp-equivalents
    bad_px_floor = ((u > 2.0) & (err > ABS_FLOOR)).any(-1)      # ... and beyond what fp32 intermediates can resolve (ABS_FLOOR below)
    return dict(finite=finite, n=int(a.size), max_err=float(err.max()), max_ulp=float(u.max()),
                within0=float((u == 0).mean()), within1=float((u <= 1.0).mean()), within2=float((u <= 2.0).mean()),
                beyond2_values=int((u > 2.0).sum()), beyond2_pixels=int(bad_px.sum()), beyond2_above_floor_pixels=int(bad_px_floor.sum()), psnr=psnr,
                alpha_mean=float(a[..., 3].mean()))


# The tightened cloud gate, written against what the kernel achieves on MI355X (profiles/r02/parity_stats.txt,
# profiles/r02/parity_log_all_gpu_tests.txt): full C3 frames have 99.84 % of values bit-identical, 99.993 % within 1 fp16 ulp and
# ~100 of 2 097 152 pixels with a channel beyond 2 ulp-equivalents -- the pixels where the t > 0 branch (clouds.glsl:184) flips for a
# sample whose density is within rounding of zero (their absolute error stays below 1e-4: many ulps of a small value).
TIGHT = dict(bad_pixel_frac=1e-4, bad_pixel_floor=2, max_err=2e-3, psnr=70.0, within1=0.999)


def cloud_tight(test, ref, bad_pixel_frac=TIGHT["bad_pixel_frac"], max_err=TIGHT["max_err"], psnr=TIGHT["psnr"], sparse=False):
    """>= 99.99 % of the PIXELS have every channel within 2 fp16 ulp-equivalents of the oracle (small frames: at most 2 pixels beyond,
    the branch-flip allowance), >= 99.9 % of values within 1, every value within 2e-3 absolute, PSNR >= 70 dB.  Returns (ok, stats)."""
    s = cloud_ulp_stats(test, ref)
    pixels = s["n"] // 4
    allowed = max(TIGHT["bad_pixel_floor"], int(bad_pixel_frac * pixels))
    beyond = s["beyond2_above_floor_pixels"] if sparse else s["beyond2_pixels"]
    ok = (s["finite"] and beyond <= allowed and s["within1"] >= TIGHT["within1"] and s["max_err"] <= max_err and s["psnr"] >= psnr)
    s["allowed_bad_pixels"] = allowed
    return ok, s
