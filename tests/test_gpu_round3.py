"""-m gpu, round 3: the "compact-ilp" kernel (two primary steps / two light samples in flight per wavefront, the kernel of small launches),
the C4 workload in its real shape on one GPU, the grazing end points of the C5 sun sweep at full size, importer-style (non-box) mip
chains rendered against the oracle, and the asynchronous submit/collect pair of the host form.  Gates as in tests/test_gpu_round2.py
(`cloud_tight`: >= 99.99 % of pixels with every channel within 2 fp16 ulp-equivalents, max |d| <= 2e-3, PSNR >= 70 dB)."""
import os

import numpy as np
import pytest

from conftest import SUNS, cloud_close, cloud_tight, norm

pytestmark = pytest.mark.gpu

ILP = 4   # csky_variant_name(4) == "compact-ilp"


def test_ilp_variant_is_registered(pkg):
    L = pkg.lib()
    assert L.csky_variant_count() >= 5 and L.csky_variant_name(ILP) == b"compact-ilp"


@pytest.mark.parametrize("sun_name", list(SUNS))
def test_ilp_variant_vs_oracle(gpu_ctx, oracle, otex, o_skies, sun_name):
    """kernels.hip::march_compact_ilp against the oracle at the tight gate, in-cloud counts included (clouds.glsl:172-212)."""
    sun = SUNS[sun_name]
    gpu_ctx.set_march(128, 6); gpu_ctx.set_early_out(0.0); gpu_ctx.set_variant(ILP)
    try:
        gpu_ctx.render_sky_lut(norm(sun), 200, 100)
        p = oracle.default_params(256, 128, sun)
        img = gpu_ctx.render_clouds(p)
        st = gpu_ctx.cloud_stats()
        ref, st_o = oracle.clouds(otex, p, o_skies[sun_name], nthreads=oracle.max_threads(), return_stats=True)
        ok, info = cloud_tight(img, ref)
        assert ok, info
        assert abs(int(st["incloud_samples"]) - st_o["incloud_samples"]) <= 1e-3 * st_o["incloud_samples"]
        assert st["primary_samples"] == st_o["primary_samples"]
        f = img.astype(np.float32)
        assert (f[0] == 0).all() and (f[:, 0] == 0).all()
    finally:
        gpu_ctx.set_variant(-1)


@pytest.mark.parametrize("steps", [(128, 6), (64, 4), (127, 5), (33, 3), (16, 1), (9, 0), (1, 6)])
def test_ilp_variant_matches_compact_for_every_march_shape(gpu_ctx, oracle, steps):
    """Odd and even primary step counts (the last round of a ray takes one step instead of two) and odd / even / zero light steps (the odd
    cone sample is paired with the distant one, clouds.glsl:186-199): same samples, same counts as the compact kernel, frames equal to
    rounding (the two bodies may be contracted differently), under the static and the cost-feedback order, ragged frame."""
    ps, ls = steps
    sun = (1, 1, 0)
    gpu_ctx.render_sky_lut(norm(sun), 200, 100)
    gpu_ctx.set_march(ps, ls)
    p = oracle.default_params(200, 72, sun)
    try:
        gpu_ctx.set_variant(3); gpu_ctx.set_segments(1)
        ref = gpu_ctx.render_clouds(p); st_ref = gpu_ctx.cloud_stats()
        gpu_ctx.set_segments(0); gpu_ctx.set_variant(ILP)
        first = None
        for sch in (-1, 5, 2, 7, 7):
            gpu_ctx.set_schedule(sch)
            img = gpu_ctx.render_clouds(p)
            assert gpu_ctx.cloud_stats() == st_ref, (steps, sch, gpu_ctx.cloud_stats(), st_ref)
            ok, info = cloud_close(img, ref, frac=0.9999, atol=5e-4, rtol=2e-3)
            assert ok, (steps, sch, info)
            if first is None:
                first = img.view(np.uint16).copy()
            assert (img.view(np.uint16) == first).all(), (steps, sch)       # the order of the workgroups never changes a bit
    finally:
        gpu_ctx.set_variant(-1); gpu_ctx.set_schedule(-1); gpu_ctx.set_segments(0); gpu_ctx.set_march(128, 6)


def test_ilp_variant_early_out_is_bounded(gpu_ctx, oracle):
    sun = (1, 1, 0)
    gpu_ctx.render_sky_lut(norm(sun), 200, 100)
    gpu_ctx.set_march(128, 6)
    p = oracle.default_params(256, 128, sun, coverage=0.5)
    try:
        gpu_ctx.set_variant(ILP)
        ref = gpu_ctx.render_clouds(p).astype(np.float32)
        gpu_ctx.set_early_out(1e-3)
        img = gpu_ctx.render_clouds(p).astype(np.float32)
        assert np.abs(img - ref).max() <= 4e-3
    finally:
        gpu_ctx.set_early_out(0.0); gpu_ctx.set_variant(-1)
