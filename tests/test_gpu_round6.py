"""-m gpu, round 6: the HIP path through the C ABI against the reference's own shader text executed on the CPU
(tests/golden/glslexec.npz -- see tests/test_oracle_glslexec.py for what that fixture is), plus the round's new rehearsals.
Gates are the ones the HIP-vs-oracle tests use (the oracle is bit-identical to the fixture, tests/test_oracle_glslexec.py):
LUTs <= 1 fp16 ulp, cloud frames `cloud_tight`."""
import numpy as np
import pytest

from conftest import SUNS, cloud_tight, norm, ulp_diff
from glslexec_fixture import GlslExec, SKY_OF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gx():
    return GlslExec()


def test_luts_vs_executed_shader_text(gpu_ctx, gx):
    t = gpu_ctx.render_transmittance(256, 64)
    d = ulp_diff(t, gx.fold("trans"))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    suns = {k: norm(s) for k, s in SUNS.items()}
    suns["windy"] = gx.z["windy_params"][16:19]
    suns["below"] = norm((0.3, -0.2, 0.5))
    for k in gx.extra:
        if k not in SKY_OF:
            suns[k] = gx.z[k + "_params"][16:19]
    for k, s in suns.items():
        d = ulp_diff(gpu_ctx.render_sky_lut(s, 200, 100), gx.fold("sky_" + k))
        assert d.max() <= 1 and (d > 0).mean() < 0.02, (k, d.max(), (d > 0).mean())


@pytest.mark.parametrize("variant", [-1, 0])
def test_cloud_frames_vs_executed_shader_text(gpu_ctx, gx, variant):
    """Every cloud fixture (three default-config suns, heavy cover, low sun, the windy offset tile, two random push-constant
    blocks) rendered by the default kernel and by the lock-step variant; the context renders its own LUTs first, as a host does."""
    gpu_ctx.set_variant(variant)
    gpu_ctx.set_march(128, 6)
    gpu_ctx.set_early_out(0.0)
    gpu_ctx.render_transmittance(256, 64)
    worst = {}
    for k, (pc, rect, sky) in gx.cloud_cases(SUNS).items():
        pc = np.array(pc, np.float32)
        gpu_ctx.render_sky_lut(pc[16:19], 200, 100)
        gx0, gy0, w, h = rect
        p = pc.copy()
        p[2] += gx0; p[3] += gy0                                         # the rectangle's origin as update_position (clouds.glsl:260)
        img = gpu_ctx.render_clouds(p, w, h)
        ok, info = cloud_tight(img, gx.fold("clouds_" + k))
        assert ok, (k, info)
        worst[k] = (info["within0"], info["max_ulp"])
    gpu_ctx.set_variant(-1)
    assert min(v[0] for v in worst.values()) > 0.98, worst
