"""-m gpu, round 5: the tunable stand-in generator on the device, the bench line's two protocols (tests/test_bench_contract.py), the public / internal
header split (tests/test_abi.py), the instruction diet's parity (the whole-frame gates of rounds 2-4 run on the kernels as they are now)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tuned_shape_generator_on_the_gpu_is_byte_identical_to_the_host(pkg, gpu_ctx):
    """csky_generate_shape_noise_tuned_device == csky_generate_shape_noise_tuned for the settings tools/demo_scene.py sweeps (noise_core.h is one source
    for both: integer hashing + IEEE +,-,*,/,sqrt, contraction off)."""
    for knobs in (dict(perlin_freq=8, perlin_octaves=4, dilate=0.8), dict(worley_freq=8, contrast=2.5, offset=0.42)):
        host = pkg.assets.generate_shape_noise(1, 128, **knobs)
        dev = gpu_ctx.generate_shape_noise(1, 128, **knobs)
        assert (host == dev).all(), knobs
    with pytest.raises(pkg.CloudSkyError):
        gpu_ctx.generate_shape_noise(1, 128, perlin_freq=64, perlin_octaves=5)
