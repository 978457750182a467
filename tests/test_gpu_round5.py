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


def test_external_frame_import_and_release_leave_no_file_descriptor_behind(pkg, gpu_ctx):
    """csky_external_frame_import_fd hands the runtime a DUPLICATE of the caller's fd and closes the caller's on success; whoever ends up
    owning the duplicate, sixteen import / release cycles must not leave sixteen descriptors open in the process."""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ext_frame_roundtrip as X
    hip = X.load_hip()
    try:
        ex = X.ExportedAllocation(hip, 0, 1 << 20)
    except RuntimeError as e:
        pytest.skip("the runtime cannot export an allocation as a file descriptor here: %s" % e)
    L = pkg.lib()
    try:
        def cycle():
            ef, d = C.c_void_p(), C.c_void_p()
            assert L.csky_external_frame_import_fd(gpu_ctx._h, os.dup(ex.fd), C.c_size_t(ex.size), C.c_size_t(0), C.c_size_t(1 << 20), C.byref(ef), C.byref(d)) == 0
            L.csky_external_frame_release(ef)
        cycle()                                                  # (the first import may open driver files that stay open)
        before = len(os.listdir("/proc/self/fd"))
        for _ in range(16):
            cycle()
        after = len(os.listdir("/proc/self/fd"))
        assert after - before < 4, "%d descriptors before, %d after sixteen import / release cycles" % (before, after)
    finally:
        ex.close()
