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


def test_two_frames_of_one_allocation_survive_each_others_release(pkg, gpu_ctx):
    """Two external frames imported from duplicates of ONE exported descriptor (a ring of textures inside one VkDeviceMemory): releasing the first must
    not close anything the second -- or the caller -- still uses, whichever party the runtime makes the owner of the duplicate it is handed."""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ext_frame_roundtrip as X
    from bench import default_params
    hip = X.load_hip()
    W, H = 128, 64
    frame = W * H * 8
    try:
        ex = X.ExportedAllocation(hip, 0, 4 * frame)
    except RuntimeError as e:
        pytest.skip("the runtime cannot export an allocation as a file descriptor here: %s" % e)
    L = pkg.lib()
    try:
        gpu_ctx.set_variant(-1); gpu_ctx.set_schedule(-1); gpu_ctx.set_segments(0); gpu_ctx.set_early_out(0.0); gpu_ctx.set_march(64, 4)
        p, sun = default_params(W, H, (1, 1, 0))
        gpu_ctx.render_sky_lut(sun, 200, 100)
        ref = gpu_ctx.render_clouds(p).view(np.uint16)
        ex.fill(0)
        e1, d1, e2, d2 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert L.csky_external_frame_import_fd(gpu_ctx._h, os.dup(ex.fd), C.c_size_t(ex.size), C.c_size_t(0), C.c_size_t(frame), C.byref(e1), C.byref(d1)) == 0
        keep = os.dup(ex.fd)                                     # a descriptor of the same object the caller holds on to (may take a number the runtime freed)
        assert L.csky_external_frame_import_fd(gpu_ctx._h, os.dup(ex.fd), C.c_size_t(ex.size), C.c_size_t(2 * frame), C.c_size_t(frame), C.byref(e2), C.byref(d2)) == 0
        L.csky_external_frame_release(e1)
        os.fstat(keep)                                           # still open: the release closed nothing of the caller's
        gpu_ctx.render_clouds_device(p, W, (H, 0, 1, 1), d2.value, W * 8, 0)
        gpu_ctx.sync()
        assert (ex.read(frame, 2 * frame).view(np.uint16).reshape(H, W, 4) == ref).all()
        assert (ex.read(frame, 0) == 0).all()
        L.csky_external_frame_release(e2)
        os.fstat(keep); os.close(keep)
    finally:
        ex.close()
