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


def test_lut_rows_beside_the_march_give_the_same_rows_and_the_same_bands(gpu_ctx, oracle):
    """csky_set_lut_rows_overlap: a rank's rows of the sky LUT run on a side stream beside the march that follows them on the caller's stream, which
    is ordered behind them after the march is enqueued.  Eight frames rotating over four streams (rows + bands each, a copy behind every march on
    its stream standing in for the gather): rows and bands are byte-identical to the in-order form, and a rows call that no march follows is
    ordered by csky_sync."""
    import torch
    W, H, n = 512, 256, 8
    bands = (8, 1, n, H // 8 // n)
    suns = [np.array([np.cos(t), np.sin(t), 0.2], np.float32) / np.float32(np.sqrt(1.04)) for t in np.linspace(0.3, 2.6, 8)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    nrow = (100 - 1 + n - 1) // n

    def run(overlap):
        gpu_ctx.set_lut_rows_overlap(overlap)
        gpu_ctx.set_frames_in_flight(4)
        rows = [torch.zeros(nrow * 200 * 8, dtype=torch.uint8, device="cuda") for _ in range(4)]
        out = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(4)]
        got = []
        torch.cuda.synchronize()
        for k, sun in enumerate(suns):
            i = k % 4
            gpu_ctx.render_sky_lut_rows_device(sun, 1, n, rows[i].data_ptr(), rows[i].numel(), 200, 100, streams[i].cuda_stream)
            gpu_ctx.render_clouds_device(oracle.default_params(W, H, sun), W, bands, out[i].data_ptr(), W * 8, streams[i].cuda_stream)
            with torch.cuda.stream(streams[i]):                  # the "gather": reads rows and bands in stream order behind the march
                got.append((rows[i].clone(), out[i].clone()))
        for s in streams:
            s.synchronize()
        lone = torch.zeros(nrow * 200 * 8, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu_ctx.render_sky_lut_rows_device(suns[3], 1, n, lone.data_ptr(), lone.numel(), 200, 100, streams[0].cuda_stream)
        gpu_ctx.sync()                                           # no march follows: csky_sync orders the side stream
        streams[0].synchronize()
        gpu_ctx.set_frames_in_flight(1)
        return got, lone.cpu()

    try:
        a, la = run(False)
        b, lb = run(True)
    finally:
        gpu_ctx.set_lut_rows_overlap(False)
        gpu_ctx.render_sky_lut(suns[0], 200, 100)                # leave the shared context with a whole LUT
    for (ra, oa), (rb, ob) in zip(a, b):
        assert bool((ra == rb).all().item()) and bool((oa == ob).all().item())
    assert bool((la == lb).all().item()) and bool((la == a[3][0].cpu()).all().item()) and int(la.sum().item()) > 0
    assert float(a[0][1].view(torch.float16)[..., 3].float().mean().item()) > 0.0
