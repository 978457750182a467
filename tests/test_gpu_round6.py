"""-m gpu, round 6: the HIP path through the C ABI against the reference's own shader text executed on the CPU
(tests/golden/glslexec.npz -- see tests/test_oracle_glslexec.py for what that fixture is), plus the round's new rehearsals.
Gates are the ones the HIP-vs-oracle tests use (the oracle is bit-identical to the fixture, tests/test_oracle_glslexec.py):
LUTs <= 1 fp16 ulp, cloud frames `cloud_tight`."""
import numpy as np
import pytest

from conftest import SUNS, cloud_tight, norm, ulp_diff
from glslexec_fixture import COMPOSITES, GlslExec, SKY_OF

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


def test_compositor_vs_executed_shader_text(gpu_ctx, gx):
    """csky_composite_sky against clouds.gdshader's own sky() executed per panorama pixel: the gate of the HIP-vs-oracle compositor
    test, <= 2 fp16 ulp.  The context's own transmittance LUT is bit-identical to the fixture's (test_luts_vs_executed_shader_text)."""
    gpu_ctx.render_transmittance(256, 64)
    for k, c in COMPOSITES.items():
        w, h = c["size"]
        img = gpu_ctx.composite_sky(gx.fold("clouds_" + c["from"]), gx.fold("clouds_" + c["to"]), gx.fold("sky_" + c["from"]), gx.fold("sky_" + c["to"]),
                                    norm(SUNS[c["sun"]]), c["blend"], c["disk"], w, h)
        d = ulp_diff(img, gx.fold("composite_" + k))
        assert d.max() <= 2 and (d > 0).mean() < 0.02, (k, d.max(), (d > 0).mean())


def test_multi_handle_reports_its_preconditions_and_times_its_devices(pkg, noise, oracle):
    """csky_multi proves its preconditions and measures itself.  n = 2 and n = 8 contexts on device 0: every device reports peer
    access, the in-place form is in use, and with csky_multi_set_timing every device's march of the last frame is timed (copy_ms 0: the stores ARE
    the march); in the staged form the peer copy is timed too."""
    p = oracle.default_params(512, 256, (1.0, 1.0, 0.0))
    for n in (2, 8):
        m = pkg.MultiContext([0] * n)
        try:
            assert m.last_warning() == ""
            m.set_noise(*noise)
            m.set_timing(True)
            m.render_sky_lut(norm((1.0, 1.0, 0.0)))
            a = m.render_clouds(p)
            st = m.stats()
            assert st["n_devices"] == n and st["all_peer"] and not st["staged"] and st["timing"] and st["peer_access"] == [1] * n
            assert all(0.0 < x < 50.0 for x in st["march_ms"]) and st["copy_ms"] == [0.0] * n, st
            m.set_staged(True)
            b = m.render_clouds(p)
            st = m.stats()
            assert st["staged"] and all(0.0 < x < 50.0 for x in st["march_ms"]) and st["copy_ms"][0] == 0.0 and all(0.0 < x < 50.0 for x in st["copy_ms"][1:]), st
            assert (a.view(np.uint16) == b.view(np.uint16)).all()
        finally:
            m.close()


def test_multi_handle_falls_back_when_a_device_has_no_peer_access(pkg, noise, oracle, monkeypatch):
    """A device that cannot store into the first device's memory (faked here for device INDEX 2: CSKY_MULTI_FAKE_NO_PEER) no longer fails
    csky_multi_create: the handle switches to staged copies + a whole sky LUT on the first device, says so in csky_multi_last_warning, refuses to
    switch the staged form off, and renders the same frame."""
    p = oracle.default_params(512, 256, (1.0, 1.0, 0.0))
    ref = pkg.MultiContext([0, 0, 0, 0])
    try:
        ref.set_noise(*noise)
        ref.render_sky_lut(norm((1.0, 1.0, 0.0)))
        want = ref.render_clouds(p)
        want_lut = ref.ctx(0).read_sky_lut()
    finally:
        ref.close()
    monkeypatch.setenv("CSKY_MULTI_FAKE_NO_PEER", "2")
    m = pkg.MultiContext([0, 0, 0, 0])
    try:
        w = m.last_warning()
        assert "index 2" in w and "no peer access" in w and "staged" in w, w
        st = m.stats()
        assert st["peer_access"] == [1, 1, 0, 1] and not st["all_peer"] and st["staged"]
        with pytest.raises(pkg.CloudSkyError):
            m.set_staged(False)
        m.set_noise(*noise)
        m.render_sky_lut(norm((1.0, 1.0, 0.0)))
        got = m.render_clouds(p)
        assert (got.view(np.uint16) == want.view(np.uint16)).all()
        assert (m.ctx(0).read_sky_lut().view(np.uint16) == want_lut.view(np.uint16)).all()
    finally:
        m.close()


def test_shape_generator_on_the_device_refuses_what_the_host_refuses(gpu_ctx, pkg):
    """the n-independent parameter bounds guard the device twin too (perlin_freq = 0 at n = 32 used to reach `i % 0` inside the kernel)."""
    for n, bad in ((32, dict(perlin_freq=0)), (16, dict(worley_freq=0)), (64, dict(perlin_freq=1 << 30, perlin_octaves=3)), (32, dict(perlin_octaves=1 << 20)),
                   (32, dict(contrast=float("inf")))):
        with pytest.raises(pkg.CloudSkyError):
            gpu_ctx.generate_shape_noise(1, n, **bad)
    assert (gpu_ctx.generate_shape_noise(3, 32) == pkg.assets.generate_shape_noise(3, 32)).all()


def test_rccl_calls_of_the_multi_rank_path_in_a_world_of_one():
    """No second GPU exists here, so RCCL cannot move a byte between devices -- but every torch.distributed call bench.py's N > 1 path makes can at least be
    issued on the real backend (nccl = RCCL) with the dtypes, streams and options it uses: init with device_id, the identity all_gather_object, new_group,
    the asynchronous byte gather of FrameGroups on a side stream with wait() under that stream, all_reduce MAX of the float64 timing, all_reduce SUM of the
    share vector, barrier.  A call the backend rejects (dtype, async gather, object collectives) fails here instead of on the driver's 8-GPU node."""
    import os
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent('''
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from gvcd_amd import tiling
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(0); dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        ids = [None]; dist.all_gather_object(ids, "host|pci 0000:0a:00"); assert ids == ["host|pci 0000:0a:00"]
        fg = tiling.FrameGroups(0, 1, 1, dist)
        sub = dist.new_group([0])
        side = torch.cuda.Stream(device=dev)
        src = torch.arange(4096, dtype=torch.int16, device=dev).view(torch.uint8)
        gathered = torch.empty((1, src.numel()), dtype=torch.uint8, device=dev)
        with torch.cuda.stream(side):
            w = fg.gather(0, src, gathered, async_op=True)
            w.wait()
        side.synchronize()
        assert (gathered[0] == src).all()
        dist.gather(src, gather_list=[gathered[0]], dst=0, group=sub)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.25
        s = torch.zeros(1, dtype=torch.float64, device=dev); s[0] = 0.5; dist.all_reduce(s, op=dist.ReduceOp.SUM); assert float(s.item()) == 0.5
        dist.barrier(); torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rccl world-of-one ok")
    ''') % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl world-of-one ok" in r.stdout, (r.stdout[-800:], r.stderr[-2500:])


def test_benchmark_frame_hip_vs_oracle_vs_executed_text(gpu_ctx, gx, oracle, oracle_frames):
    """The chain at full benchmark size (BASELINE configs[2], 2 097 152 rays): the frame the reference's own shader text wrote (committed as hashes) IS the
    oracle's frame, and the HIP frame of the same push constants is within the tight gate of it -- one test, the same arrays."""
    import hashlib
    for sk, key in (("zenith", "c3_zenith_sha256"), ("demo", "c3_demo_sha256")):      # the two parity frames of SURVEY 8(d): the oracle IS the executed text there too
        fr, _ = oracle_frames(2048, 1024, sk)
        assert hashlib.sha256(np.ascontiguousarray(fr).view(np.uint16).tobytes()).hexdigest() == str(gx.z[key]), sk
    ref, st_o = oracle_frames(2048, 1024, "deg45")
    assert hashlib.sha256(np.ascontiguousarray(ref).view(np.uint16).tobytes()).hexdigest() == str(gx.z["c3_sha256"])
    gpu_ctx.set_variant(-1); gpu_ctx.set_march(128, 6); gpu_ctx.set_early_out(0.0)
    gpu_ctx.render_transmittance(256, 64)
    gpu_ctx.render_sky_lut(norm(SUNS["deg45"]), 200, 100)
    img = gpu_ctx.render_clouds(oracle.default_params(2048, 1024, SUNS["deg45"]))
    ok, info = cloud_tight(img, ref)
    assert ok, info
    assert abs(int(gpu_ctx.cloud_stats()["incloud_samples"]) - int(st_o["incloud_samples"])) <= 64
