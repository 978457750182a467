"""-m gpu, round 4: the GPU noise bake against the fixture of an independent restatement (oracle/noise_restatement.py ->
tests/golden/noise_fixture.npz), the exact fp32-coefficient cells at headline size, the multi handle's sky LUT on every
group, csky_last_warning."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, SUNS, cloud_tight, norm

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("seed", [1, 7])
def test_gpu_noise_bake_matches_the_independent_restatement(gpu_ctx, seed):
    """shape_noise_kernel / detail_noise_kernel (kernels.hip; README.md:30 TODO 3, perlworlnoise.tga.import:24-27 for the layout) against hashes
    and blocks rendered by numpy, not by noise_core.h: round 3 compared the header on gfx950 with the same header on x86."""
    fix = np.load(os.path.join(GOLDEN, "noise_fixture.npz"))
    vol = gpu_ctx.generate_shape_noise(seed, 128)
    assert vol.shape == (128, 128, 128, 4) and sha(vol) == str(fix["shape_sha256_seed%d" % seed])
    det = gpu_ctx.generate_detail_noise(seed, 32)
    assert det.shape == (32, 32, 32, 3) and sha(det) == str(fix["detail_sha256_seed%d" % seed])
    if seed == 1:
        assert np.array_equal(vol[8:24, 72:88, 40:56], fix["shape_block_z8_y72_x40"])
        assert np.array_equal(vol[120:128, 112:128, 0:16], fix["shape_block_z120_y112_x0"])
        assert np.array_equal(det[0:16, 8:24, 16:32], fix["detail_block_z0_y8_x16"])


def test_exact_cells_at_headline_size_are_the_same_frame(pkg, noise):
    """csky_set_exact_cells(1): the whole C3 frame marched on fp32-coefficient cells is bit-identical to the fp16-cell frame (the shipped textures
    fit fp16, so both hold the same numbers; whole-ray compact kernel both times) -- the exact path IS the product's filter, only wider."""
    sun = norm(SUNS["deg45"])
    frames = []
    for mode in (0, 1):
        ctx = pkg.Context(0)
        try:
            ctx.set_exact_cells(mode)
            ctx.set_noise(*noise)
            ctx.set_segments(1)
            ctx.render_transmittance(256, 64)
            ctx.render_sky_lut(sun, 200, 100)
            from oracle import oracle as O
            frames.append((ctx.render_clouds(O.default_params(2048, 1024, SUNS["deg45"])).view(np.uint16), ctx.cloud_stats()["incloud_samples"]))
        finally:
            ctx.close()
    assert frames[0][1] == frames[1][1] and np.array_equal(frames[0][0], frames[1][0])


def test_warnings_do_not_sit_in_the_error_slot(pkg, noise):
    """csky_set_frames_in_flight / csky_set_noise succeed with a caveat: the text is in csky_last_warning, csky_last_error stays clean."""
    L = pkg.lib()
    ctx = pkg.Context(0)
    try:
        ctx.set_noise(*noise)
        assert ctx.last_warning() == "" and (L.csky_last_error(ctx._h) or b"") == b""
        old = os.environ.get("GPU_MAX_HW_QUEUES")
        os.environ["GPU_MAX_HW_QUEUES"] = "2"
        try:
            ctx.set_frames_in_flight(4)
            assert "GPU_MAX_HW_QUEUES" in ctx.last_warning() and (L.csky_last_error(ctx._h) or b"") == b""
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
            ctx.set_frames_in_flight(2)
            assert ctx.last_warning() == ""
        finally:
            if old is None:
                del os.environ["GPU_MAX_HW_QUEUES"]
            else:
                os.environ["GPU_MAX_HW_QUEUES"] = old
    finally:
        ctx.close()


def test_multi_groups_render_the_sky_lut_on_every_device(pkg, noise, oracle, oracle_frames):
    """with frame groups the sky LUT used to go to the NEXT frame's group only; a host with a static sun that rendered it once marched
    the other groups with no LUT (CSKY_ERR_STATE) or a stale one.  One LUT call, then one frame per group: every frame is the oracle's."""
    import torch
    W, H = 512, 256
    sun = SUNS["deg45"]
    ref, _ = oracle_frames(W, H, "deg45")
    m = pkg.MultiContext([0, 0, 0, 0])
    try:
        m.set_noise(*noise); m.set_groups(2)
        m.render_sky_lut(norm(sun), 200, 100)                               # ONCE
        p = oracle.default_params(W, H, sun)
        out = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
        for k in range(4):                                                  # frames alternate between the two groups
            out.zero_(); torch.cuda.synchronize()                            # (torch's stream and the handle's own are not ordered against each other)
            m.render_clouds_device(p, W, H, out.data_ptr(), W * 8)
            m.sync(); torch.cuda.synchronize()
            ok, info = cloud_tight(out.cpu().numpy().view(np.float16), ref)
            assert ok, (k, info)
    finally:
        m.close()


def _lut_rows(ctx, torch, sun, r, n, w=200, h=100):
    """rank r of n: its rows of the LUT, compact, as [rows, w, 4] uint16"""
    rows = len(range(r, h, n))
    buf = torch.zeros(max(1, rows) * w * 8, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                       # (the fill runs on torch's stream, the render on `st`: nothing else orders them)
    st = torch.cuda.Stream()
    ctx.render_sky_lut_rows_device(sun, r, n, buf.data_ptr(), buf.numel(), w, h, st.cuda_stream)
    st.synchronize()
    return buf.cpu().numpy().view(np.uint16).reshape(-1, w, 4)[:rows]


@pytest.mark.parametrize("n", [2, 8, 3])
def test_sky_lut_rows_interleave_to_the_whole_lut(gpu_ctx, n):
    """csky_render_sky_lut_rows_device (sky_lut.gd:43-52 split over the ranks of a frame split): rows r::n of every rank, interleaved, are the LUT
    one context renders whole, to the byte (ragged for n = 3 / 8: 100 rows)."""
    import torch
    sun = norm(SUNS["demo"])
    whole = gpu_ctx.render_sky_lut(sun, 200, 100).view(np.uint16)
    got = np.zeros_like(whole)
    for r in range(n):
        got[r::n] = _lut_rows(gpu_ctx, torch, sun, r, n)
    assert np.array_equal(got, whole)
    gpu_ctx.render_sky_lut(sun, 200, 100)          # leave the shared context with a whole LUT


@pytest.mark.parametrize("sun_key,light", [("deg45", None), ("zenith", None), ("demo", None), ("deg45", (-0.3, -0.8, 0.5)), ("demo", (0.0, 1.0, 0.0))])
def test_frames_marched_on_a_rows_only_lut_are_byte_identical(gpu_ctx, oracle, sun_key, light):
    """A context whose LUT went to the caller as rows keeps none: its frame set-up renders the <= 12 texels it filters itself, for the light
    direction of each frame (clouds.glsl:163-167; frame_setup_taps_kernel) -- also when LIGHT_DIRECTION is not the sky LUT's sun, below the
    horizon (v clamps at row 0) or straight up (atan2(0, 0), top row): this rank's bands are byte-identical to the ones marched on the whole LUT."""
    import torch
    W, H = 512, 256
    sun = norm(SUNS[sun_key])
    params = oracle.default_params(W, H, light if light is not None else SUNS[sun_key])
    bands = (8, 3, 8, H // 8 // 8)
    out = [torch.zeros((bands[3] * 8, W, 4), dtype=torch.int16, device="cuda") for _ in range(2)]
    st = torch.cuda.Stream()
    other = norm((-0.2, 0.15, 0.9))                # both slots of the context's LUT ring hold ANOTHER sun's texels when the rows-only frame is marched
    rows = torch.zeros(13 * 200 * 8, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                       # the fills above run on torch's stream, the renders on `st`
    gpu_ctx.render_sky_lut_device(sun, 200, 100, st.cuda_stream)
    gpu_ctx.render_clouds_device(params, W, bands, out[0].data_ptr(), W * 8, st.cuda_stream)
    gpu_ctx.render_sky_lut_device(other, 200, 100, st.cuda_stream)
    gpu_ctx.render_sky_lut_device(other, 200, 100, st.cuda_stream)
    gpu_ctx.render_sky_lut_rows_device(sun, 3, 8, rows.data_ptr(), rows.numel(), 200, 100, st.cuda_stream)
    gpu_ctx.render_clouds_device(params, W, bands, out[1].data_ptr(), W * 8, st.cuda_stream)
    st.synchronize()
    assert bool((out[0] == out[1]).all().item()) and float(out[0].view(torch.float16)[..., 3].float().mean().item()) > 0.0
    gpu_ctx.render_sky_lut(sun, 200, 100)


def test_a_rows_only_lut_cannot_be_read_as_a_whole_one(gpu_ctx, pkg):
    import torch
    sun = norm(SUNS["deg45"])
    buf = torch.zeros(200 * 100 * 8, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.render_sky_lut_rows_device(sun, 1, 4, buf.data_ptr(), buf.numel(), 200, 100, None)
    with pytest.raises(pkg.CloudSkyError) as e:
        gpu_ctx.read_sky_lut()
    assert e.value.code == pkg._lib.ERR_STATE
    with pytest.raises(pkg.CloudSkyError) as e:
        gpu_ctx.copy_sky_lut_device(buf.data_ptr(), None)
    assert e.value.code == pkg._lib.ERR_STATE
    with pytest.raises(pkg.CloudSkyError) as e:
        gpu_ctx.render_sky_lut_rows_device(sun, 1, 4, buf.data_ptr(), 25 * 200 * 8 - 1, 200, 100, None)      # 25 rows of 1600 bytes needed
    assert e.value.code == pkg._lib.ERR_INVALID
    with pytest.raises(pkg.CloudSkyError) as e:
        gpu_ctx.render_sky_lut_rows_device(sun, 4, 4, buf.data_ptr(), buf.numel(), 200, 100, None)            # first_row < row_stride
    assert e.value.code == pkg._lib.ERR_INVALID
    whole = gpu_ctx.render_sky_lut(sun, 200, 100)
    assert gpu_ctx.read_sky_lut().shape == (100, 200, 4)
    gpu_ctx.render_sky_lut_rows_device(sun, 0, 1, buf.data_ptr(), buf.numel(), 200, 100, None)                # stride 1: every row
    gpu_ctx.sync()
    assert np.array_equal(buf.cpu().numpy().view(np.uint16).reshape(100, 200, 4), whole.view(np.uint16))
    gpu_ctx.render_sky_lut(sun, 200, 100)


def test_gpu_bc7_encoder_is_the_host_twin_to_the_byte(gpu_ctx, hostsim, noise):
    """bc7enc.hip (one block per lane) against the same per-block code compiled for the host (tests/hostsim): the fits use IEEE + - * / only,
    everything else is integer, so the 16 bytes of every block must agree -- weather map, shape slices, the whole detail volume, a ragged size."""
    import ctypes as C
    large, small, weather = noise
    def opaque(rgb):
        return np.concatenate([rgb, np.full(rgb.shape[:-1] + (1,), 255, np.uint8)], -1)
    rng = np.random.default_rng(2)
    ragged = np.clip(rng.normal(128, 30, size=(3, 13, 22, 4)), 0, 255).astype(np.uint8)
    for name, img in (("weather", opaque(weather)[None]), ("shape", large[30:38]), ("detail", opaque(small)), ("ragged", ragged)):
        img = np.ascontiguousarray(img)
        n, h, w = img.shape[:3]
        want = np.zeros((n, (h + 3) // 4, (w + 3) // 4, 16), np.uint8)
        hostsim.hostsim_bc7_encode(img.ctypes.data_as(C.c_void_p), w, h, n, want.ctypes.data_as(C.c_void_p))
        got = gpu_ctx.encode_bc7(img)
        assert got.shape == want.shape and (got == want).all(), (name, int((got != want).any(-1).sum()))


def test_frame_marched_on_bc7_round_tripped_textures_vs_oracle(pkg, gpu_ctx, noise, oracle, o_trans):
    """assets.vram_compressed_chains: the inputs as compress/mode=2 of the *.import files would leave them (with this library's encoder in the
    importer's place), bound with csky_set_noise_mips.  The HIP path must march exactly those texels -- the oracle is fed the same chains, tight
    gate -- and the frame must differ from the uncompressed one (that difference is what tools/bc7_sensitivity.py reports)."""
    large, small, weather = noise
    (lc, sc, wq), stats = pkg.assets.vram_compressed_chains(gpu_ctx, large, small, weather)
    assert 30.0 < stats["large_psnr_level0"] < 60.0 and 30.0 < stats["small_psnr_level0"] < 60.0 and stats["weather_psnr"] > 45.0, stats
    sun = (1, 1, 0)
    sk_o = oracle.sky_lut(norm(sun), o_trans, 200, 100)
    p = oracle.default_params(256, 128, sun)
    ref, st_o = oracle.clouds(oracle.OracleTextures.from_chains(lc, sc, wq), p, sk_o, nthreads=oracle.max_threads(), return_stats=True)
    ref_raw, _ = oracle.clouds(oracle.OracleTextures(large, small, weather), p, sk_o, nthreads=oracle.max_threads(), return_stats=True)
    ctx = pkg.Context(0)
    try:
        ctx.set_noise_mips(lc, sc, wq)                     # (cells that do not fit fp16 pairs, if any, are marched as exact fp32 cells)
        ctx.set_march(128, 6); ctx.render_transmittance(256, 64); ctx.render_sky_lut(norm(sun), 200, 100)
        img = ctx.render_clouds(p)
        ok, info = cloud_tight(img, ref)
        assert ok, info
        assert ctx.cloud_stats()["primary_samples"] == st_o["primary_samples"]
        assert float(np.abs(ref.astype(np.float32) - ref_raw.astype(np.float32)).max()) > 1e-2      # compression is visible at this precision
    finally:
        ctx.close()


def test_multi_handle_sky_lut_is_assembled_on_the_first_device(pkg, noise):
    """csky_multi_render_sky_lut: device i stores rows i::n into the LUT of the first device (4 contexts on device 0 here, which runs the same
    peer-store code); read through context 0 it is the whole LUT of a single context to the byte, also as a device copy, two suns in a row
    (both ring slots); the other contexts hold none."""
    import torch
    ref_ctx = pkg.Context(0)
    m = pkg.MultiContext([0, 0, 0, 0])
    try:
        m.set_noise(*noise)
        buf = torch.zeros(200 * 100 * 8, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for sun in (SUNS["deg45"], SUNS["demo"], SUNS["zenith"]):
            want = ref_ctx.render_sky_lut(norm(sun), 200, 100).view(np.uint16)
            m.render_sky_lut(norm(sun), 200, 100)
            c0 = m.ctx(0)
            assert np.array_equal(c0.read_sky_lut().view(np.uint16), want)
            c0.copy_sky_lut_device(buf.data_ptr(), None)
            c0.sync()
            assert np.array_equal(buf.cpu().numpy().view(np.uint16).reshape(100, 200, 4), want)
            with pytest.raises(pkg.CloudSkyError) as e:
                m.ctx(2).read_sky_lut()
            assert e.value.code == pkg._lib.ERR_STATE
    finally:
        m.close(); ref_ctx.close()


@pytest.mark.parametrize("world,groups", [(8, 1), (3, 1), (8, 2)])
def test_library_interleave_is_tilings_interleave(gpu_ctx, pkg, world, groups):
    """csky_interleave_bands_device (rank 0's last step at N > 1) against tiling.FrameGroups.split / assemble / assemble_lut in torch: frame and
    sky LUT byte-identical, ragged splits (3 ranks: 43 + 43 + 42 bands, 34 + 33 + 33 LUT rows) and a group whose gather starts with rank 0's dummy."""
    import torch
    T = pkg.tiling
    H, W, LH, LW = 1024, 256, 100, 200
    fg = T.FrameGroups(0, world, groups)
    bb, lb = fg.max_bands(H) * 8 * W * 8, fg.max_lut_rows(LH) * LW * 8
    g = torch.randint(0, 256, (fg.max_members, bb + lb), dtype=torch.uint8, device="cuda")
    for frame_no in range(groups):
        img, lut = fg.split(g, H, W, LH, LW)
        want_f, want_l = fg.assemble(frame_no, img, H), fg.assemble_lut(frame_no, lut, LH)
        out_f = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda"); out_l = torch.zeros((LH, LW, 4), dtype=torch.int16, device="cuda")
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        fg.assemble_device(frame_no, g, gpu_ctx, st.cuda_stream, H, W, out_f, LH, LW, out_l)
        st.synchronize()
        assert bool((out_f == want_f).all().item()) and bool((out_l == want_l).all().item()), (world, groups, frame_no)
