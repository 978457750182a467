"""Asset layer: BMP loader, Godot 3-D slicing, deterministic shape-noise generator, mip chains (CPU only)."""
import os

import numpy as np

SHAPE_SHA256 = "d52ddc23e25fff16c39e6fd672ab28c4d39ac4167f6d167f719ea37cd7704c0d"   # tests/golden/INPUTS.txt
SMALL_SHA256 = "9cfdc06c3ff2483aec9c6837fd83464509db609c1769a58430247f428561f6b6"
WEATHER_SHA256 = "2c15fb3c19a0e5fca66644fc08b9a9a2ac71f12ea284d5f03b618bca2385154c"


def test_bmp_loader_matches_pil(pkg):
    from PIL import Image
    for name in ("weather.bmp", "worlnoise.bmp"):
        p = os.path.join(pkg.assets.ASSET_DIR, name)
        assert (pkg.assets.load_bmp_rgb8(p) == np.array(Image.open(p).convert("RGB"))).all()


def test_bmp_loader_errors(pkg, tmp_path):
    import pytest
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.load_bmp_rgb8(str(tmp_path / "missing.bmp"))
    bad = tmp_path / "bad.bmp"
    bad.write_bytes(b"XX" + b"\0" * 100)
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.load_bmp_rgb8(str(bad))


def test_strip_to_volume_layout(pkg):
    # worlnoise.bmp.import:26-27: slices/horizontal=32, vertical=1 -> voxel(x,y,z) = strip[y][32 z + x]
    strip = pkg.assets.load_bmp_rgb8(os.path.join(pkg.assets.ASSET_DIR, "worlnoise.bmp"))
    vol = pkg.assets.strip_to_volume(strip, 32)
    rng = np.random.default_rng(0)
    for x, y, z in rng.integers(0, 32, (64, 3)):
        assert (vol[z, y, x] == strip[y, 32 * z + x]).all()


def test_inputs_pinned(pkg, noise):
    large, small, weather = noise
    assert pkg.assets.sha256(large) == SHAPE_SHA256      # integer hash + IEEE ops only: machine independent
    assert pkg.assets.sha256(small) == SMALL_SHA256
    assert pkg.assets.sha256(weather) == WEATHER_SHA256
    assert large.shape == (128, 128, 128, 4) and small.shape == (32, 32, 32, 3) and weather.shape == (512, 512, 3)


def test_shape_noise_is_tileable_and_calibrated(noise):
    large = noise[0].astype(np.float32)
    for ax in range(3):   # REPEAT sampler (cloud_sky.gd:302-304): the wrap seam must look like any interior step
        seam = np.abs(np.take(large, 0, ax) - np.take(large, 127, ax)).mean()
        inner = np.abs(np.take(large, 64, ax) - np.take(large, 63, ax)).mean()
        assert seam < 2.0 * inner + 1.0
    m = large.reshape(-1, 4).mean(0) / 255.0
    assert 0.6 < m[0] < 0.9 and all(0.3 < v < 0.7 for v in m[1:])


def test_generator_small_and_bad_args(pkg):
    import pytest
    a = pkg.assets.generate_shape_noise(7, 16)
    b = pkg.assets.generate_shape_noise(7, 16)
    c = pkg.assets.generate_shape_noise(8, 16)
    assert (a == b).all() and (a != c).any()
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.generate_shape_noise(1, 12)


def _spectral_centroid(ch):
    F = np.abs(np.fft.fftn(ch - ch.mean())) ** 2
    k = np.fft.fftfreq(ch.shape[0]) * ch.shape[0]
    K = np.sqrt(k[:, None, None] ** 2 + k[None, :, None] ** 2 + k[None, None, :] ** 2)
    return float((K * F).sum() / F.sum())


def test_generated_detail_noise_matches_the_shipped_asset_statistics(pkg, noise):
    """SURVEY §8f row 2 / README.md:30 TODO 3: the generated 32^3 Worley detail volume is compared with worlnoise.bmp ITSELF (not with
    another run of the generator): per-channel mean, spread, range, dominant spatial frequency and 3-D tileability."""
    asset = noise[1].astype(np.float64) / 255.0
    gen8 = pkg.assets.generate_detail_noise(1, 32)
    assert gen8.shape == (32, 32, 32, 3) and (gen8 == pkg.assets.generate_detail_noise(1, 32)).all()       # deterministic
    assert (gen8 != pkg.assets.generate_detail_noise(2, 32)).any()
    gen = gen8.astype(np.float64) / 255.0
    for c in range(3):
        a, g = asset[..., c], gen[..., c]
        assert abs(g.mean() - a.mean()) < 0.03, (c, g.mean(), a.mean())          # asset: 0.711 / 0.706 / 0.708
        assert abs(g.std() - a.std()) < 0.03, (c, g.std(), a.std())              # asset: 0.112 / 0.112 / 0.140
        assert g.max() > 0.97 and a.max() > 0.97 and g.min() < 0.45              # full bright range, dark cell borders
        assert abs(_spectral_centroid(g) - _spectral_centroid(a)) < 0.8, (c, _spectral_centroid(g), _spectral_centroid(a))   # asset: 2.4 / 4.6 / 7.0
        for ax in range(3):   # REPEAT sampler (cloud_sky.gd:302-304): the wrap seam looks like an interior step, as in the asset
            seam = np.abs(np.take(g, 0, ax) - np.take(g, 31, ax)).mean()
            inner = np.abs(np.diff(g, axis=ax)).mean()
            assert seam < 1.5 * inner + 0.01, (c, ax, seam, inner)
    import pytest
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.generate_detail_noise(1, 12)


def test_mips_match_oracle_and_numpy(pkg, oracle, noise):
    from oracle import numpy_restatement as NR
    large, small, _ = noise
    for vol, n, ch, lv in ((large, 128, 4, 8), (small, 32, 3, 6)):
        prod = pkg.assets.build_mips(vol, lv)
        assert (prod == oracle.build_mip_chain(vol, n, ch, lv)).all()     # integer work: bit exact
        chain = NR.mip_chain(vol)
        flat = np.concatenate([l.reshape(-1) for l in chain[:lv]])
        assert (prod == flat).all()
    assert prod[-3:].tolist() == NR.mip_chain(small)[5].reshape(-1).tolist()   # LOD 5 = 1x1x1 mean texel


def test_tga_loader_matches_pil(pkg, tmp_path):
    """TGA types 2 and 10, 24 and 32 bpp, both origins (the container of cloud_sky/perlworlnoise.tga)."""
    from PIL import Image
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (16, 64, 4), dtype=np.uint8)
    img[4:9, 10:40] = img[4, 10]                                    # runs, so RLE packets of both kinds appear
    for mode, rle in (("RGBA", False), ("RGBA", True), ("RGB", False), ("RGB", True)):
        p = str(tmp_path / ("t_%s_%d.tga" % (mode, rle)))
        im = Image.fromarray(img if mode == "RGBA" else img[..., :3], mode)
        im.save(p, compression="tga_rle" if rle else None)
        got = pkg.assets.load_tga_rgba8(p)
        ref = np.array(Image.open(p).convert("RGBA"))
        assert (got == ref).all(), (mode, rle)
    # top-left origin flag
    p = str(tmp_path / "top.tga")
    Image.fromarray(img, "RGBA").save(p, orientation=1)
    assert (pkg.assets.load_tga_rgba8(p) == np.array(Image.open(p).convert("RGBA"))).all()
    # a 3-D strip: 8 slices of 8x8 -> volume layout of perlworlnoise.tga.import:26-27
    strip = rng.integers(0, 256, (8, 64, 4), dtype=np.uint8)
    p = str(tmp_path / "strip.tga")
    Image.fromarray(strip, "RGBA").save(p)
    vol = pkg.assets.load_shape_noise_tga(p, 8)
    assert (vol[3, 5, 2] == strip[5, 8 * 3 + 2]).all()
    import pytest
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.load_tga_rgba8(str(tmp_path / "missing.tga"))
    bad = tmp_path / "bad.tga"
    bad.write_bytes(bytes(18))
    with pytest.raises(pkg.CloudSkyError):
        pkg.assets.load_tga_rgba8(str(bad))


def test_tuned_shape_generator_defaults_are_the_benchmark_volume_and_knobs_move_only_their_channels(pkg):
    """csky_generate_shape_noise_tuned (README.md:30 TODO 3: a generator that can be tweaked): with csky_shape_noise_default_params it IS
    csky_generate_shape_noise; the R-channel knobs leave G / B / A alone; out-of-range settings are refused, not rendered."""
    import pytest
    base = pkg.assets.generate_shape_noise(3, 64)
    assert (pkg.assets.generate_shape_noise(3, 64, **{}) == base).all()
    p = pkg._lib.shape_noise_params()
    assert (p.perlin_freq, p.perlin_octaves, p.worley_freq) == (4, 5, 4) and abs(p.dilate - 0.55) < 1e-7 and abs(p.contrast - 1.75) < 1e-7
    finer = pkg.assets.generate_shape_noise(3, 64, perlin_freq=8, perlin_octaves=4, dilate=0.8)
    assert (finer[..., 1:] == base[..., 1:]).all() and (finer[..., 0] != base[..., 0]).mean() > 0.5
    w8 = pkg.assets.generate_shape_noise(3, 128, worley_freq=8)
    assert (w8[..., 1] != pkg.assets.generate_shape_noise(3, 128)[..., 1]).mean() > 0.5
    for bad in (dict(perlin_freq=0), dict(perlin_freq=32, perlin_octaves=4), dict(worley_freq=16), dict(contrast=0.0), dict(dilate=1.5), dict(perlin_gain=float("nan"))):
        with pytest.raises(pkg.CloudSkyError):
            pkg.assets.generate_shape_noise(3, 64, **bad)
    with pytest.raises(TypeError):
        pkg.assets.generate_shape_noise(3, 64, no_such_knob=1)
    # the n-independent bounds hold for the small preview volumes too (n < 64 used to skip every check: perlin_freq = 0 reached `i % 0`,
    # SIGFPE across the ABI), shifts and products cannot wrap (perlin_freq = 1 << 30 with 3 octaves, worley_freq = 1 << 28), and an infinity is not "> 0"
    inf = float("inf")
    for n, bad in ((32, dict(perlin_freq=0)), (32, dict(worley_freq=0)), (16, dict(perlin_octaves=0)), (8, dict(perlin_octaves=1 << 20)), (32, dict(perlin_gain=inf)),
                   (64, dict(perlin_freq=1 << 30, perlin_octaves=3)), (64, dict(worley_freq=1 << 28)), (128, dict(contrast=inf)), (64, dict(centre=inf)),
                   (64, dict(offset=-inf)), (32, dict(offset=float("nan"))), (64, dict(perlin_freq=-4))):
        with pytest.raises(pkg.CloudSkyError):
            pkg.assets.generate_shape_noise(3, n, **bad)
    small = pkg.assets.generate_shape_noise(3, 32)                      # the default knobs at a preview size still render (the texels-per-cell rule starts at 64)
    assert small.shape == (32, 32, 32, 4) and small[..., 0].std() > 5
