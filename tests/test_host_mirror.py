"""Host mirror of the GDScript drivers (cloud_sky.gd / sky_lut.gd / transmittance_lut.gd): property defaults,
push-constant packing, call order, tile walk, ring rotation.  The C-ABI context is replaced by a recording fake
whose images come from the oracle (tests may use the oracle as a stand-in; the product never does)."""
import numpy as np
import pytest

from conftest import norm


class FakeContext:
    device_id = 0

    def __init__(self, oracle, otex):
        self.O, self.tex, self.calls = oracle, otex, []
        self.trans = self.sky = None
        self.primary, self.light = 128, 6

    def set_noise(self, *a):
        self.calls.append(("set_noise",))

    def render_transmittance(self, w, h):
        self.calls.append(("transmittance", w, h))
        self.trans = self.O.transmittance_lut(w, h)
        return self.trans

    def render_sky_lut_device(self, sun, w, h, stream=None):
        self.calls.append(("sky", tuple(np.round(np.asarray(sun, np.float64), 6)), w, h))
        self.sky = self.O.sky_lut(np.asarray(sun, np.float32), self.trans, w, h)

    def render_sky_lut(self, sun, w, h, readback=True):
        self.render_sky_lut_device(sun, w, h)
        return self.sky

    def read_sky_lut(self):
        return self.sky

    def render_clouds(self, params, tile_w=None, tile_h=None):
        p = np.asarray(params, np.float32)
        self.calls.append(("clouds", p.copy(), tile_w, tile_h))
        return self.O.clouds(self.tex, p, self.sky, rect=(0, 0, tile_w, tile_h), primary_steps=self.primary, light_steps=self.light)

    def close(self):
        self.calls.append(("close",))


@pytest.fixture()
def fake(oracle, otex):
    return FakeContext(oracle, otex)


def make_sky(pkg, fake, noise, **kw):
    sky = pkg.CloudSky.from_default_resource(ctx=fake, noise=noise, clock=lambda: 0.0, **kw)
    sky.sun = pkg.cloud_sky.DirectionalLight(direction=(1, 1, 0))
    return sky


def test_defaults_match_gdscript(pkg, fake, noise):
    sky = pkg.CloudSky(ctx=fake, noise=noise, clock=lambda: 0.0)
    # cloud_sky.gd:10-50 script defaults
    assert (sky.wind_direction, sky.wind_speed, sky.density, sky.cloud_coverage, sky.time_offset) == (0.0, 1.0, 0.05, 0.25, 0.0)
    assert sky.texture_size == (768, 768) and sky.sun_disk_scale == 1.0
    assert sky.frame_data.LIGHT_DIRECTION.tolist() == [0.0, -1.0, 0.0]       # cloud_sky.gd:72
    assert sky.sky_lut.texture_size == (200, 100) and sky.transmittance_tex.texture_size == (256, 64)
    res = pkg.CloudSky.from_default_resource(ctx=fake, noise=noise, clock=lambda: 0.0)
    assert (res.density, res.cloud_coverage, res.sun_disk_scale) == (0.05, 0.2, 2.0)      # clouds_sky.tres:13-17
    np.testing.assert_allclose(res.ground_color, [0.270588, 0.188235, 0.027451, 1.0])


def test_update_performance_and_reference_tile_split(pkg, fake, noise):
    sky = pkg.CloudSky(ctx=fake, noise=noise, clock=lambda: 0.0, frames_to_update=64)
    assert sky.update_region_size == [96, 96] and sky.num_workgroups == [12, 12]           # cloud_sky.gd:83-84
    sky.texture_size = 100                                                                 # not a multiple of 8
    assert sky.texture_size == (96, 96) and sky.update_region_size == [12, 12] and sky.num_workgroups == [2, 2]   # :110-115
    with pytest.raises(ValueError):
        sky.frames_to_update = 9


def test_push_constant_block_default_config(pkg, fake, noise, oracle):
    sky = make_sky(pkg, fake, noise, texture_size=(64, 32))
    sky.update_sky()
    pc = sky._fill_push_constant()
    ref = oracle.default_params(64, 32, (1, 1, 0))         # SURVEY A.2 default-config block
    np.testing.assert_allclose(pc, ref, rtol=0, atol=1e-7)
    assert pc.dtype == np.float32 and pc.nbytes == 112
    assert sky.sky_lut.push_constant().nbytes == 32 and sky.transmittance_tex.push_constant().nbytes == 16


def test_call_order_matches_reference(pkg, fake, noise):
    sky = make_sky(pkg, fake, noise, texture_size=(32, 16))
    sky.update_sky()
    kinds = [c[0] for c in fake.calls]
    # transmittance once at load (transmittance_lut.gd:15-18); noise set; first update renders the sky LUT three
    # times (sky_lut.gd:49-52) before any cloud dispatch; initialize_sky renders 2 passes (cloud_sky.gd:124-127)
    assert kinds[0] == "transmittance" and kinds.count("transmittance") == 1
    assert kinds.index("set_noise") < kinds.index("sky") < kinds.index("clouds")
    first_clouds = kinds.index("clouds")
    assert kinds[:first_clouds].count("sky") == 3
    assert kinds.count("clouds") == 3                      # 2 (initialize_sky) + 1 (this frame)
    sun = [c for c in fake.calls if c[0] == "sky"][0][1]
    np.testing.assert_allclose(sun, norm((1, 1, 0)), atol=1e-6)


def test_wind_integration(pkg, fake, noise):
    t = [0.0]
    sky = pkg.CloudSky.from_default_resource(ctx=fake, noise=noise, clock=lambda: t[0], texture_size=(16, 8))
    sky.wind_direction, sky.wind_speed, sky.time_offset = np.pi / 2, 3.0, 2.0
    sky.sun = pkg.cloud_sky.DirectionalLight(direction=(0, 1, 0))
    sky.update_sky()
    t[0] = 10.0
    sky.update_sky()                                      # frame >= frames_to_update -> _update_per_frame_data
    fd = sky.frame_data
    # cloud_sky.gd:176-185
    np.testing.assert_allclose(fd._detailed_pos, [0.0, 10.0], atol=1e-9)
    np.testing.assert_allclose(fd._cloud_pos, [0.0, 30.0], atol=1e-9)
    # every _update_per_frame_data adds 0.005*time_offset even at delta == 0; there were 4 so far
    # (initialize_sky, its 2nd nested pass, the outer first frame, this frame: cloud_sky.gd:124-142)
    assert abs(fd._weather_pos[1] - ((10.0 * 0.001) * 3.0 + 4 * 0.005 * 2.0 * 3.0)) < 1e-9 and abs(fd._weather_pos[0]) < 1e-9
    pc = sky._fill_push_constant()
    assert pc[23] == np.float32(10.0) and pc[27] == np.float32(2.0)


def test_temporal_split_assembles_the_full_frame(pkg, fake, noise, oracle, otex):
    """frames_to_update = 4: four tile dispatches walked by update_position (cloud_sky.gd:156-161) assemble the
    same texture a single full-hemisphere call renders."""
    sky = make_sky(pkg, fake, noise, texture_size=(32, 16), frames_to_update=4)
    for _ in range(4):
        tex = sky.update_sky()
    walked = [tuple(c[1][2:4]) for c in fake.calls if c[0] == "clouds"][-4:]
    assert walked == [(0.0, 0.0), (16.0, 0.0), (0.0, 8.0), (16.0, 8.0)]
    full = oracle.clouds(otex, oracle.default_params(32, 16, (1, 1, 0)), fake.sky)
    assert (tex.view(np.uint16) == full.view(np.uint16)).all()
    assert sky.blend_amount == 0.75 and sky.frame == 4
    # ring rotation (cloud_sky.gd:137-142): initialize_sky's second pass and the first outer frame rotated twice
    assert (sky.texture_to_update, sky.texture_to_blend_from, sky.texture_to_blend_to) == (2, 0, 1)
    sky.update_sky()                                      # frame >= frames_to_update -> third rotation
    assert (sky.texture_to_update, sky.texture_to_blend_from, sky.texture_to_blend_to) == (0, 1, 2)
    assert sky.frame == 1 and sky.update_position == [16, 0]


def test_full_hemisphere_single_call(pkg, fake, noise, oracle, otex):
    sky = make_sky(pkg, fake, noise, texture_size=(32, 16), frames_to_update=1)
    tex = sky.update_sky()
    full = oracle.clouds(otex, oracle.default_params(32, 16, (1, 1, 0)), fake.sky)
    assert (tex.view(np.uint16) == full.view(np.uint16)).all()


def test_light_data(pkg):
    fd = pkg.FrameData()
    # cloud-demo.tscn:21 DirectionalLight3D transform; third basis column = direction towards the sun
    basis = np.array([[-0.0492487, -0.00526289, -0.998773], [-0.993118, -0.106134, 0.0495291], [-0.106264, 0.994338, 2.69869e-07]])
    light = pkg.cloud_sky.DirectionalLight(basis=basis, light_energy=1.5, light_color=(0.5, 0.2, 1.0, 1.0))
    fd.update_light_data(light)
    np.testing.assert_allclose(fd.LIGHT_DIRECTION, [-0.998773, 0.0495291, 2.69869e-07], atol=2e-6)
    np.testing.assert_allclose(fd.LIGHT_COLOR[:3], [0.21404114, 0.03310477, 1.0], rtol=1e-5)     # Color.srgb_to_linear
    assert fd.LIGHT_ENERGY == 1.5
