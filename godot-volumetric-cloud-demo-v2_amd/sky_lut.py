"""Host mirror of cloud_sky/sky_lut.gd (sky-view LUT with a 3-texture ring)."""
import numpy as np


class SkyLut:
    """sky_lut.gd:1-148.  Renders the 200x100 sky-view LUT for a sun direction and keeps the reference's three-copy ring
    (`texture_rd[3]`, `current_texture`, `back_texture[2]`, :16-18,143-146): every render fills ring slot `current_texture` and
    advances it, so `back_texture` = the two OLDER copies, which clouds.gdshader cross-fades as sky_blend_from/to
    (cloud_sky.gd:147-148).  The clouds kernel always binds the copy rendered LAST (cloud_sky.gd:242), which is the library
    context's internal LUT.  Ring copies are numpy float16 arrays (host form) or torch float16 CUDA tensors filled by an
    asynchronous device copy (`device_buffers=True`: no host hop, no synchronisation in the frame loop)."""

    def __init__(self, ctx, transmittance, texture_size=(200, 100), device_buffers=False):
        self.ctx = ctx
        self.transmittance_tex = transmittance          # sky_lut.gd:22
        self.texture_size = tuple(int(v) for v in texture_size)   # sky_lut.gd:4
        self.light_direction = np.array([0.0, -1.0, 0.0], np.float32)  # sky_lut.gd:5
        self.needs_update = True
        self.initialized = transmittance is not None
        self.needs_full_update = True
        self.current_texture = 0
        self.device_buffers = bool(device_buffers)
        self.texture = [None, None, None]               # texture_rd[0..2]
        self.renders = 0

    def request_update(self):  # sky_lut.gd:39-40
        self.needs_update = True

    def update_lut(self, sun_direction, stream=None):  # sky_lut.gd:43-52
        self.light_direction = np.asarray(sun_direction, np.float32)
        if not self.initialized:
            print("Attempting to update uninitialized sky lut")
            return
        self.render_lut(stream)
        if self.needs_full_update:       # first use fills all three ring slots (sky_lut.gd:49-52)
            self.render_lut(stream)
            self.render_lut(stream)
            self.needs_full_update = False

    def push_constant(self):  # sky_lut.gd:123-132
        d = self.light_direction
        return np.array([self.texture_size[0], self.texture_size[1], 0.0, 0.0, d[0], d[1], d[2], 0.0], np.float32)

    def render_lut(self, stream=None):  # sky_lut.gd:122-148 (dispatch 25 x 13 groups)
        w, h = self.texture_size
        c = self.current_texture
        if self.device_buffers:
            import torch
            self.ctx.render_sky_lut_device(self.light_direction, w, h, stream)
            if self.texture[c] is None:
                self.texture[c] = torch.empty((h, w, 4), dtype=torch.float16, device=torch.device("cuda", self.ctx.device_id))
            self.ctx.copy_sky_lut_device(self.texture[c].data_ptr(), stream)
        else:
            self.texture[c] = self.ctx.render_sky_lut(self.light_direction, w, h)
        self.renders += 1
        self.current_texture = (c + 1) % 3
        self.needs_update = False

    @property
    def image(self):
        """The LUT rendered last, float16 [h, w, 4] (host copy)."""
        return self.ctx.read_sky_lut()

    @property
    def back_texture(self):  # sky_lut.gd:9,145-146
        c = self.current_texture
        return [self.texture[c], self.texture[(c + 1) % 3]]
