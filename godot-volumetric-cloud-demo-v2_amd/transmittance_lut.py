"""Host mirror of cloud_sky/transmittance_lut.gd (a Texture2DRD that renders itself once at load)."""
import numpy as np


class TransmittanceLut:
    """transmittance_lut.gd:1-78.  `texture_size` = Vector2i(256, 64) (:6); the LUT is rendered once when the
    resource is created (:15-18 -> :51-77) and then only sampled."""

    def __init__(self, ctx, texture_size=(256, 64)):
        self.ctx = ctx
        self.texture_size = tuple(int(v) for v in texture_size)
        self._image = None
        self._initialize_compute_code()

    def _initialize_compute_code(self):  # transmittance_lut.gd:51-77: create pipeline + the one dispatch (32, 8, 1)
        w, h = self.texture_size
        self._image = self.ctx.render_transmittance(w, h)

    @property
    def image(self):
        """RGBA16F texels as float16 [h, w, 4] (what `texture_rd` holds)."""
        return self._image

    def push_constant(self):  # transmittance_lut.gd:66-70
        return np.array([self.texture_size[0], self.texture_size[1], 0.0, 0.0], np.float32)
