"""Host mirror of cloud_sky/cloud_sky.gd (the `Sky` resource that drives clouds.glsl).

Same property names, defaults, call order and push-constant packing as the GDScript, over the C ABI of
libcloudsky.so.  Differences, all deliberate (BASELINE.json north_star):
  * `frames_to_update = 1` is allowed and is the default: the full hemisphere is rendered every call instead of
    being split over 64 frames (the reference's 4/16/64/256 temporal split is still available);
  * the texture may be rectangular (`texture_size = (W, H)`; the shader's `texture_size` is a vec2, clouds.glsl:19);
  * `clock` is injectable (the reference reads `Time.get_ticks_msec()`, cloud_sky.gd:176) so that benchmarks and
    tests can freeze the wind;
  * textures are numpy float16 arrays (host form) or torch int16 CUDA tensors (device form, `device_buffers=True`).
"""
import math
import time as _time

import numpy as np

from . import assets as _assets
from . import tiling as _tiling
from ._lib import Context
from .sky_lut import SkyLut
from .transmittance_lut import TransmittanceLut


def srgb_to_linear(c):
    """Color.srgb_to_linear() (cloud_sky.gd:79)."""
    c = np.asarray(c, np.float64)
    return np.where(c < 0.04045, c * (1.0 / 12.92), ((c + 0.055) * (1.0 / 1.055)) ** 2.4).astype(np.float32)


class FrameData:
    """cloud_sky.gd:56-79: everything the compute shader reads, frozen for a whole update pass."""

    def __init__(self):
        self.wind_direction = np.array([1.0, 0.0])
        self.wind_speed = 1.0
        self.density = 0.05
        self.cloud_coverage = 0.25
        self.time_offset = 0.0
        self.ground_color = np.array([1.0, 1.0, 1.0, 1.0])
        self._time = 0.0
        self._cloud_pos = np.zeros(2)
        self._detailed_pos = np.zeros(2)
        self._weather_pos = np.zeros(2)
        self.LIGHT_DIRECTION = np.array([0.0, -1.0, 0.0])   # cloud_sky.gd:72 (no light attached: sun below horizon)
        self.LIGHT_ENERGY = 1.0
        self.LIGHT_COLOR = np.array([1.0, 1.0, 1.0, 1.0])

    def update_light_data(self, light):  # cloud_sky.gd:76-79
        b = np.asarray(light.basis, np.float64)  # 3x3, columns = basis vectors
        d = b @ np.array([0.0, 0.0, 1.0])
        self.LIGHT_DIRECTION = d / np.linalg.norm(d)
        self.LIGHT_ENERGY = float(light.light_energy)
        c = np.asarray(light.light_color, np.float64)
        self.LIGHT_COLOR = np.concatenate([srgb_to_linear(c[:3]), c[3:4] if c.size > 3 else [1.0]])


class DirectionalLight:
    """Stand-in for the DirectionalLight3D that sun.gd registers (sun.gd:11-13): only what update_light_data reads."""

    def __init__(self, direction=None, basis=None, light_energy=1.0, light_color=(1.0, 1.0, 1.0, 1.0)):
        if basis is None:
            d = np.asarray(direction, np.float64)
            d = d / np.linalg.norm(d)
            basis = np.zeros((3, 3))
            basis[:, 2] = d                      # basis * (0,0,1) = third column = direction towards the sun
        self.basis = np.asarray(basis, np.float64)
        self.light_energy = light_energy
        self.light_color = light_color


class CloudSky:
    """cloud_sky.gd.  Construct, optionally set `sun`, then call `update_sky()` once per frame."""

    FRAMES_TO_UPDATE_CHOICES = (1, 4, 16, 64, 256)   # cloud_sky.gd:36 plus 1 = full hemisphere per call

    def __init__(self, device_id=0, texture_size=768, frames_to_update=1, noise=None, clock=None, device_buffers=False,
                 rank=0, world_size=1, dist=None, ctx=None, async_host=False):
        # exported properties, cloud_sky.gd:5-50 with the defaults of the script (clouds_sky.tres overrides some)
        self.wind_direction = 0.0
        self.wind_speed = 1.0
        self.density = 0.05
        self.cloud_coverage = 0.25
        self.time_offset = 0.0
        self.sun_disk_scale = 1.0
        self.ground_color = np.array([1.0, 1.0, 1.0, 1.0])
        self._frames_to_update = int(frames_to_update)
        self._texture_size = self._as_size(texture_size)
        self.sun = None
        self.frame_data = FrameData()
        self.update_position = [0, 0]
        self.update_region_size = [96, 96]
        self.num_workgroups = [12, 12]
        self.textures = [None, None, None]
        self.texture_to_update, self.texture_to_blend_from, self.texture_to_blend_to = 0, 1, 2
        self.frame = 0
        self.blend_amount = 0.0
        self.can_run = False
        self.needs_full_sky_init = True
        if clock is None:           # Time.get_ticks_msec() counts from engine start (cloud_sky.gd:176): seconds since construction, so the
            t0 = _time.monotonic()  # first delta is ~0 like the reference's and wind offsets stay in fp32's accurate range
            clock = lambda: _time.monotonic() - t0  # noqa: E731
        self.clock = clock
        self.device_buffers = bool(device_buffers)
        # async_host: the host-buffer path through csky_submit_clouds / csky_collect (what the GDExtension's submit_clouds() / collect() wrap,
        # INTEGRATION.md): _render_process() collects the tile submitted by the PREVIOUS call into its texture and submits this call's tile, so
        # march + copy of one tile overlap the host's work on the next; the textures trail by one call, flush() collects the last tile.  The
        # reference's loop tolerates that by construction: it draws with the textures finished in earlier passes (cloud_sky.gd:137-148).
        self.async_host = bool(async_host) and not self.device_buffers
        self._pending = None
        self.rank, self.world_size, self.dist = int(rank), int(world_size), dist
        self.last_frame = None
        self._side_stream = None
        # render-thread side (cloud_sky.gd:218-232): the C-ABI context owns every device resource
        self.ctx = ctx if ctx is not None else Context(device_id)
        self.transmittance_tex = TransmittanceLut(self.ctx)                    # cloud_sky.gd:92
        self.sky_lut = SkyLut(self.ctx, self.transmittance_tex, device_buffers=self.device_buffers)   # cloud_sky.gd:91
        large, small, weather = noise if noise is not None else _assets.load_default_noise()
        self.ctx.set_noise(large, small, weather)                              # _create_noise_uniform_set, :298-341
        self.update_performance()

    # ---- clouds_sky.tres:11-18 ----------------------------------------------------------------------------
    @classmethod
    def from_default_resource(cls, **kw):
        """The demo's `clouds_sky.tres` values (the benchmark configuration)."""
        sky = cls(**kw)
        sky.wind_direction, sky.wind_speed = 0.0, 1.0
        sky.density, sky.cloud_coverage, sky.time_offset = 0.05, 0.2, 0.0
        sky.sun_disk_scale = 2.0
        sky.ground_color = np.array([0.270588, 0.188235, 0.027451, 1.0])
        return sky

    @staticmethod
    def _as_size(v):
        if np.isscalar(v):
            return [int(v), int(v)]
        return [int(v[0]), int(v[1])]

    @property
    def frames_to_update(self):
        return self._frames_to_update

    @frames_to_update.setter
    def frames_to_update(self, value):  # cloud_sky.gd:37-42
        self._frames_to_update = int(value)
        self.cleanup()
        self.update_performance()
        self.request_full_sky_init()

    @property
    def texture_size(self):
        return tuple(self._texture_size)

    @texture_size.setter
    def texture_size(self, value):  # cloud_sky.gd:45-50
        self._texture_size = self._as_size(value)
        self.cleanup()
        self.update_performance()
        self.request_full_sky_init()

    def update_performance(self):  # cloud_sky.gd:109-118
        if self._frames_to_update not in self.FRAMES_TO_UPDATE_CHOICES:
            raise ValueError("frames_to_update must be one of %s" % (self.FRAMES_TO_UPDATE_CHOICES,))
        frames_sqrt = int(math.isqrt(self._frames_to_update))
        for k in range(2):
            self.update_region_size[k] = self._texture_size[k] // frames_sqrt
            if self._texture_size[k] % frames_sqrt != 0:
                self._texture_size[k] = self.update_region_size[k] * frames_sqrt
                print("texture_size is not a multiple of sqrt(frames_to_update), changing to: ", self._texture_size[k])
            self.num_workgroups[k] = (self.update_region_size[k] + 7) // 8
        self._initialize_compute_code()

    def request_full_sky_init(self):  # cloud_sky.gd:120-121
        self.needs_full_sky_init = True

    def initialize_sky(self):  # cloud_sky.gd:124-127
        self._update_per_frame_data()
        for _ in range(self._frames_to_update * 2):
            self.update_sky()

    def update_sky(self):  # cloud_sky.gd:129-163
        if not self.can_run:
            return None
        if self.needs_full_sky_init:
            self.needs_full_sky_init = False
            self.initialize_sky()
        if self.frame >= self._frames_to_update:
            self.texture_to_update = (self.texture_to_update + 1) % 3
            self.texture_to_blend_from = (self.texture_to_blend_from + 1) % 3
            self.texture_to_blend_to = (self.texture_to_blend_to + 1) % 3
            self._update_per_frame_data()  # only once per pass, otherwise tiles get out of sync (cloud_sky.gd:142)
            self.frame = 0
        self.blend_amount = float(self.frame) / float(self._frames_to_update)
        out = self._render_process(self.texture_to_update)
        self.update_position[0] += self.update_region_size[0]
        if self.update_position[0] >= self._texture_size[0]:
            self.update_position[0] = 0
            self.update_position[1] += self.update_region_size[1]
        if self.update_position[1] >= self._texture_size[1]:
            self.update_position = [0, 0]
        self.frame += 1
        return out

    def _update_per_frame_data(self):  # cloud_sky.gd:165-187
        fd = self.frame_data
        if self.sun is not None:
            fd.update_light_data(self.sun)
        fd.wind_direction = np.array([math.cos(self.wind_direction), math.sin(self.wind_direction)])  # Vector2.from_angle
        fd.wind_speed = self.wind_speed
        fd.density = self.density
        fd.cloud_coverage = self.cloud_coverage
        fd.time_offset = self.time_offset
        fd.ground_color = np.asarray(self.ground_color, np.float64)
        t = float(self.clock())
        delta = t - fd._time
        delta2 = delta * 0.001 + 0.005 * fd.time_offset
        wdn = fd.wind_direction / np.linalg.norm(fd.wind_direction)
        fd._time = t
        fd._detailed_pos = fd._detailed_pos + delta * wdn
        fd._cloud_pos = fd._cloud_pos + delta * wdn * fd.wind_speed
        fd._weather_pos = fd._weather_pos + delta2 * wdn * fd.wind_speed
        stream, done = self._march_stream()
        self.sky_lut.update_lut(fd.LIGHT_DIRECTION, stream)
        done()

    def flush(self):
        """async_host: wait for the tile in flight (if any) and write it into its texture."""
        if self._pending is not None:
            ticket, texidx, (x0, y0, rw, rh) = self._pending
            self._pending = None
            tile = self.ctx.collect(ticket, copy=False)
            if self.textures[texidx] is not None:
                self.textures[texidx][y0:y0 + rh, x0:x0 + rw] = tile

    def cleanup(self):  # cloud_sky.gd:197-212
        if getattr(self, "_pending", None) is not None:                     # a tile of the old geometry is in flight: drop it
            try:
                self.ctx.collect(self._pending[0], copy=False)
            except Exception:
                pass
            self._pending = None
        self.can_run = False
        self.frame = 0
        self.texture_to_update, self.texture_to_blend_from, self.texture_to_blend_to = 0, 1, 2
        self.update_position = [0, 0]
        self.textures = [None, None, None]

    def close(self):  # NOTIFICATION_PREDELETE, cloud_sky.gd:193-195
        self.cleanup()
        self.ctx.close()

    def sky_panorama(self, out_w=2048, out_h=1024):
        """What clouds_material.tres + clouds.gdshader draw: the two blend textures cross-faded by blend_amount over the
        atmosphere with the sun disk, evaluated on an equirectangular panorama (csky_composite_sky).  float16 [h, w, 4]."""
        def host(t):
            return t.cpu().numpy() if hasattr(t, "cpu") else t
        self.flush()
        bf, bt = host(self.textures[self.texture_to_blend_from]), host(self.textures[self.texture_to_blend_to])
        sf, st = (host(t) for t in self.sky_lut.back_texture)   # sky_blend_from/to_texture = the two OLDER ring copies (cloud_sky.gd:147-148)
        return self.ctx.composite_sky(bf, bt, sf, st, self.frame_data.LIGHT_DIRECTION, self.blend_amount, self.sun_disk_scale, out_w, out_h)

    def sky_view(self, basis, fov_y_degrees=75.0, out_w=1152, out_h=648):
        """The same through a perspective camera: one EYEDIR per SCREEN pixel, the way the engine evaluates clouds.gdshader
        (csky_composite_view).  basis: 3x3, columns = the camera's right / up / back axes (Camera3D.global_transform.basis)."""
        def host(t):
            return t.cpu().numpy() if hasattr(t, "cpu") else t
        self.flush()
        bf, bt = host(self.textures[self.texture_to_blend_from]), host(self.textures[self.texture_to_blend_to])
        sf, st = (host(t) for t in self.sky_lut.back_texture)
        return self.ctx.composite_view(bf, bt, sf, st, self.frame_data.LIGHT_DIRECTION, basis, fov_y_degrees, self.blend_amount, self.sun_disk_scale, out_w, out_h)

    # ---- render thread ------------------------------------------------------------------------------------
    def _march_stream(self):
        """(hip stream handle, done()) for one batch of library calls in device-buffer mode.  The work is enqueued on torch's CURRENT
        stream so that torch work before it (tensor creation / fills) and after it (copies, .cpu(), collectives) is ordered around it
        without a device-wide synchronise.  torch's default stream is the HIP null stream, whose handle is 0 = "use the context's own
        non-blocking stream" to the library, which the null stream does NOT order against: in that case the batch runs
        on a private side stream that waits for the default stream first, and done() makes the default stream wait for it."""
        if not self.device_buffers:
            return None, (lambda: None)
        import torch
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != 0:
            return cur.cuda_stream, (lambda: None)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=torch.device("cuda", self.ctx.device_id))
        side = self._side_stream
        side.wait_stream(cur)
        return side.cuda_stream, (lambda: cur.wait_stream(side))

    def _fill_push_constant(self):  # cloud_sky.gd:251-289, same order incl. padding
        fd = self.frame_data
        pc = [self._texture_size[0], self._texture_size[1], self.update_position[0], self.update_position[1],
              fd._cloud_pos[0], fd._cloud_pos[1], fd._detailed_pos[0], fd._detailed_pos[1],
              fd._weather_pos[0], fd._weather_pos[1], 0.0, 0.0,
              fd.ground_color[0], fd.ground_color[1], fd.ground_color[2], fd.ground_color[3],
              fd.LIGHT_DIRECTION[0], fd.LIGHT_DIRECTION[1], fd.LIGHT_DIRECTION[2], fd.LIGHT_ENERGY,
              fd.LIGHT_COLOR[0], fd.LIGHT_COLOR[1], fd.LIGHT_COLOR[2], fd._time,
              0.0, fd.density, fd.cloud_coverage, fd.time_offset]
        return np.asarray(pc, np.float32)

    def _initialize_compute_code(self):  # cloud_sky.gd:355-408: three RGBA16F textures cleared to (1,0,0,0),(0,1,0,0),(0,0,1,0)
        w, h = self._texture_size
        if self.device_buffers:
            import torch
            dev = torch.device("cuda", self.ctx.device_id)
            self.textures = []
            for i in range(3):
                t = torch.zeros((h, w, 4), dtype=torch.float16, device=dev)
                t[..., i] = 1.0
                self.textures.append(t)
        else:
            self.textures = []
            for i in range(3):
                t = np.zeros((h, w, 4), np.float16)
                t[..., i] = 1.0
                self.textures.append(t)
        self.can_run = True

    def _render_process(self, p_texture_to_update):  # cloud_sky.gd:234-248
        pc = self._fill_push_constant()
        tex = self.textures[p_texture_to_update]
        W, H = self._texture_size
        rw, rh = self.update_region_size
        x0, y0 = self.update_position
        if self.async_host:
            self.flush()                                                    # the previous call's tile -> its texture (rd.texture_update there)
            self._pending = (self.ctx.submit_clouds(pc, rw, rh), p_texture_to_update, (x0, y0, rw, rh))
            self.last_frame = tex
            return tex
        if not self.device_buffers:
            tile = self.ctx.render_clouds(pc, rw, rh)                       # dispatch(num_workgroups, num_workgroups, 1)
            tex[y0:y0 + rh, x0:x0 + rw] = tile
            self.last_frame = tex
            return tex
        import torch
        if self._frames_to_update == 1:
            # full hemisphere per call, sharded over the ranks of one node
            def render_bands(bands, out):
                stream, done = self._march_stream()                             # after `out` was created / cleared on torch's stream
                self.ctx.render_clouds_device(pc, W, bands, out.data_ptr(), W * 8, stream)
                done()                                                          # the gather / interleave that follow run on torch's stream
            frame = _tiling.render_sharded(render_bands, H, W, self.rank, self.world_size, self.dist, tex.device)
            if frame is not None:
                self.textures[p_texture_to_update] = frame.view(torch.float16)
                self.last_frame = self.textures[p_texture_to_update]
            return self.last_frame if self.rank == 0 else None
        region = torch.empty((rh, rw, 4), dtype=torch.float16, device=tex.device)
        stream, done = self._march_stream()
        self.ctx.render_clouds_device(pc, rw, (rh, 0, 1, 1), region.data_ptr(), rw * 8, stream)
        done()
        tex[y0:y0 + rh, x0:x0 + rw] = region
        self.last_frame = tex
        return tex
