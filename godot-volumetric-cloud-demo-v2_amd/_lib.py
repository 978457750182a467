"""ctypes binding of libcloudsky.so (include/cloudsky.h).  Fails loudly: a missing library or a missing GPU is
an error, never a silent fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_STATE = 0, -1, -2, -3, -4, -5


class CloudSkyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcloudsky error %d: %s" % (code, msg))
        self.code = code


class CloudParams(C.Structure):  # clouds.glsl:18-40
    _fields_ = [("f", C.c_float * 28)]


class SkyParams(C.Structure):  # sky-lut.glsl:12-18
    _fields_ = [("f", C.c_float * 8)]


class TransParams(C.Structure):  # transmittance-lut.glsl:12-15
    _fields_ = [("f", C.c_float * 4)]


class Bands(C.Structure):
    _fields_ = [("band_rows", C.c_int), ("first_band", C.c_int), ("band_stride", C.c_int), ("n_bands", C.c_int)]


class CompositeParams(C.Structure):
    _fields_ = [("out_w", C.c_int), ("out_h", C.c_int), ("cloud_w", C.c_int), ("cloud_h", C.c_int), ("sky_w", C.c_int), ("sky_h", C.c_int),
                ("blend_amount", C.c_float), ("sun_disk_scale", C.c_float), ("light_direction", C.c_float * 3)]


class CloudStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_samples", C.c_uint64), ("incloud_samples", C.c_uint64)]


class ShapeNoiseParams(C.Structure):
    """csky_shape_noise_params (include/cloudsky.h): the knobs of the stand-in shape-noise generator."""
    _fields_ = [("perlin_freq", C.c_int32), ("perlin_octaves", C.c_int32), ("worley_freq", C.c_int32), ("perlin_gain", C.c_float), ("dilate", C.c_float),
                ("centre", C.c_float), ("contrast", C.c_float), ("offset", C.c_float)]


MULTI_STATS_MAX = 16


class MultiStats(C.Structure):
    """csky_multi_stats (include/cloudsky_internal.h)."""
    _fields_ = [("n_devices", C.c_int32), ("staged", C.c_int32), ("all_peer", C.c_int32), ("groups", C.c_int32), ("frames_in_flight", C.c_int32), ("timing", C.c_int32),
                ("device_id", C.c_int32 * MULTI_STATS_MAX), ("peer_access", C.c_int32 * MULTI_STATS_MAX), ("march_ms", C.c_float * MULTI_STATS_MAX),
                ("copy_ms", C.c_float * MULTI_STATS_MAX)]


def shape_noise_params(**knobs):
    """The generator's defaults with the given fields replaced."""
    p = ShapeNoiseParams()
    lib().csky_shape_noise_default_params(C.byref(p))
    for k, v in knobs.items():
        if k not in dict(ShapeNoiseParams._fields_):
            raise TypeError("unknown shape-noise knob %r" % k)
        setattr(p, k, v)
    return p


# every symbol include/cloudsky.h and include/cloudsky_internal.h declare: (name, restype, argtypes); INTERNAL names the second header's (the lab bench)
SYMBOLS = [
    ("csky_abi_version", C.c_int, []),
    ("csky_device_count", C.c_int, []),
    ("csky_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    ("csky_destroy", None, [C.c_void_p]),
    ("csky_last_error", C.c_char_p, [C.c_void_p]),
    ("csky_set_noise", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_set_noise_mips", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_noise_inexact_coeffs", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("csky_encode_bc7", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("csky_encode_bc7_quality", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("csky_set_march", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("csky_set_early_out", C.c_int, [C.c_void_p, C.c_float]),
    ("csky_render_transmittance", C.c_int, [C.c_void_p, C.POINTER(TransParams), C.c_void_p]),
    ("csky_render_sky_lut", C.c_int, [C.c_void_p, C.POINTER(SkyParams), C.c_void_p]),
    ("csky_render_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    ("csky_render_sky_lut_device", C.c_int, [C.c_void_p, C.POINTER(SkyParams), C.c_void_p]),
    ("csky_render_clouds_device", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.POINTER(Bands), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("csky_copy_sky_lut_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_render_sky_lut_rows_device", C.c_int, [C.c_void_p, C.POINTER(SkyParams), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("csky_interleave_bands_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    ("csky_sync", C.c_int, [C.c_void_p]),
    ("csky_set_host_ring", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_submit_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    ("csky_collect", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("csky_poll", C.c_int, [C.c_void_p, C.c_int64]),
    ("csky_external_frame_import_fd", C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("csky_external_frame_import_semaphore_fd", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("csky_external_frame_signal", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_external_frame_fence", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_external_frame_ready", C.c_int, [C.c_void_p, C.c_void_p]),
    ("csky_external_frame_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("csky_external_frame_release", None, [C.c_void_p]),
    ("csky_read_transmittance", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("csky_read_sky_lut", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("csky_composite_sky", C.c_int, [C.c_void_p, C.POINTER(CompositeParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_composite_view", C.c_int, [C.c_void_p, C.POINTER(CompositeParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_time_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.POINTER(Bands), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(CloudStats)]),
    ("csky_get_cloud_stats", C.c_int, [C.c_void_p, C.POINTER(CloudStats)]),
    ("csky_set_kernel_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_get_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("csky_set_frames_in_flight", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_set_variant", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_variant_count", C.c_int, []),
    ("csky_set_schedule", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_set_height_window", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_set_segments", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_set_exact_cells", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_last_warning", C.c_char_p, [C.c_void_p]),
    ("csky_variant_name", C.c_char_p, [C.c_int]),
    ("csky_multi_create", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]),
    ("csky_multi_destroy", None, [C.c_void_p]),
    ("csky_multi_device_count", C.c_int, [C.c_void_p]),
    ("csky_multi_ctx", C.c_void_p, [C.c_void_p, C.c_int]),
    ("csky_multi_last_error", C.c_char_p, [C.c_void_p]),
    ("csky_multi_last_warning", C.c_char_p, [C.c_void_p]),
    ("csky_multi_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_multi_get_stats", C.c_int, [C.c_void_p, C.c_void_p]),
    ("csky_multi_set_noise", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_multi_set_noise_mips", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csky_multi_set_frames_in_flight", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_multi_set_groups", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_multi_set_staged", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_multi_set_march", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("csky_multi_render_sky_lut", C.c_int, [C.c_void_p, C.POINTER(SkyParams)]),
    ("csky_multi_render_clouds_device", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("csky_multi_render_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    ("csky_multi_sync", C.c_int, [C.c_void_p]),
    ("csky_multi_set_host_ring", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_multi_submit_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    ("csky_multi_collect", C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("csky_load_bmp_rgb8", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    ("csky_load_tga_rgba8", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    ("csky_strip_to_volume", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("csky_generate_shape_noise", C.c_int, [C.c_uint32, C.c_int, C.c_void_p]),
    ("csky_generate_shape_noise_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    ("csky_shape_noise_default_params", None, [C.c_void_p]),
    ("csky_check_shape_noise_params", C.c_int, [C.c_void_p, C.c_int]),
    ("csky_generate_shape_noise_tuned", C.c_int, [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    ("csky_generate_shape_noise_tuned_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    ("csky_generate_detail_noise", C.c_int, [C.c_uint32, C.c_int, C.c_void_p]),
    ("csky_generate_detail_noise_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    ("csky_build_mips_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("csky_read_baked_texture", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("csky_test_sqrt_shell", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("csky_census_clouds", C.c_int, [C.c_void_p, C.POINTER(CloudParams), C.c_int, C.POINTER(Bands), C.c_void_p, C.c_int]),
    ("csky_mip_offset", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("csky_build_mips", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    ("csky_decode_bc7", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    ("csky_load_ctex", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    ("csky_load_ctex3d", C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    ("csky_assets_last_error", C.c_char_p, []),
]


DEFAULT_VARIANT = 3   # include/cloudsky.h CSKY_DEFAULT_VARIANT ("compact"); set_variant(-1) selects it
ABI_VERSION = 7       # include/cloudsky.h CSKY_ABI_VERSION


def library_path():
    # CSKY_LIBRARY: explicit path of an alternative build (A/B timing of kernel experiments); default = the in-tree build
    return os.environ.get("CSKY_LIBRARY") or os.path.join(_HERE, "libcloudsky.so")


def lib():
    """Load libcloudsky.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise CloudSkyError(ERR_IO, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                        "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        try:
            # torch bundles its own libamdhip64; it must be the FIRST HIP runtime in the process so that libcloudsky's
            # DT_NEEDED resolves to the same (already loaded) runtime and device pointers / streams can be shared.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError = ABI mismatch, surfaced loudly
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def cloud_params(values):
    p = CloudParams()
    v = np.asarray(values, np.float32).reshape(-1)
    if v.size != 28:
        raise ValueError("cloud push-constant block is 28 floats (clouds.glsl:18-40), got %d" % v.size)
    for i in range(28):
        p.f[i] = float(v[i])
    return p


class Context:
    """One csky_ctx = one GPU.  Thin, explicit wrapper; raises CloudSkyError on any non-zero return."""

    def __init__(self, device_id=0, _borrowed=None):
        self._L = lib()
        self._owned = _borrowed is None
        if _borrowed is not None:                 # a context owned by a MultiContext (csky_multi_ctx): never destroyed from here
            self._h = C.c_void_p(_borrowed)
            self.device_id = int(device_id)
            return
        h = C.c_void_p()
        rc = self._L.csky_create(C.byref(h), int(device_id))
        if rc != OK:
            raise CloudSkyError(rc, (self._L.csky_last_error(None) or b"").decode())
        self._h = h
        self.device_id = int(device_id)

    def _chk(self, rc):
        if rc != OK:
            raise CloudSkyError(rc, (self._L.csky_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.csky_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- inputs
    def set_noise(self, large_rgba8, small_rgb8, weather_rgb8):
        a = np.ascontiguousarray(large_rgba8, np.uint8)
        b = np.ascontiguousarray(small_rgb8, np.uint8)
        c = np.ascontiguousarray(weather_rgb8, np.uint8)
        if a.size != 128 ** 3 * 4 or b.size != 32 ** 3 * 3 or c.size != 512 * 512 * 3:
            raise ValueError("set_noise: expected 128^3 RGBA8, 32^3 RGB8, 512^2 RGB8")
        self._chk(self._L.csky_set_noise(self._h, _ptr(a), _ptr(b), _ptr(c)))   # (textures that do not fit the fp16 cells are marched on exact fp32 cells: last_warning() says so)

    def set_noise_mips(self, large_chain_rgba8, small_chain_rgb8, weather_rgb8):
        """csky_set_noise_mips: full mip chains supplied by the caller (all levels back to back), e.g. the importer's own from
        assets.load_ctex3d; no box filter is applied by the library."""
        a = np.ascontiguousarray(large_chain_rgba8, np.uint8)
        b = np.ascontiguousarray(small_chain_rgb8, np.uint8)
        c = np.ascontiguousarray(weather_rgb8, np.uint8)
        if a.size != self._L.csky_mip_offset(128, 8, 4) or b.size != self._L.csky_mip_offset(32, 6, 3) or c.size != 512 * 512 * 3:
            raise ValueError("set_noise_mips: expected the 8-level 128^3 RGBA8 chain, the 6-level 32^3 RGB8 chain, 512^2 RGB8")
        self._chk(self._L.csky_set_noise_mips(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def encode_bc7(self, images, quality=0):
        """csky_encode_bc7[_quality]: [n, h, w, 4] (or [h, w, 4]) uint8 -> [n, ceil(h/4), ceil(w/4), 16] uint8 BC7 blocks, encoded on the GPU
        (quality 1: more partitions + end-point coordinate descent, the sensitivity study's second encoder)."""
        a = np.ascontiguousarray(images, np.uint8)
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[3] != 4:
            raise ValueError("encode_bc7: expected [n, h, w, 4] uint8")
        n, h, w = a.shape[:3]
        out = np.zeros((n, (h + 3) // 4, (w + 3) // 4, 16), np.uint8)
        self._chk(self._L.csky_encode_bc7_quality(self._h, _ptr(a), w, h, n, int(quality), _ptr(out)))
        return out

    def noise_inexact_coeffs(self):
        """Finite-difference coefficients of the bound textures that fp16 could not hold exactly (0 for natural noise)."""
        n = C.c_uint64()
        self._chk(self._L.csky_noise_inexact_coeffs(self._h, C.byref(n)))
        return n.value

    def set_march(self, primary_steps=128, light_steps=6):
        self._chk(self._L.csky_set_march(self._h, primary_steps, light_steps))

    def set_early_out(self, eps):
        self._chk(self._L.csky_set_early_out(self._h, float(eps)))

    def set_schedule(self, mode):
        self._chk(self._L.csky_set_schedule(self._h, int(mode)))

    def set_height_window(self, enabled):
        self._chk(self._L.csky_set_height_window(self._h, int(bool(enabled))))

    def set_segments(self, n):
        self._chk(self._L.csky_set_segments(self._h, int(n)))

    def set_variant(self, v):
        self._chk(self._L.csky_set_variant(self._h, int(v)))

    def set_exact_cells(self, mode):
        """1: build and march the exact fp32-coefficient cells at the next set_noise whatever the textures need; 0: only when a coefficient does not fit fp16."""
        self._chk(self._L.csky_set_exact_cells(self._h, int(mode)))

    def last_warning(self):
        return (self._L.csky_last_warning(self._h) or b"").decode()

    # ---- kernels, host-buffer forms
    def render_transmittance(self, w=256, h=64):
        p = TransParams()
        p.f[0], p.f[1] = float(w), float(h)
        out = np.zeros((h, w, 4), np.uint16)
        self._chk(self._L.csky_render_transmittance(self._h, C.byref(p), _ptr(out)))
        return out.view(np.float16)

    def render_sky_lut(self, sun_dir, w=200, h=100, readback=True):
        p = SkyParams()
        p.f[0], p.f[1] = float(w), float(h)
        p.f[4], p.f[5], p.f[6] = [float(x) for x in sun_dir]
        out = np.zeros((h, w, 4), np.uint16) if readback else None
        self._chk(self._L.csky_render_sky_lut(self._h, C.byref(p), _ptr(out) if readback else None))
        return out.view(np.float16) if readback else None

    def render_clouds(self, params, tile_w=None, tile_h=None):
        p = cloud_params(params)
        w = int(p.f[0]) if tile_w is None else int(tile_w)
        h = int(p.f[1]) if tile_h is None else int(tile_h)
        out = np.zeros((h, w, 4), np.uint16)
        self._chk(self._L.csky_render_clouds(self._h, C.byref(p), w, h, _ptr(out), w * 8))
        return out.view(np.float16)

    # ---- device-buffer forms
    def render_sky_lut_device(self, sun_dir, w=200, h=100, stream=None):
        p = SkyParams()
        p.f[0], p.f[1] = float(w), float(h)
        p.f[4], p.f[5], p.f[6] = [float(x) for x in sun_dir]
        self._chk(self._L.csky_render_sky_lut_device(self._h, C.byref(p), C.c_void_p(stream or 0)))

    def render_sky_lut_rows_device(self, sun_dir, first_row, row_stride, d_rows_out, capacity_bytes, w=200, h=100, stream=None):
        """One rank's rows first_row::row_stride of the sky LUT (N processes splitting a frame), compact RGBA16F into the caller's device buffer
        on `stream`; the context keeps no LUT, render_clouds_device renders the texels its frame set-up needs itself."""
        p = SkyParams()
        p.f[0], p.f[1] = float(w), float(h)
        p.f[4], p.f[5], p.f[6] = [float(x) for x in sun_dir]
        self._chk(self._L.csky_render_sky_lut_rows_device(self._h, C.byref(p), int(first_row), int(row_stride), C.c_void_p(int(d_rows_out)),
                                                          C.c_size_t(int(capacity_bytes)), C.c_void_p(stream or 0)))

    def render_clouds_device(self, params, tile_w, bands, d_out, pitch_bytes, stream=None):
        p = cloud_params(params)
        b = Bands(*[int(x) for x in bands])
        self._chk(self._L.csky_render_clouds_device(self._h, C.byref(p), int(tile_w), C.byref(b), C.c_void_p(int(d_out)), int(pitch_bytes),
                                                    C.c_void_p(stream or 0)))

    def interleave_bands_device(self, d_gathered, member_stride_bytes, members, band_bytes, total_bands, d_frame, stream=None):
        """The gathering rank's interleave: frame band k = member k % members, local band k // members of the gathered rank-major buffer."""
        self._chk(self._L.csky_interleave_bands_device(self._h, C.c_void_p(int(d_gathered)), C.c_size_t(int(member_stride_bytes)), int(members), C.c_size_t(int(band_bytes)),
                                                       int(total_bands), C.c_void_p(int(d_frame)), C.c_void_p(stream or 0)))

    def copy_sky_lut_device(self, d_out, stream=None):
        """Async device copy of the sky LUT rendered last (w*h*8 bytes of RGBA16F) into a caller-owned device buffer."""
        self._chk(self._L.csky_copy_sky_lut_device(self._h, C.c_void_p(int(d_out)), C.c_void_p(stream or 0)))

    def sync(self):
        self._chk(self._L.csky_sync(self._h))

    def import_external_frame(self, fd, allocation_bytes, offset_bytes, frame_bytes):
        """Zero-copy interop (csky_external_frame_import_fd): memory another API allocated, handed over as a POSIX fd (the library owns the fd on
        success).  Returns an ExternalFrame whose `.ptr` is usable as d_out of render_clouds_device."""
        ef, dptr = C.c_void_p(), C.c_void_p()
        self._chk(self._L.csky_external_frame_import_fd(self._h, int(fd), C.c_size_t(int(allocation_bytes)), C.c_size_t(int(offset_bytes)), C.c_size_t(int(frame_bytes)),
                                                        C.byref(ef), C.byref(dptr)))
        return ExternalFrame(self, ef, dptr.value)

    def census_clouds(self, params, tile_w, bands, n=256):
        """Basic-block execution counts of one launch (non-zero only with the census build of the library, tools/isa_profile.py)."""
        p = cloud_params(params)
        b = Bands(*[int(x) for x in bands])
        out = np.zeros(n, np.uint32)
        self._chk(self._L.csky_census_clouds(self._h, C.byref(p), int(tile_w), C.byref(b), _ptr(out), int(n)))
        return out

    # ---- asynchronous host form (pinned ring)
    def set_host_ring(self, slots):
        self._chk(self._L.csky_set_host_ring(self._h, int(slots)))

    def submit_clouds(self, params, tile_w=None, tile_h=None):
        """Enqueue march + copy into a pinned ring slot; returns the ticket at once."""
        p = cloud_params(params)
        w = int(p.f[0]) if tile_w is None else int(tile_w)
        h = int(p.f[1]) if tile_h is None else int(tile_h)
        t = C.c_int64(-1)
        self._chk(self._L.csky_submit_clouds(self._h, C.byref(p), w, h, C.byref(t)))
        self._shapes = getattr(self, "_shapes", {})
        self._shapes[t.value] = (h, w)
        return t.value

    def collect(self, ticket, copy=True):
        """Wait for the frame of `ticket`: float16 [h, w, 4].  copy=False returns a VIEW of the pinned ring slot (valid until the slot is reused)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.csky_collect(self._h, int(ticket), C.byref(ptr), C.byref(n)))
        h, w = self._shapes.pop(int(ticket))
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(h, w, 4))
        return (a.copy() if copy else a).view(np.float16)

    def poll(self, ticket):
        rc = self._L.csky_poll(self._h, int(ticket))
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def read_transmittance(self):
        w, h = C.c_int(), C.c_int()
        self._chk(self._L.csky_read_transmittance(self._h, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value, 4), np.uint16)
        self._chk(self._L.csky_read_transmittance(self._h, _ptr(out), C.byref(w), C.byref(h)))
        return out.view(np.float16)

    def read_sky_lut(self):
        w, h = C.c_int(), C.c_int()
        self._chk(self._L.csky_read_sky_lut(self._h, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value, 4), np.uint16)
        self._chk(self._L.csky_read_sky_lut(self._h, _ptr(out), C.byref(w), C.byref(h)))
        return out.view(np.float16)

    def composite_sky(self, cloud_from, cloud_to, sky_from, sky_to, light_dir, blend_amount=0.0, sun_disk_scale=2.0, out_w=2048, out_h=1024):
        """clouds.gdshader sky() on an equirectangular panorama; inputs float16 [h, w, 4] host arrays."""
        a = [np.ascontiguousarray(x).view(np.uint16) for x in (cloud_from, cloud_to, sky_from, sky_to)]
        p = CompositeParams(out_w, out_h, a[0].shape[1], a[0].shape[0], a[2].shape[1], a[2].shape[0], float(blend_amount), float(sun_disk_scale))
        for k in range(3):
            p.light_direction[k] = float(light_dir[k])
        out = np.zeros((out_h, out_w, 4), np.uint16)
        self._chk(self._L.csky_composite_sky(self._h, C.byref(p), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(out)))
        return out.view(np.float16)

    def composite_view(self, cloud_from, cloud_to, sky_from, sky_to, light_dir, basis, fov_y_degrees, blend_amount=0.0, sun_disk_scale=2.0, out_w=1152, out_h=648):
        """clouds.gdshader sky() per SCREEN pixel of a perspective camera (basis: 3x3, columns = the camera's right / up / back axes)."""
        a = [np.ascontiguousarray(x).view(np.uint16) for x in (cloud_from, cloud_to, sky_from, sky_to)]
        p = CompositeParams(out_w, out_h, a[0].shape[1], a[0].shape[0], a[2].shape[1], a[2].shape[0], float(blend_amount), float(sun_disk_scale))
        for k in range(3):
            p.light_direction[k] = float(light_dir[k])
        v = (C.c_float * 10)(*([float(x) for x in np.asarray(basis, np.float32).T.reshape(-1)] + [float(fov_y_degrees)]))   # column-major basis, then the fov
        out = np.zeros((out_h, out_w, 4), np.uint16)
        self._chk(self._L.csky_composite_view(self._h, C.byref(p), C.cast(v, C.c_void_p), _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(out)))
        return out.view(np.float16)

    def generate_shape_noise(self, seed=1, n=128, **knobs):
        """GPU bake of the stand-in shape volume: uint8 [n, n, n, 4], byte-identical to assets.generate_shape_noise (knobs: ShapeNoiseParams fields)."""
        vol = np.zeros((n, n, n, 4), np.uint8)
        p = shape_noise_params(**knobs)
        self._chk(self._L.csky_generate_shape_noise_tuned_device(self._h, seed, n, C.byref(p), _ptr(vol)))
        return vol

    def generate_detail_noise(self, seed=1, n=32):
        """GPU bake of a generated detail volume: uint8 [n, n, n, 3], byte-identical to assets.generate_detail_noise."""
        vol = np.zeros((n, n, n, 3), np.uint8)
        self._chk(self._L.csky_generate_detail_noise_device(self._h, seed, n, _ptr(vol)))
        return vol

    def build_mips(self, level0, levels):
        """2x2x2 box mip chain on the GPU (flat uint8, level 0 first): byte-identical to assets.build_mips."""
        level0 = np.ascontiguousarray(level0, np.uint8)
        n, ch = level0.shape[0], level0.shape[3]
        buf = np.zeros(self._L.csky_mip_offset(n, levels, ch), np.uint8)
        buf[: level0.size] = level0.reshape(-1)
        self._chk(self._L.csky_build_mips_device(self._h, _ptr(buf), n, ch, levels))
        return buf

    def test_sqrt_shell(self, x):
        """Test hook: cloud_core.h::sqrt_shell on the device over a float32 array."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros_like(x)
        self._chk(self._L.csky_test_sqrt_shell(self._h, _ptr(x), _ptr(out), x.size))
        return out

    def read_baked_texture(self, which):
        """Test hook: the device layouts / mip chains csky_set_noise built, as raw bytes (0 shape, 1 detail, 2 weather, 3 / 4 8-bit chains)."""
        n = C.c_size_t()
        self._chk(self._L.csky_read_baked_texture(self._h, int(which), None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        self._chk(self._L.csky_read_baked_texture(self._h, int(which), _ptr(out), out.nbytes, C.byref(n)))
        return out

    # ---- measurement
    def time_clouds(self, params, tile_w, bands, warmup=2, iters=10):
        p = cloud_params(params)
        b = Bands(*[int(x) for x in bands])
        ms = C.c_float()
        st = CloudStats()
        self._chk(self._L.csky_time_clouds(self._h, C.byref(p), int(tile_w), C.byref(b), warmup, iters, C.byref(ms), C.byref(st)))
        return ms.value, dict(rays=st.rays, primary_samples=st.primary_samples, incloud_samples=st.incloud_samples)

    def set_frames_in_flight(self, frames):
        self._chk(self._L.csky_set_frames_in_flight(self._h, int(frames)))

    def set_kernel_timing(self, enabled=True):
        self._chk(self._L.csky_set_kernel_timing(self._h, int(bool(enabled))))

    def kernel_ms(self):
        """(sum of cloud-kernel durations in ms, launches) since the last call; waits for those launches."""
        ms, n = C.c_float(), C.c_int()
        self._chk(self._L.csky_get_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def cloud_stats(self):
        st = CloudStats()
        self._chk(self._L.csky_get_cloud_stats(self._h, C.byref(st)))
        return dict(rays=st.rays, primary_samples=st.primary_samples, incloud_samples=st.incloud_samples)


class ExternalFrame:
    """A frame that lives in imported memory.  Ordering is the host-side fence (ROCm 7.2 on Linux refuses external semaphores)."""

    def __init__(self, ctx, handle, ptr):
        self._ctx, self._h, self.ptr = ctx, handle, ptr

    def fence(self, stream=None):
        self._ctx._chk(self._ctx._L.csky_external_frame_fence(self._ctx._h, self._h, C.c_void_p(stream or 0)))

    def ready(self):
        rc = self._ctx._L.csky_external_frame_ready(self._ctx._h, self._h)
        if rc < 0:
            self._ctx._chk(rc)
        return bool(rc)

    def wait(self):
        self._ctx._chk(self._ctx._L.csky_external_frame_wait(self._ctx._h, self._h))

    def release(self):
        if self._h is not None:
            self._ctx._L.csky_external_frame_release(self._h)
            self._h, self.ptr = None, 0


class MultiContext:
    """csky_multi: the GPUs of one node behind one handle, one host thread (include/cloudsky.h).  Device i of n renders the
    8-row bands i, i+n, ... and stores them straight into the frame on the first device (xGMI peer access)."""

    def __init__(self, device_ids):
        self._L = lib()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        rc = self._L.csky_multi_create(C.byref(h), ids, len(device_ids))
        if rc != OK:
            raise CloudSkyError(rc, (self._L.csky_multi_last_error(None) or b"").decode())
        self._h = h
        self.device_ids = [int(d) for d in device_ids]

    def _chk(self, rc):
        if rc != OK:
            raise CloudSkyError(rc, (self._L.csky_multi_last_error(self._h) or b"").decode())

    def __len__(self):
        return self._L.csky_multi_device_count(self._h)

    def last_warning(self):
        """"" or what csky_multi_create fell back on (a device without peer access to the first: staged copies, whole LUT on the first device)."""
        return (self._L.csky_multi_last_warning(self._h) or b"").decode()

    def set_timing(self, enabled=True):
        self._chk(self._L.csky_multi_set_timing(self._h, 1 if enabled else 0))

    def stats(self):
        """Preconditions + (set_timing) the last frame's per-device march / peer-copy milliseconds; waits for the handle's work."""
        st = MultiStats()
        self._chk(self._L.csky_multi_get_stats(self._h, C.byref(st)))
        n = min(st.n_devices, MULTI_STATS_MAX)
        return dict(n_devices=st.n_devices, staged=bool(st.staged), all_peer=bool(st.all_peer), groups=st.groups, frames_in_flight=st.frames_in_flight, timing=bool(st.timing),
                    device_id=list(st.device_id[:n]), peer_access=list(st.peer_access[:n]), march_ms=[round(float(x), 4) for x in st.march_ms[:n]],
                    copy_ms=[round(float(x), 4) for x in st.copy_ms[:n]])

    def ctx(self, i):
        """The per-device context (borrowed: per-context settings such as set_variant / set_schedule go through it)."""
        p = self._L.csky_multi_ctx(self._h, int(i))
        if not p:
            raise IndexError(i)
        return Context(self.device_ids[i], _borrowed=p)

    def close(self):
        if getattr(self, "_h", None):
            self._L.csky_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def set_noise(self, large_rgba8, small_rgb8, weather_rgb8):
        a, b, c = (np.ascontiguousarray(x, np.uint8) for x in (large_rgba8, small_rgb8, weather_rgb8))
        if a.size != 128 ** 3 * 4 or b.size != 32 ** 3 * 3 or c.size != 512 * 512 * 3:
            raise ValueError("set_noise: expected 128^3 RGBA8, 32^3 RGB8, 512^2 RGB8")
        self._chk(self._L.csky_multi_set_noise(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def set_noise_mips(self, large_chain_rgba8, small_chain_rgb8, weather_rgb8):
        a, b, c = (np.ascontiguousarray(x, np.uint8) for x in (large_chain_rgba8, small_chain_rgb8, weather_rgb8))
        if a.size != self._L.csky_mip_offset(128, 8, 4) or b.size != self._L.csky_mip_offset(32, 6, 3) or c.size != 512 * 512 * 3:
            raise ValueError("set_noise_mips: expected the 8-level 128^3 RGBA8 chain, the 6-level 32^3 RGB8 chain, 512^2 RGB8")
        self._chk(self._L.csky_multi_set_noise_mips(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def set_march(self, primary_steps=128, light_steps=6):
        self._chk(self._L.csky_multi_set_march(self._h, primary_steps, light_steps))

    def set_frames_in_flight(self, frames):
        """2..8: consecutive render_clouds_device calls rotate that many consumer streams (per frame group); every device does too."""
        self._chk(self._L.csky_multi_set_frames_in_flight(self._h, int(frames)))

    def set_groups(self, groups):
        """Frame groups for throughput workloads: consecutive frames go to `groups` groups of len(self)/groups devices in turn."""
        self._chk(self._L.csky_multi_set_groups(self._h, int(groups)))

    def set_staged(self, staged):
        """True: local band buffers + one strided peer copy per device instead of in-place peer stores from inside the march."""
        self._chk(self._L.csky_multi_set_staged(self._h, 1 if staged else 0))

    def render_sky_lut(self, sun_dir, w=200, h=100):
        p = SkyParams()
        p.f[0], p.f[1] = float(w), float(h)
        p.f[4], p.f[5], p.f[6] = [float(x) for x in sun_dir]
        self._chk(self._L.csky_multi_render_sky_lut(self._h, C.byref(p)))

    def render_clouds(self, params, tile_w=None, tile_h=None):
        p = cloud_params(params)
        w = int(p.f[0]) if tile_w is None else int(tile_w)
        h = int(p.f[1]) if tile_h is None else int(tile_h)
        out = np.zeros((h, w, 4), np.uint16)
        self._chk(self._L.csky_multi_render_clouds(self._h, C.byref(p), w, h, _ptr(out), w * 8))
        return out.view(np.float16)

    def render_clouds_device(self, params, tile_w, tile_h, d_out, pitch_bytes, stream=None):
        p = cloud_params(params)
        self._chk(self._L.csky_multi_render_clouds_device(self._h, C.byref(p), int(tile_w), int(tile_h), C.c_void_p(int(d_out)), int(pitch_bytes),
                                                          C.c_void_p(stream or 0)))

    def sync(self):
        self._chk(self._L.csky_multi_sync(self._h))

    def set_host_ring(self, slots):
        self._chk(self._L.csky_multi_set_host_ring(self._h, int(slots)))

    def submit_clouds(self, params, tile_w=None, tile_h=None):
        p = cloud_params(params)
        w = int(p.f[0]) if tile_w is None else int(tile_w)
        h = int(p.f[1]) if tile_h is None else int(tile_h)
        t = C.c_int64(-1)
        self._chk(self._L.csky_multi_submit_clouds(self._h, C.byref(p), w, h, C.byref(t)))
        self._shapes = getattr(self, "_shapes", {})
        self._shapes[t.value] = (h, w)
        return t.value

    def collect(self, ticket, copy=True):
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.csky_multi_collect(self._h, int(ticket), C.byref(ptr), C.byref(n)))
        h, w = self._shapes.pop(int(ticket))
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(h, w, 4))
        return (a.copy() if copy else a).view(np.float16)
