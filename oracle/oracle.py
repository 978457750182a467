"""ctypes wrapper of the CPU oracle (oracle/cloudsky_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as
the checker / reported baseline.  PARITY UNPINNED (see cloudsky_oracle.h): the reference has no golden
vectors and cannot be run; the oracle is cross-checked against oracle/numpy_restatement.py fixtures and is
bit-identical to the reference's own shader text executed under oracle/glsl_exec (a builder-written GLSL
stand-in: strong evidence about the transcription, not a pin).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Textures(C.Structure):
    _fields_ = [("large_rgba8", C.c_void_p), ("large_levels", C.c_int), ("small_rgb8", C.c_void_p),
                ("small_levels", C.c_int), ("weather_rgb8", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("primary_samples", C.c_uint64), ("incloud_samples", C.c_uint64), ("rays", C.c_uint64),
                ("rays_marched", C.c_uint64)]


def build():
    """(Re)build liboracle with gcc if missing or stale."""
    so = os.path.join(_HERE, "libcskoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("cloudsky_oracle.c", "cloudsky_oracle.h", "Makefile")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.csko_mip_offset.restype = C.c_size_t
        L.csko_mip_offset.argtypes = [C.c_int, C.c_int, C.c_int]
        L.csko_mip_total.restype = C.c_size_t
        L.csko_mip_total.argtypes = [C.c_int, C.c_int, C.c_int]
        L.csko_build_mips.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.csko_f2h.restype = C.c_uint16
        L.csko_f2h.argtypes = [C.c_float]
        L.csko_h2f.restype = C.c_float
        L.csko_h2f.argtypes = [C.c_uint16]
        L.csko_transmittance_lut.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.csko_sky_lut.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.csko_clouds.argtypes = [C.POINTER(Textures), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int,
                                  C.POINTER(Stats)]
        L.csko_clouds_bands.argtypes = [C.POINTER(Textures), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(Stats)]
        L.csko_composite.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.csko_composite_view.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.csko_hash_probe.restype = C.c_float
        L.csko_hash_probe.argtypes = [C.c_float] * 3
        L.csko_pixel_dir.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.csko_sky_lut_lookup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.csko_density_probe.restype = C.c_float
        L.csko_density_probe.argtypes = [C.POINTER(Textures), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.csko_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def build_mip_chain(level0, n, ch, levels):
    """level0: uint8 [n,n,n,ch] (z,y,x,c).  Returns the flat uint8 chain (level 0 first)."""
    L = lib()
    total = L.csko_mip_total(n, levels, ch)
    buf = np.zeros(total, np.uint8)
    buf[: n * n * n * ch] = np.ascontiguousarray(level0, np.uint8).reshape(-1)
    L.csko_build_mips(_ptr(buf), n, ch, levels)
    return buf


class OracleTextures:
    """Holds the mip-chained textures the oracle samples (keeps the numpy buffers alive)."""

    def __init__(self, large_rgba8, small_rgb8, weather_rgb8):
        self.large = build_mip_chain(large_rgba8, 128, 4, 8)
        self.small = build_mip_chain(small_rgb8, 32, 3, 6)
        self.weather = np.ascontiguousarray(weather_rgb8, np.uint8).reshape(512, 512, 3)
        self.c = Textures(_ptr(self.large), 8, _ptr(self.small), 6, _ptr(self.weather))

    @classmethod
    def from_chains(cls, large_chain_rgba8, small_chain_rgb8, weather_rgb8):
        """Caller-supplied mip chains (all levels back to back, level 0 first) instead of the 2x2x2 box chains: what a host binds with
        csky_set_noise_mips when it has the importer's own chains (perlworlnoise.tga.import:24, worlnoise.bmp.import:24)."""
        o = cls.__new__(cls)
        o.large = np.ascontiguousarray(large_chain_rgba8, np.uint8).reshape(-1).copy()
        o.small = np.ascontiguousarray(small_chain_rgb8, np.uint8).reshape(-1).copy()
        assert o.large.size == lib().csko_mip_total(128, 8, 4) and o.small.size == lib().csko_mip_total(32, 6, 3)
        o.weather = np.ascontiguousarray(weather_rgb8, np.uint8).reshape(512, 512, 3)
        o.c = Textures(_ptr(o.large), 8, _ptr(o.small), 6, _ptr(o.weather))
        return o


def transmittance_lut(w=256, h=64):
    out = np.zeros((h, w, 4), np.uint16)
    lib().csko_transmittance_lut(w, h, _ptr(out))
    return out.view(np.float16)


def sky_lut(sun_dir, trans, w=200, h=100):
    sun = np.asarray(sun_dir, np.float32)
    t = np.ascontiguousarray(trans).view(np.uint16)
    out = np.zeros((h, w, 4), np.uint16)
    lib().csko_sky_lut(w, h, _ptr(sun), _ptr(t), t.shape[1], t.shape[0], _ptr(out))
    return out.view(np.float16)


def clouds(tex, params, sky, rect=None, primary_steps=128, light_steps=6, nthreads=1, return_stats=False):
    """params: 28 float32 (clouds.glsl:18-40).  rect = (gx0, gy0, w, h) in gl_GlobalInvocationID space
    (default: the full texture_size frame).  Returns float16 [h, w, 4]."""
    p = np.ascontiguousarray(params, np.float32)
    assert p.size == 28
    if rect is None:
        rect = (0, 0, int(p[0]), int(p[1]))
    gx0, gy0, w, h = rect
    s = np.ascontiguousarray(sky).view(np.uint16)
    out = np.zeros((h, w, 4), np.uint16)
    st = Stats()
    lib().csko_clouds(C.byref(tex.c), _ptr(p), primary_steps, light_steps, _ptr(s), s.shape[1], s.shape[0], gx0, gy0,
                      w, h, _ptr(out), w * 8, nthreads, C.byref(st))
    img = out.view(np.float16)
    if return_stats:
        return img, dict(rays=st.rays, rays_marched=st.rays_marched, primary_samples=st.primary_samples,
                         incloud_samples=st.incloud_samples)
    return img


def clouds_bands(tex, params, sky, tile_w, bands, primary_steps=128, light_steps=6, nthreads=1):
    """Band-set form (bands = (band_rows, first_band, band_stride, n_bands)); returns (float16 [rows, tile_w, 4], stats)."""
    p = np.ascontiguousarray(params, np.float32)
    s = np.ascontiguousarray(sky).view(np.uint16)
    br, first, stride, n = [int(v) for v in bands]
    out = np.zeros((br * n, tile_w, 4), np.uint16)
    st = Stats()
    lib().csko_clouds_bands(C.byref(tex.c), _ptr(p), primary_steps, light_steps, _ptr(s), s.shape[1], s.shape[0], tile_w, br, first,
                            stride, n, _ptr(out), nthreads, C.byref(st))
    return out.view(np.float16), dict(rays=st.rays, rays_marched=st.rays_marched, primary_samples=st.primary_samples,
                                      incloud_samples=st.incloud_samples)


def composite(cloud_from, cloud_to, sky_from, sky_to, trans, light_dir, blend_amount=0.0, sun_disk_scale=2.0, out_w=256, out_h=128):
    """clouds.gdshader sky() on an equirectangular panorama; all inputs float16 [h, w, 4]; returns float16 [out_h, out_w, 4]."""
    a = [np.ascontiguousarray(x).view(np.uint16) for x in (cloud_from, cloud_to, sky_from, sky_to, trans)]
    ld = np.asarray(light_dir, np.float32)
    out = np.zeros((out_h, out_w, 4), np.uint16)
    lib().csko_composite(out_w, out_h, _ptr(a[0]), _ptr(a[1]), a[0].shape[1], a[0].shape[0], _ptr(a[2]), _ptr(a[3]), a[2].shape[1],
                         a[2].shape[0], _ptr(a[4]), a[4].shape[1], a[4].shape[0], blend_amount, sun_disk_scale, _ptr(ld), _ptr(out))
    return out.view(np.float16)


def composite_view(cloud_from, cloud_to, sky_from, sky_to, trans, light_dir, basis, fov_y_degrees, blend_amount=0.0, sun_disk_scale=2.0, out_w=256, out_h=144):
    """clouds.gdshader sky() per SCREEN pixel of a perspective camera (basis 3x3: columns = the camera's right / up / back axes)."""
    a = [np.ascontiguousarray(x).view(np.uint16) for x in (cloud_from, cloud_to, sky_from, sky_to, trans)]
    ld = np.asarray(light_dir, np.float32)
    b = np.ascontiguousarray(np.asarray(basis, np.float32).T.reshape(-1))                 # column-major
    out = np.zeros((out_h, out_w, 4), np.uint16)
    lib().csko_composite_view(out_w, out_h, _ptr(b), float(fov_y_degrees), _ptr(a[0]), _ptr(a[1]), a[0].shape[1], a[0].shape[0], _ptr(a[2]), _ptr(a[3]),
                              a[2].shape[1], a[2].shape[0], _ptr(a[4]), a[4].shape[1], a[4].shape[0], blend_amount, sun_disk_scale, _ptr(ld), _ptr(out))
    return out.view(np.float16)


def default_params(w, h, sun, coverage=0.2, density=0.05):
    """Default-config push-constant block (clouds_sky.tres:11-17; wind frozen), SURVEY A.2."""
    s = np.asarray(sun, np.float64)
    s = (s / np.linalg.norm(s)).astype(np.float32)
    return np.array([w, h, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0,
                     1.0, 1.0, 1.0, 0.0, 0.0, density, coverage, 0.0], np.float32)


def max_threads():
    return lib().csko_max_threads()
