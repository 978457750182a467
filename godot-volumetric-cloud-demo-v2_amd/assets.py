"""Asset layer: what the reference gets from Godot's importers (weather.bmp.import, worlnoise.bmp.import,
perlworlnoise.tga.import).  Host-only functions of libcloudsky; usable without a GPU."""
import ctypes as C
import hashlib
import os

import numpy as np

from . import _lib

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
SHAPE_SEED = 1
# SHA-256 of csky_generate_shape_noise(seed=1, n=128): integer hashing + IEEE +,-,*,/,sqrt only => machine independent
SHAPE_SHA256 = None  # filled in by tests/test_assets.py expectations (see tests/golden/INPUTS.txt)


def _chk(rc):
    if rc != 0:
        raise _lib.CloudSkyError(rc, (_lib.lib().csky_assets_last_error() or b"").decode())


def load_bmp_rgb8(path):
    L = _lib.lib()
    w, h = C.c_int(), C.c_int()
    _chk(L.csky_load_bmp_rgb8(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 3), np.uint8)
    _chk(L.csky_load_bmp_rgb8(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), out.nbytes))
    return out


def load_tga_rgba8(path):
    L = _lib.lib()
    w, h = C.c_int(), C.c_int()
    _chk(L.csky_load_tga_rgba8(path.encode(), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    _chk(L.csky_load_tga_rgba8(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), out.nbytes))
    return out


def load_shape_noise_tga(path, n=128):
    """The reference's own shape volume, if available: perlworlnoise.tga = n horizontal slices of n x n (perlworlnoise.tga.import:26-27)."""
    img = load_tga_rgba8(path)
    if img.shape[0] != n or img.shape[1] != n * n:
        raise ValueError("expected a %d x %d strip, got %d x %d" % (n * n, n, img.shape[1], img.shape[0]))
    return strip_to_volume(img, n)


def strip_to_volume(strip, n):
    """Godot 3-D import, slices/horizontal=n, vertical=1: voxel (x,y,z) = strip[y][n*z + x] -> [z,y,x,ch]."""
    strip = np.ascontiguousarray(strip, np.uint8)
    ch = strip.shape[2]
    assert strip.shape[0] == n and strip.shape[1] == n * n
    vol = np.zeros((n, n, n, ch), np.uint8)
    _chk(_lib.lib().csky_strip_to_volume(strip.ctypes.data_as(C.c_void_p), n, ch, vol.ctypes.data_as(C.c_void_p)))
    return vol


def generate_shape_noise(seed=SHAPE_SEED, n=128, **knobs):
    """Deterministic stand-in for the missing cloud_sky/perlworlnoise.tga: [z,y,x,4] uint8.  knobs: fields of _lib.ShapeNoiseParams (README.md:30
    TODO 3, a generator that can be tweaked); none = the calibration every benchmark and parity input uses."""
    vol = np.zeros((n, n, n, 4), np.uint8)
    p = _lib.shape_noise_params(**knobs)
    _chk(_lib.lib().csky_generate_shape_noise_tuned(seed, n, C.byref(p), vol.ctypes.data_as(C.c_void_p)))
    return vol


def generate_detail_noise(seed=1, n=32):
    """A generated detail volume in the role of worlnoise.bmp: [z,y,x,3] uint8 (three inverted-Worley fBm channels)."""
    vol = np.zeros((n, n, n, 3), np.uint8)
    _chk(_lib.lib().csky_generate_detail_noise(seed, n, vol.ctypes.data_as(C.c_void_p)))
    return vol


def build_mips(level0, levels):
    level0 = np.ascontiguousarray(level0, np.uint8)
    n, ch = level0.shape[0], level0.shape[3]
    L = _lib.lib()
    total = L.csky_mip_offset(n, levels, ch)
    buf = np.zeros(total, np.uint8)
    buf[: level0.size] = level0.reshape(-1)
    _chk(L.csky_build_mips(buf.ctypes.data_as(C.c_void_p), n, ch, levels))
    return buf


def decode_bc7(blocks, w, h):
    """BC7 / BPTC RGBA UNORM blocks (16 bytes per 4x4 pixels, row-major) -> [h, w, 4] uint8 (csky_decode_bc7)."""
    blocks = np.ascontiguousarray(np.frombuffer(blocks, np.uint8) if isinstance(blocks, (bytes, bytearray)) else blocks, np.uint8)
    need = ((w + 3) // 4) * ((h + 3) // 4) * 16
    if blocks.size < need:
        raise ValueError("decode_bc7: %d bytes of blocks, %d needed for %dx%d" % (blocks.size, need, w, h))
    out = np.zeros((h, w, 4), np.uint8)
    _chk(_lib.lib().csky_decode_bc7(blocks.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p)))
    return out


def load_ctex(path):
    """Godot CompressedTexture2D (.ctex, e.g. .godot/imported/weather.bmp-<md5>.bptc.ctex) -> list of [h, w, 4] uint8 levels."""
    L = _lib.lib()
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    _chk(L.csky_load_ctex(path.encode(), C.byref(w), C.byref(h), C.byref(n), None, 0))
    dims = [(max(1, h.value >> l), max(1, w.value >> l)) for l in range(n.value)]
    buf = np.zeros(sum(a * b * 4 for a, b in dims), np.uint8)
    _chk(L.csky_load_ctex(path.encode(), C.byref(w), C.byref(h), C.byref(n), buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    out, o = [], 0
    for a, b in dims:
        out.append(buf[o:o + a * b * 4].reshape(a, b, 4)); o += a * b * 4
    return out


def load_ctex3d(path):
    """Godot CompressedTexture3D (.ctex3d) -> list of [d, h, w, 4] uint8 levels (level 0 first, then the importer's own mips)."""
    L = _lib.lib()
    w, h, d, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _chk(L.csky_load_ctex3d(path.encode(), C.byref(w), C.byref(h), C.byref(d), C.byref(n), None, 0))
    dims = [(max(1, d.value >> l), max(1, h.value >> l), max(1, w.value >> l)) for l in range(n.value)]
    buf = np.zeros(sum(a * b * c * 4 for a, b, c in dims), np.uint8)
    _chk(L.csky_load_ctex3d(path.encode(), C.byref(w), C.byref(h), C.byref(d), C.byref(n), buf.ctypes.data_as(C.c_void_p), buf.nbytes))
    out, o = [], 0
    for a, b, c in dims:
        out.append(buf[o:o + a * b * c * 4].reshape(a, b, c, 4)); o += a * b * c * 4
    return out


def chains_from_godot_import(large_ctex3d, small_ctex3d, weather_ctex):
    """The three imported files of a Godot project (.godot/imported/perlworlnoise.tga-*.bptc.ctex3d, worlnoise.bmp-*.bptc.ctex3d,
    weather.bmp-*.bptc.ctex) -> (large chain RGBA8, small chain RGB8, weather RGB8) for Context.set_noise_mips: the texels the
    reference's samplers see (decoded BC7 blocks, the importer's mip chains)."""
    large = load_ctex3d(large_ctex3d)
    small = load_ctex3d(small_ctex3d)
    weather = load_ctex(weather_ctex)[0]
    if large[0].shape[:3] != (128, 128, 128) or len(large) != 8 or small[0].shape[:3] != (32, 32, 32) or len(small) != 6 or weather.shape[:2] != (512, 512):
        raise ValueError("unexpected sizes: large %s x%d, small %s x%d, weather %s" % (large[0].shape, len(large), small[0].shape, len(small), weather.shape))
    return (np.concatenate([l.reshape(-1) for l in large]), np.concatenate([l[..., :3].reshape(-1) for l in small]), np.ascontiguousarray(weather[..., :3]))


def vram_compressed_chains(ctx, large, small, weather, quality=0):
    """What compress/mode=2 of the three *.import files does to the inputs, with THIS library's encoder in the importer's place (it is not the
    engine's: see csky_encode_bc7): the box-filtered mip chains of the two volumes (mipmaps/generate=true) and the weather map, every level
    BC7-encoded slice by slice on the GPU of `ctx` and decoded again -> (large chain RGBA8, small chain RGB8, weather RGB8) for
    Context.set_noise_mips, plus the per-texture PSNR of the round trip."""
    def psnr(a, b):
        d = a.astype(np.float64) - b.astype(np.float64)
        m = float((d * d).mean())
        return float("inf") if m == 0 else 10.0 * np.log10(255.0 * 255.0 / m)

    def roundtrip(img4):                                        # [n, h, w, 4]
        n, h, w = img4.shape[:3]
        blocks = ctx.encode_bc7(img4, quality)
        return np.stack([decode_bc7(blocks[i], w, h) for i in range(n)])

    def volume_chain(level0, levels, ch):
        n = level0.shape[0]
        chain = build_mips(level0, levels)
        out, o = [], 0
        for l in range(levels):
            m = n >> l
            lv = chain[o:o + m * m * m * ch].reshape(m, m, m, ch); o += m * m * m * ch
            rgba = np.concatenate([lv, np.full((m, m, m, 1), 255, np.uint8)], -1) if ch == 3 else lv
            out.append((lv, roundtrip(rgba)[..., :ch]))
        return out

    big = volume_chain(np.ascontiguousarray(large, np.uint8), 8, 4)
    sml = volume_chain(np.ascontiguousarray(small, np.uint8), 6, 3)
    w0 = np.ascontiguousarray(weather, np.uint8)
    w4 = np.concatenate([w0, np.full(w0.shape[:2] + (1,), 255, np.uint8)], -1)
    wq = roundtrip(w4[None])[0][..., :3]
    stats = {"large_psnr_level0": psnr(*big[0]), "small_psnr_level0": psnr(*sml[0]), "weather_psnr": psnr(w0, wq)}
    return (np.concatenate([q.reshape(-1) for _, q in big]), np.concatenate([q.reshape(-1) for _, q in sml]), np.ascontiguousarray(wq)), stats


_CACHE = {}


def load_default_noise(seed=SHAPE_SEED):
    """(large 128^3 RGBA8, small 32^3 RGB8, weather 512^2 RGB8): the benchmark inputs."""
    if seed not in _CACHE:
        weather = load_bmp_rgb8(os.path.join(ASSET_DIR, "weather.bmp"))
        small = strip_to_volume(load_bmp_rgb8(os.path.join(ASSET_DIR, "worlnoise.bmp")), 32)
        large = generate_shape_noise(seed, 128)
        _CACHE[seed] = (large, small, weather)
    return _CACHE[seed]


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
