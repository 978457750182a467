"""The BC7 encoder (csrc/bc7enc_core.h: the per-block code of bc7enc.hip, here compiled for the host by tests/hostsim): SURVEY 8f row 4,
"BC7 encode/decode".  The reference's inputs are imported with compress/mode=2, high_quality=true (weather.bmp.import:19-20,
worlnoise.bmp.import:19-20, perlworlnoise.tga.import:19-20).  What can be pinned: every emitted block is a VALID BC7 block that an independent
decoder (Pillow) and csky_decode_bc7 expand to the same texels, and those texels are close to the input.  What cannot: agreement with the
engine's own encoder."""
import ctypes as C

import numpy as np
import pytest


def _enc(hostsim, img):
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 3:
        img = img[None]
    n, h, w = img.shape[:3]
    out = np.zeros((n, (h + 3) // 4, (w + 3) // 4, 16), np.uint8)
    hostsim.hostsim_bc7_encode(img.ctypes.data_as(C.c_void_p), w, h, n, out.ctypes.data_as(C.c_void_p))
    return out


def _psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    m = float((d * d).mean())
    return 99.0 if m == 0 else 10.0 * np.log10(255.0 * 255.0 / m)


def _modes(blocks):
    first = blocks.reshape(-1, 16)[:, 0].astype(np.int32)
    assert (first != 0).all()                                               # a mode bit in the first byte: never the reserved encoding
    return np.bincount(np.log2(first & -first).astype(np.int64), minlength=8)


def _opaque(rgb):
    return np.concatenate([rgb, np.full(rgb.shape[:-1] + (1,), 255, np.uint8)], -1)


def test_encoder_tables_are_the_decoders(hostsim):
    """the encoder's compact partition masks / anchors (bc7enc_core.h) against bc7_tables.h (tools/derive_bc7_tables.py)"""
    assert hostsim.hostsim_bc7_table_mismatches() == 0


def test_blocks_decode_the_same_everywhere_and_close_to_the_input(pkg, hostsim, noise):
    """weather map, shape slices (RGBA with an independent fourth channel), detail slices: Pillow's decoder and csky_decode_bc7 agree on every
    block, the round trip stays within what BC7 can do for such data, the modes used are the ones the block class calls for, and all eight occur."""
    Image = pytest.importorskip("PIL.Image")
    large, small, weather = noise
    cases = [("weather", _opaque(weather[128:256, 64:320])[None], 48.0, {0, 1, 2, 3, 6}),
             ("shape", large[40:42], 33.5, {4, 5, 6, 7}),
             ("detail", _opaque(small[:6]), 33.5, {0, 1, 2, 3, 6})]
    seen = set()
    for name, img, floor, allowed in cases:
        n, h, w = img.shape[:3]
        blocks = _enc(hostsim, img)
        used = _modes(blocks)
        assert set(np.nonzero(used)[0]) <= allowed, (name, used)
        seen |= set(int(m) for m in np.nonzero(used)[0])
        for i in range(n):
            mine = pkg.assets.decode_bc7(blocks[i], w, h)
            theirs = np.asarray(Image.frombytes("RGBA", (w, h), blocks[i].tobytes(), "bcn", (7,)))
            assert (mine == theirs).all(), (name, i)
        dec = np.stack([pkg.assets.decode_bc7(blocks[i], w, h) for i in range(n)])
        p = _psnr(dec[..., :3], img[..., :3]) if name != "shape" else _psnr(dec, img)
        assert p >= floor, (name, p)
        if name != "shape":
            assert (dec[..., 3] >= 254).all()                                # (mode 6 carries alpha with a p-bit: 254 or 255)
    assert seen == set(range(8)), seen                                       # every one of the eight modes was emitted (and cross-decoded) somewhere


def test_easy_blocks_are_near_exact(pkg, hostsim):
    rng = np.random.default_rng(5)
    # solid colours, any alpha: every channel within one level
    solid = np.zeros((64, 4, 4, 4), np.uint8)
    solid[:] = rng.integers(0, 256, size=(64, 1, 1, 4), dtype=np.uint8)
    b = _enc(hostsim, solid)
    for i in range(64):
        assert np.abs(pkg.assets.decode_bc7(b[i], 4, 4).astype(int) - solid[i]).max() <= 1
    # a smooth two-channel ramp: well inside one subset's reach
    g = np.zeros((16, 16, 4), np.uint8)
    g[..., 0] = np.arange(16)[None, :] * 16; g[..., 1] = np.arange(16)[:, None] * 8; g[..., 2] = 77; g[..., 3] = 255
    d = pkg.assets.decode_bc7(_enc(hostsim, g)[0], 16, 16)
    assert _psnr(d, g) >= 40.0 and np.abs(d.astype(int) - g).max() <= 5
    # two regions, each with its own pair of colours (four colours, not on one line), split along a partition shape (left / right halves of a
    # block = partition 0): only a multi-subset mode can do that well, and it must find the shape
    e = np.zeros((4, 8, 4), np.uint8)
    for x in range(8):
        for y in range(4):
            left = (x % 4) < 2
            a, b2 = ((200, 30, 40), (160, 90, 40)) if left else ((20, 180, 220), (70, 200, 150))
            e[y, x, :3] = a if (x + y) % 2 == 0 else b2
    e[..., 3] = 255
    blocks = _enc(hostsim, e)
    d = pkg.assets.decode_bc7(blocks[0], 8, 4)
    assert np.abs(d[..., :3].astype(int) - e[..., :3]).max() <= 3 and _modes(blocks)[[0, 1, 2, 3]].sum() == 2, (_modes(blocks), np.abs(d.astype(int) - e).max())
    # an alpha ramp over a flat colour: the scalar channel of mode 5 (or mode 6's fourth component) carries it
    a = np.zeros((4, 8, 4), np.uint8)
    a[...] = (90, 90, 200, 0); a[..., 3] = (np.arange(8) * 36)[None, :]
    d = pkg.assets.decode_bc7(_enc(hostsim, a)[0], 8, 4)
    assert np.abs(d.astype(int) - a).max() <= 8 and np.abs(d[..., :3].astype(int) - a[..., :3]).max() <= 2


def test_ragged_sizes_pad_with_the_edge_texels(pkg, hostsim):
    """a 10 x 7 image is 3 x 2 blocks; the texels outside repeat the last column / row (so they pull no end point away from the image)"""
    yy, xx = np.mgrid[0:7, 0:10]
    t = xx + yy                                                             # one direction in colour space: within a single subset's reach
    img = np.stack([10 + 9 * t, 200 - 7 * t, 3 + 5 * t, np.full_like(xx, 255)], -1).astype(np.uint8)
    b = _enc(hostsim, img)
    assert b.shape == (1, 2, 3, 16)
    d = pkg.assets.decode_bc7(b[0], 10, 7)
    full = pkg.assets.decode_bc7(b[0], 12, 8)
    assert (d == full[:7, :10]).all() and _psnr(d[..., :3], img[..., :3]) >= 38.0
    assert np.abs(full[:7, 10:, :3].astype(int) - full[:7, 9:10, :3].astype(int)).max() <= 6 and np.abs(full[7, :10, :3].astype(int) - full[6, :10, :3].astype(int)).max() <= 6


def test_no_block_is_off_by_its_own_range(pkg, hostsim, noise):
    """Block by block: the largest error of a decoded block stays well inside the block's own value range (measured: at most 0.55 of it on these
    inputs).  A wrong anchor bit, swapped end points or a misplaced index field would put whole blocks off by about their range, which an average
    (PSNR) over thousands of good blocks can hide."""
    large, small, weather = noise
    for name, img in (("weather", _opaque(weather[:256, :256])[None]), ("shape", large[60:64]), ("detail", _opaque(small[8:20]))):
        img = np.ascontiguousarray(img)
        n, h, w = img.shape[:3]
        blocks = _enc(hostsim, img)
        dec = np.stack([pkg.assets.decode_bc7(blocks[i], w, h) for i in range(n)]).astype(np.int32)
        tiles = lambda a: a.reshape(n, h // 4, 4, w // 4, 4, 4)
        err = np.abs(tiles(dec) - tiles(img.astype(np.int32))).max(axis=(2, 4, 5))
        span = (tiles(img).max(axis=(2, 4)).astype(np.int32) - tiles(img).min(axis=(2, 4))).max(-1)
        worst = (err / np.maximum(span, 8)).max()
        assert worst <= 0.75, (name, float(worst), int(err.max()))


def test_quality_one_is_never_worse_block_by_block_and_still_decodes_everywhere(pkg, hostsim, noise):
    """The sensitivity study's second encoder (bc7_encode_block quality 1: eight partitions fitted in full, coordinate descent on the stored end points):
    it only ever ACCEPTS a move that lowers a subset's squared error, so no block may come out worse than at quality 0; Pillow's decoder and
    csky_decode_bc7 agree on its blocks too; and on the shape volume's slices (four independent channels, the hard case) it gains measurably."""
    Image = pytest.importorskip("PIL.Image")
    large, small, weather = noise
    img = np.ascontiguousarray(large[40:44, :64, :64])                       # four 64 x 64 RGBA slices of the shape volume
    n, h, w = img.shape[:3]

    def enc(q):
        out = np.zeros((n, h // 4, w // 4, 16), np.uint8)
        hostsim.hostsim_bc7_encode_quality(img.ctypes.data_as(C.c_void_p), w, h, n, q, out.ctypes.data_as(C.c_void_p))
        return out

    b0, b1 = enc(0), enc(1)
    d0 = np.stack([pkg.assets.decode_bc7(b0[i], w, h) for i in range(n)]).astype(np.int64)
    d1 = np.stack([pkg.assets.decode_bc7(b1[i], w, h) for i in range(n)]).astype(np.int64)
    for i in range(n):
        assert (np.asarray(Image.frombytes("RGBA", (w, h), b1[i].tobytes(), "bcn", (7,))) == d1[i]).all()

    def block_err(d):
        e = ((d - img.astype(np.int64)) ** 2).sum(-1)                        # [n, h, w]
        return e.reshape(n, h // 4, 4, w // 4, 4).sum((2, 4))

    e0, e1 = block_err(d0), block_err(d1)
    assert (e1 <= e0).all(), int((e1 > e0).sum())
    assert _psnr(d1, img) > _psnr(d0, img) + 0.2, (_psnr(d0, img), _psnr(d1, img))
