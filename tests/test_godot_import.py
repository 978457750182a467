"""BC7 decoding and the Godot import containers (csrc/godot_import.cpp): SURVEY §8 (f)-4, the reference's *.import files say
compress/mode=2, compress/high_quality=true => the samplers read decoded BPTC blocks."""
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_bc7_decoder_matches_the_committed_vectors_of_an_independent_decoder(pkg):
    """96 random blocks of each of the eight modes (partitions, rotations, index selection, p-bits, anchors all exercised), expected
    pixels from Pillow (tests/golden/make_bc7_vectors.py).  Bit-exact: BC7 decoding is integer arithmetic fixed by the format."""
    v = np.load(os.path.join(HERE, "golden", "bc7_vectors.npz"))
    for m in range(8):
        blocks, want = v["blocks"][m], v["pixels"][m]
        assert ((blocks[:, 0] >> m) & 1).all() and not (blocks[:, 0] & ((1 << m) - 1)).any()
        n = blocks.shape[0]
        got = pkg.assets.decode_bc7(blocks.reshape(-1), 4 * n, 4)          # n blocks side by side = a 4n x 4 image
        got = got.reshape(4, n, 4, 4).transpose(1, 0, 2, 3).reshape(n, 16, 4)
        bad = (got != want).any(axis=(1, 2))
        assert not bad.any(), (m, int(bad.sum()), blocks[bad][0].tolist())


def test_bc7_decoder_against_pillow_live(pkg):
    """A larger live sample when Pillow is importable (it is in this image): 4000 random blocks per mode, plus purely random blocks
    (mode distribution 1/2, 1/4, ...; first byte 0 = the reserved encoding -> all zero)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    n = 4000
    for m in list(range(8)) + [None]:
        blocks = rng.integers(0, 256, size=(n, 16), dtype=np.uint8)
        if m is not None:
            blocks[:, 0] = (blocks[:, 0] & ~np.uint8((1 << (m + 1)) - 1)) | np.uint8(1 << m)
        else:
            blocks[:8, 0] = 0
        want = np.asarray(Image.frombytes("RGBA", (4 * n, 4), blocks.tobytes(), "bcn", (7,))).reshape(4, n, 4, 4)
        got = pkg.assets.decode_bc7(blocks.reshape(-1), 4 * n, 4).reshape(4, n, 4, 4)
        reserved = blocks[:, 0] == 0                                         # no mode bit: the format reserves it and specifies an all-zero result
        assert (got[:, ~reserved] == want[:, ~reserved]).all(), (m, int((got != want).any(-1).sum()))
        assert not got[:, reserved].any() and (m is not None or reserved.sum() >= 8)   # (Pillow returns opaque black there; no encoder emits it)


def test_bc7_image_sizes_that_are_not_multiples_of_four(pkg):
    rng = np.random.default_rng(3)
    w, h = 10, 7                                                            # 3 x 2 blocks, the last column / row partly outside
    blocks = rng.integers(0, 256, size=(6, 16), dtype=np.uint8)
    full = pkg.assets.decode_bc7(blocks.reshape(-1), 12, 8)
    part = pkg.assets.decode_bc7(blocks.reshape(-1), w, h)
    assert part.shape == (h, w, 4) and (part == full[:h, :w]).all()
    with pytest.raises(ValueError):
        pkg.assets.decode_bc7(blocks.reshape(-1)[:-1], 12, 8)


# ---- container writers (the layouts documented in include/cloudsky.h / godot_import.cpp) ------------------------------------------
FMT_R8, FMT_RGB8, FMT_RGBA8, FMT_BPTC = 2, 4, 5, 22


def image_record(w, h, fmt, levels_bytes, data_format=0):
    return struct.pack("<IHHII", data_format, w, h, len(levels_bytes) - 1, fmt) + b"".join(levels_bytes)


def write_ctex(path, record, version=1, magic=b"GST2"):
    with open(path, "wb") as f:
        f.write(magic + struct.pack("<IIIIIIII", version, 0, 0, 0, 0, 0, 0, 0) + record)


def write_ctex3d(path, depth, records, n_mip_images, version=1, magic=b"GSTL"):
    with open(path, "wb") as f:
        f.write(magic + struct.pack("<IIIIIII", version, depth, 2, 0, n_mip_images, 0, 0) + b"".join(records))


def test_ctex_2d_bptc_and_raw(pkg, tmp_path):
    rng = np.random.default_rng(11)
    w, h = 16, 8
    lv = [rng.integers(0, 256, size=((max(1, w >> l) + 3) // 4) * ((max(1, h >> l) + 3) // 4) * 16, dtype=np.uint8) for l in range(3)]
    p = str(tmp_path / "weather.bmp-0123.bptc.ctex")
    write_ctex(p, image_record(w, h, FMT_BPTC, [b.tobytes() for b in lv]))
    got = pkg.assets.load_ctex(p)
    assert [g.shape for g in got] == [(8, 16, 4), (4, 8, 4), (2, 4, 4)]
    for l, g in enumerate(got):
        assert (g == pkg.assets.decode_bc7(lv[l], max(1, w >> l), max(1, h >> l))).all()
    raw = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)              # RGB8, no mips, odd size
    p2 = str(tmp_path / "raw.ctex")
    write_ctex(p2, image_record(7, 5, FMT_RGB8, [raw.tobytes()]))
    (g,) = pkg.assets.load_ctex(p2)
    assert (g[..., :3] == raw).all() and (g[..., 3] == 255).all()
    for bad, msg in ((dict(magic=b"GSTL"), "GST2"), (dict(version=9), "version")):
        write_ctex(p2, image_record(7, 5, FMT_RGB8, [raw.tobytes()]), **bad)
        with pytest.raises(pkg.CloudSkyError, match=msg):
            pkg.assets.load_ctex(p2)
    write_ctex(p2, image_record(7, 5, FMT_RGB8, [raw.tobytes()], data_format=1))       # PNG payload: refused, not misread
    with pytest.raises(pkg.CloudSkyError, match="not supported"):
        pkg.assets.load_ctex(p2)
    write_ctex(p2, image_record(7, 5, FMT_RGB8, [raw.tobytes()[:-1]]))
    with pytest.raises(pkg.CloudSkyError, match="truncated"):
        pkg.assets.load_ctex(p2)
    with pytest.raises(pkg.CloudSkyError, match="cannot open"):
        pkg.assets.load_ctex(str(tmp_path / "missing.ctex"))


def test_ctex3d_volume_with_importer_mips_feeds_set_noise_mips_layout(pkg, tmp_path):
    """A 32^3 volume with its 5 mip levels as BPTC slices (the shape of .godot/imported/worlnoise.bmp-*.bptc.ctex3d) and a 128^3 raw one:
    the loader returns [d, h, w, 4] per level, and chains_from_godot_import lays them out as csky_set_noise_mips wants them."""
    rng = np.random.default_rng(5)

    def volume_file(path, n, fmt):
        recs, expect, nm = [], [], 0
        for l in range(n.bit_length()):
            m = n >> l
            lvl = []
            for z in range(m):
                if fmt == FMT_BPTC:
                    b = rng.integers(0, 256, size=((m + 3) // 4) ** 2 * 16, dtype=np.uint8)
                    recs.append(image_record(m, m, FMT_BPTC, [b.tobytes()]))
                    lvl.append(pkg.assets.decode_bc7(b, m, m))
                else:
                    px = rng.integers(0, 256, size=(m, m, 4), dtype=np.uint8)
                    recs.append(image_record(m, m, FMT_RGBA8, [px.tobytes()]))
                    lvl.append(px)
                nm += l > 0
            expect.append(np.stack(lvl))
        write_ctex3d(path, n, recs, nm)
        return expect

    ps, pl, pw = str(tmp_path / "small.ctex3d"), str(tmp_path / "large.ctex3d"), str(tmp_path / "weather.ctex")
    es = volume_file(ps, 32, FMT_BPTC)
    el = volume_file(pl, 128, FMT_RGBA8)
    wb = rng.integers(0, 256, size=(128 * 128 * 16), dtype=np.uint8)
    write_ctex(pw, image_record(512, 512, FMT_BPTC, [wb.tobytes()]))
    got = pkg.assets.load_ctex3d(ps)
    assert len(got) == 6 and all((g == e).all() for g, e in zip(got, es))
    large, small, weather = pkg.assets.chains_from_godot_import(pl, ps, pw)
    L = pkg.lib()
    assert large.size == L.csky_mip_offset(128, 8, 4) and small.size == L.csky_mip_offset(32, 6, 3) and weather.shape == (512, 512, 3)
    for l in range(8):
        o, m = L.csky_mip_offset(128, l, 4), 128 >> l
        assert (large[o:o + m ** 3 * 4].reshape(m, m, m, 4) == el[l]).all()
    for l in range(6):
        o, m = L.csky_mip_offset(32, l, 3), 32 >> l
        assert (small[o:o + m ** 3 * 3].reshape(m, m, m, 3) == es[l][..., :3]).all()
    assert (weather == pkg.assets.decode_bc7(wb, 512, 512)[..., :3]).all()
    # a file whose last mip level is short is refused
    recs = [image_record(2, 2, FMT_RGBA8, [bytes(16)]) for _ in range(2)]
    write_ctex3d(ps, 2, recs + [image_record(1, 1, FMT_RGBA8, [bytes(4)])], 1)
    assert [g.shape for g in pkg.assets.load_ctex3d(ps)] == [(2, 2, 2, 4), (1, 1, 1, 4)]
    write_ctex3d(ps, 4, [image_record(4, 4, FMT_RGBA8, [bytes(64)]) for _ in range(4)] + [image_record(2, 2, FMT_RGBA8, [bytes(16)])], 1)
    with pytest.raises(pkg.CloudSkyError, match="incomplete"):
        pkg.assets.load_ctex3d(ps)


@pytest.mark.gpu
def test_set_noise_mips_uses_the_callers_chains(pkg, noise, gpu_ctx):
    """csky_set_noise_mips with the library's own box-filter chains builds byte-identical device layouts to csky_set_noise; with one
    texel of a mip level changed the layouts differ exactly there (the chain is used as given, not rebuilt)."""
    large, small, weather = noise
    lc, sc = pkg.assets.build_mips(large, 8), pkg.assets.build_mips(small, 6)
    ctx = pkg.Context(0)
    try:
        ctx.set_noise(large, small, weather)
        ref = [ctx.read_baked_texture(k).copy() for k in range(5)]
        ctx.set_noise_mips(lc, sc, weather)
        for k in range(5):
            assert (ctx.read_baked_texture(k) == ref[k]).all(), k
        sc2 = sc.copy()
        o = pkg.lib().csky_mip_offset(32, 2, 3)                             # level 2 (8^3), texel (0, 0, 0), channel 0
        sc2[o] = np.uint8(int(sc2[o]) ^ 0x55)
        ctx.set_noise_mips(lc, sc2, weather)
        assert (ctx.read_baked_texture(4) == sc2).all()
        assert (ctx.read_baked_texture(0) == ref[0]).all() and not (ctx.read_baked_texture(1) == ref[1]).all()
        with pytest.raises(ValueError):
            ctx.set_noise_mips(large, small, weather)                        # level-0 arrays are not chains
    finally:
        ctx.close()


def test_import_entry_points_refuse_bad_arguments_without_a_gpu(pkg, tmp_path):
    """Error behaviour of the new C-ABI entry points (no compute): codes, not crashes; texts through csky_assets_last_error."""
    import ctypes as C
    L = pkg.lib()
    buf = (C.c_uint8 * 64)()
    assert L.csky_decode_bc7(None, 4, 4, buf) == -1 and L.csky_decode_bc7(buf, 0, 4, buf) == -1 and L.csky_decode_bc7(buf, 4, 4, None) == -1
    assert b"decode_bc7" in L.csky_assets_last_error()
    w, h, d, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.csky_load_ctex(None, C.byref(w), C.byref(h), C.byref(n), None, 0) == -4
    assert L.csky_load_ctex3d(b"/nonexistent/x.ctex3d", C.byref(w), C.byref(h), C.byref(d), C.byref(n), None, 0) == -4
    p = str(tmp_path / "huge.ctex")                                          # absurd dimensions / mip counts are refused before any allocation
    write_ctex(p, struct.pack("<IHHII", 0, 65535, 65535, 0, FMT_RGBA8))
    assert L.csky_load_ctex(p.encode(), C.byref(w), C.byref(h), C.byref(n), None, 0) == -4 and b"unsupported image" in L.csky_assets_last_error()
    write_ctex(p, struct.pack("<IHHII", 0, 8, 8, 40, FMT_RGBA8))
    assert L.csky_load_ctex(p.encode(), C.byref(w), C.byref(h), C.byref(n), None, 0) == -4
    write_ctex3d(p, 0, [], 0)
    assert L.csky_load_ctex3d(p.encode(), C.byref(w), C.byref(h), C.byref(d), C.byref(n), None, 0) == -4 and b"bad header" in L.csky_assets_last_error()
    small = np.zeros(16, np.uint8)                                          # caller buffer too small: refused, nothing written past it
    write_ctex(p, image_record(4, 4, FMT_RGBA8, [bytes(range(64))]))
    assert L.csky_load_ctex(p.encode(), C.byref(w), C.byref(h), C.byref(n), small.ctypes.data_as(C.c_void_p), small.nbytes) == -1
    assert not small.any() and (w.value, h.value, n.value) == (4, 4, 1)
    assert L.csky_set_noise_mips(None, buf, buf, buf) == -1                   # NULL context
    assert L.csky_multi_set_noise_mips(None, buf, buf, buf) == -1 and L.csky_multi_set_frames_in_flight(None, 2) == -1


def test_importer_shaped_files_with_real_bc7_payloads(pkg, hostsim, noise, tmp_path):
    """The readers above are fed random blocks; here the payload is what an importer would actually store: the detail volume and a weather crop,
    mip levels box-filtered, every slice BC7-encoded by the library's encoder (host twin of bc7enc.hip), wrapped in .ctex3d / .ctex containers.
    Reading them back gives texels close to the source and exactly the decoded blocks; chains_from_godot_import would hand them to set_noise_mips."""
    import ctypes as C
    _, small, weather = noise

    def enc(img4):
        img4 = np.ascontiguousarray(img4, np.uint8)
        n, h, w = img4.shape[:3]
        out = np.zeros((n, (h + 3) // 4, (w + 3) // 4, 16), np.uint8)
        hostsim.hostsim_bc7_encode(img4.ctypes.data_as(C.c_void_p), w, h, n, out.ctypes.data_as(C.c_void_p))
        return out

    chain = pkg.assets.build_mips(small, 6)
    recs, levels, nm, o = [], [], 0, 0
    for l in range(6):
        m = 32 >> l
        lv = chain[o:o + m ** 3 * 3].reshape(m, m, m, 3); o += m ** 3 * 3
        rgba = np.concatenate([lv, np.full((m, m, m, 1), 255, np.uint8)], -1)
        blocks = enc(rgba)
        for z in range(m):
            recs.append(image_record(m, m, FMT_BPTC, [blocks[z].tobytes()]))
            nm += l > 0
        levels.append((lv, blocks))
    p3 = str(tmp_path / "worlnoise.bmp-0.bptc.ctex3d")
    write_ctex3d(p3, 32, recs, nm)
    got = pkg.assets.load_ctex3d(p3)
    assert [g.shape for g in got] == [(32 >> l,) * 3 + (4,) for l in range(6)]
    for l, (lv, blocks) in enumerate(levels):
        m = 32 >> l
        assert all((got[l][z] == pkg.assets.decode_bc7(blocks[z], m, m)).all() for z in range(m))
        d = got[l][..., :3].astype(np.float64) - lv
        assert 10 * np.log10(255.0 ** 2 / max(1e-9, (d * d).mean())) >= (33.0 if l == 0 else 28.0), l   # (coarser levels vary faster per block; the 1- and 2-texel levels are one padded block each)
    crop = np.concatenate([weather[:64, :128], np.full((64, 128, 1), 255, np.uint8)], -1)
    pw = str(tmp_path / "weather.bmp-0.bptc.ctex")
    write_ctex(pw, image_record(128, 64, FMT_BPTC, [enc(crop[None])[0].tobytes()]))
    (w0,) = pkg.assets.load_ctex(pw)
    d = w0[..., :3].astype(np.float64) - crop[..., :3]
    assert w0.shape == (64, 128, 4) and 10 * np.log10(255.0 ** 2 / (d * d).mean()) >= 45.0
