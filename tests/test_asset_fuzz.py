"""The host-side asset entry points of libcloudsky.so under hostile input (no GPU needed): NULL and missing paths, truncated and bit-flipped BMP / TGA / .ctex /
.ctex3d files, absurd headers, out-of-range levels and sizes.  Contract (include/cloudsky.h): an error code and a text, never a crash, an exception or a
hang across the ABI.  Every case runs in a child process so that a SIGSEGV / SIGFPE / abort() is a test failure, not the end of the test session."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, os, struct, sys
    import numpy as np
    sys.path.insert(0, %r)
    import gvcd_amd
    L = gvcd_amd.lib()
    tmp = sys.argv[1]
    rng = np.random.default_rng(int(sys.argv[2]))
    w, h, d, lv = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    buf = np.zeros(1 << 22, np.uint8)
    bp = buf.ctypes.data_as(C.c_void_p)
    bad = 0
    def rc_ok(rc):
        return rc in (0, -1, -2, -3, -4, -5)
    # ---- NULL / missing paths
    for fn in (L.csky_load_bmp_rgb8, L.csky_load_tga_rgba8):
        assert fn(None, C.byref(w), C.byref(h), None, 0) < 0
        assert fn(b"/nonexistent/x", C.byref(w), C.byref(h), None, 0) < 0
    assert L.csky_load_ctex(None, C.byref(w), C.byref(h), C.byref(lv), None, 0) < 0
    assert L.csky_load_ctex3d(None, C.byref(w), C.byref(h), C.byref(d), C.byref(lv), None, 0) < 0
    # ---- levels / sizes that used to shift by >= 32 or overflow
    for lvl in (-5, 0, 7, 31, 32, 33, 64, 1 << 20, -(1 << 31)):
        L.csky_mip_offset(128, lvl, 4); L.csky_mip_offset(0, lvl, 4); L.csky_mip_offset(-8, lvl, 0)
    for levels in (0, 33, 64, 1 << 30, -1):
        assert L.csky_build_mips(bp, 8, 1, levels) < 0
    assert L.csky_build_mips(bp, 8, 3, 4) == 0
    for n in (0, 7, 12, -8, 1 << 20, 1 << 30):
        assert L.csky_generate_shape_noise(1, n, bp) < 0 and L.csky_generate_detail_noise(1, n, bp) < 0
    assert L.csky_strip_to_volume(None, 4, 3, bp) < 0 and L.csky_strip_to_volume(bp, 0, 3, bp) < 0
    assert L.csky_decode_bc7(None, 4, 4, bp) < 0 and L.csky_decode_bc7(bp, 0, 4, bp) < 0
    # ---- valid files of every container, then truncated / bit-flipped / header-mangled copies
    def bmp(W, H, bottom_up=True):
        stride = (W * 3 + 3) & ~3
        px = rng.integers(0, 256, stride * H, dtype=np.uint8).tobytes()
        hd = b"BM" + struct.pack("<IHHI", 54 + len(px), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, W, H if bottom_up else -H, 1, 24, 0, len(px), 2835, 2835, 0, 0)
        return hd + px
    def tga(W, H, rle):
        hd = bytes([0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, W & 255, W >> 8, H & 255, H >> 8, 32, 0x20])
        if not rle:
            return hd + rng.integers(0, 256, W * H * 4, dtype=np.uint8).tobytes()
        out, left = bytearray(), W * H
        while left:
            run = int(min(left, rng.integers(1, 129)))
            if rng.integers(0, 2):
                out += bytes([0x80 | (run - 1)]) + rng.integers(0, 256, 4, dtype=np.uint8).tobytes()
            else:
                out += bytes([run - 1]) + rng.integers(0, 256, 4 * run, dtype=np.uint8).tobytes()
            left -= run
        return hd + bytes(out)
    files = {"a.bmp": bmp(37, 11), "b.bmp": bmp(8, 8, False), "a.tga": tga(19, 7, False), "b.tga": tga(33, 9, True)}
    # .ctex / .ctex3d written by the suite's own writer (tests/test_godot_import.py) when available
    FMT_RGBA8, FMT_BPTC = 5, 22                                   # the container layout documented in include/cloudsky.h / godot_import.cpp
    def record(W, H, fmt, levels):
        return struct.pack("<IHHII", 0, W, H, len(levels) - 1, fmt) + b"".join(levels)
    lv2 = [rng.integers(0, 256, ((max(1, 16 >> l) + 3) // 4) * ((max(1, 8 >> l) + 3) // 4) * 16, dtype=np.uint8).tobytes() for l in range(3)]
    files["c.ctex"] = b"GST2" + struct.pack("<IIIIIIII", 1, 0, 0, 0, 0, 0, 0, 0) + record(16, 8, FMT_BPTC, lv2)
    sl0 = [record(8, 8, FMT_RGBA8, [rng.integers(0, 256, 8 * 8 * 4, dtype=np.uint8).tobytes()]) for _ in range(4)]
    sl1 = [record(4, 4, FMT_RGBA8, [rng.integers(0, 256, 4 * 4 * 4, dtype=np.uint8).tobytes()]) for _ in range(2)]
    sl2 = [record(2, 2, FMT_RGBA8, [rng.integers(0, 256, 2 * 2 * 4, dtype=np.uint8).tobytes()])]
    files["d.ctex3d"] = b"GSTL" + struct.pack("<IIIIIII", 1, 4, 2, 0, 3, 0, 0) + b"".join(sl0 + sl1 + sl2)
    def load(kind, path):
        p = path.encode()
        if kind == "bmp":
            r1 = L.csky_load_bmp_rgb8(p, C.byref(w), C.byref(h), None, 0)
            cap = min(buf.size, max(0, w.value) * max(0, h.value) * 3) if r1 == 0 else 0
            return r1, L.csky_load_bmp_rgb8(p, C.byref(w), C.byref(h), bp, C.c_size_t(cap))
        if kind == "tga":
            r1 = L.csky_load_tga_rgba8(p, C.byref(w), C.byref(h), None, 0)
            cap = min(buf.size, max(0, w.value) * max(0, h.value) * 4) if r1 == 0 else 0
            return r1, L.csky_load_tga_rgba8(p, C.byref(w), C.byref(h), bp, C.c_size_t(cap))
        if kind == "ctex":
            return L.csky_load_ctex(p, C.byref(w), C.byref(h), C.byref(lv), None, 0), L.csky_load_ctex(p, C.byref(w), C.byref(h), C.byref(lv), bp, C.c_size_t(buf.size))
        return (L.csky_load_ctex3d(p, C.byref(w), C.byref(h), C.byref(d), C.byref(lv), None, 0),
                L.csky_load_ctex3d(p, C.byref(w), C.byref(h), C.byref(d), C.byref(lv), bp, C.c_size_t(buf.size)))
    n_cases = 0
    for name, data in files.items():
        kind = name.split(".")[-1].replace("ctex3d", "ctex3d")
        path = os.path.join(tmp, "f_" + name)
        open(path, "wb").write(data)
        r = load(kind, path)
        assert r == (0, 0), (name, r, L.csky_assets_last_error())
        for trial in range(int(sys.argv[3])):
            b = bytearray(data)
            mode = trial %% 4
            if mode == 0:
                b = b[: int(rng.integers(0, len(b)))]                                   # truncated
            elif mode == 1:
                for _ in range(int(rng.integers(1, 6))):                                  # bit flips anywhere
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2:
                for _ in range(int(rng.integers(1, 4))):                                  # header bytes replaced by extremes
                    b[int(rng.integers(0, min(len(b), 64)))] = int(rng.choice([0, 1, 0x7f, 0x80, 0xff]))
            else:
                k = int(rng.integers(0, min(len(b), 60)))                                 # a 32-bit field set to a huge / negative value
                b[k:k + 4] = struct.pack("<i", int(rng.choice([-(1 << 31), -1, (1 << 31) - 1, 1 << 30, 65536, 0])))
            open(path, "wb").write(bytes(b))
            r = load(kind, path)
            assert rc_ok(r[0]) and rc_ok(r[1]), (name, trial, r)
            n_cases += 1
    print("fuzz cases survived:", n_cases)
''') % (ROOT,)


def test_asset_entry_points_survive_hostile_input(pkg, tmp_path):
    for seed in (1, 2):
        r = subprocess.run([sys.executable, "-c", CHILD, str(tmp_path), str(seed), "150"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
        assert "fuzz cases survived" in r.stdout


NULL_CHILD = textwrap.dedent('''
    import ctypes as C, sys
    sys.path.insert(0, %r)
    import gvcd_amd
    from gvcd_amd import _lib
    L = gvcd_amd.lib()
    n = 0
    for name, res, args in _lib.SYMBOLS:
        fn = getattr(L, name)
        zeros = []
        for a in args:
            if a in (C.c_void_p, C.c_char_p) or (isinstance(a, type) and issubclass(a, C._Pointer)):
                zeros.append(None)
            elif a in (C.c_float, C.c_double):
                zeros.append(0.0)
            else:
                zeros.append(0)
        r = fn(*zeros)
        if res is C.c_int and name not in ("csky_abi_version", "csky_device_count", "csky_variant_count", "csky_multi_device_count"):
            assert r <= 0, (name, r)            # an error code (or, for the few calls that have nothing to do, OK): never a positive surprise
        n += 1
    print("entry points called with all-zero arguments:", n)
''') % (ROOT,)


def test_every_entry_point_survives_all_zero_arguments(pkg):
    """Every symbol of both headers called with NULL pointers and zero scalars (no context, no GPU needed): an error code or a no-op, never a crash."""
    r = subprocess.run([sys.executable, "-c", NULL_CHILD], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert "entry points called with all-zero arguments" in r.stdout
