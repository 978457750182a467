#!/usr/bin/env python
"""bc7_ideal_bound.py -- what NO BC7 encoder can beat on the shape volume: the error floor of the format's geometry, for the error bar of
tools/bc7_sensitivity.py (no independent BC7 encoder exists in this image: Pillow 12.2 writes BC1-3 / BC5 only).

A BC7 block approximates its 16 texels by points on ONE line segment per subset (1 subset: mode 6; 2 subsets in one of 64 fixed partitions:
modes 1 / 3 / 7; 3 subsets: modes 0 / 2, opaque blocks only), or by a colour line plus an independently indexed scalar channel (modes 4 / 5, four
channel rotations).  Drop every quantisation (real-valued end points, real-valued positions on the line): the residual of the best line through a
subset is its scatter minus the largest eigenvalue, the rotated scalar channel is reproduced exactly.  The minimum of that over the modes' shapes is
a LOWER bound of any encoder's squared error on the block; its PSNR over the volume is an UPPER bound of what compress/mode=2 can reach, whatever
the engine's encoder does.  Host only (numpy); prints the bound next to the library's two encoders when tests/hostsim is built."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gvcd_amd  # noqa: E402

# bit i of PART2[p] = subset of texel i in two-subset partition p (the BPTC format's table, as in csrc/bc7enc_core.h::bc7_part2_mask)
PART2 = [0xCCCC, 0x8888, 0xEEEE, 0xECC8, 0xC880, 0xFEEC, 0xFEC8, 0xEC80, 0xC800, 0xFFEC, 0xFE80, 0xE800, 0xFFE8, 0xFF00, 0xFFF0, 0xF000,
         0xF710, 0x008E, 0x7100, 0x08CE, 0x008C, 0x7310, 0x3100, 0x8CCE, 0x088C, 0x3110, 0x6666, 0x366C, 0x17E8, 0x0FF0, 0x718E, 0x399C,
         0xAAAA, 0xF0F0, 0x5A5A, 0x33CC, 0x3C3C, 0x55AA, 0x9696, 0xA55A, 0x73CE, 0x13C8, 0x324C, 0x3BDC, 0x6996, 0xC33C, 0x9966, 0x0660,
         0x0272, 0x04E4, 0x4E40, 0x2720, 0xC936, 0x936C, 0x39C6, 0x639C, 0x9336, 0x9CC6, 0x817E, 0xE718, 0xCCF0, 0x0FCC, 0x7744, 0xEE22]


def line_residual(x, mask=None):
    """x [B, 16, D]; mask [16] bool (texels of the subset) or None: summed squared distance to the best line through the subset, per block."""
    if mask is not None:
        x = x[:, mask]
    xc = x - x.mean(1, keepdims=True)
    s = np.einsum("bnd,bne->bde", xc, xc)
    return np.maximum(np.trace(s, axis1=1, axis2=2) - np.linalg.eigvalsh(s)[:, -1], 0.0)


def ideal_block_error(blocks):
    """blocks [B, 16, 4] float64 -> per block: the smallest residual over the shapes the format offers (see the module text)."""
    best = line_residual(blocks)                                                       # one subset, four channels (mode 6)
    for rot in range(4):                                                               # modes 4 / 5: channel `rot` on its own, exactly
        best = np.minimum(best, line_residual(blocks[:, :, [c for c in range(4) if c != rot]]))
    for p in range(64):                                                                # two subsets (mode 7; for opaque blocks 1 / 3)
        m = np.array([(PART2[p] >> i) & 1 for i in range(16)], bool)
        best = np.minimum(best, line_residual(blocks, m) + line_residual(blocks, ~m))
    return best


def line_projection(x, mask=None):
    """x [B, 16, D] -> (residual [B], reconstruction [B, 16, D]: every texel of the subset moved onto the best line through it; others untouched)"""
    sub = x if mask is None else x[:, mask]
    mean = sub.mean(1, keepdims=True)
    xc = sub - mean
    s = np.einsum("bnd,bne->bde", xc, xc)
    w, v = np.linalg.eigh(s)
    ax = v[:, :, -1]                                                                   # [B, D] unit principal axis
    t = np.einsum("bnd,bd->bn", xc, ax)
    rec_sub = mean + t[:, :, None] * ax[:, None, :]
    rec = x.copy()
    if mask is None:
        rec = rec_sub
    else:
        rec[:, mask] = rec_sub
    return np.maximum(np.trace(s, axis1=1, axis2=2) - w[:, -1], 0.0), rec


def ideal_reconstruction(blocks):
    """blocks [B, 16, 4] -> (error [B], texels [B, 16, 4]) of the best shape per block: the picture an encoder with unlimited precision would store."""
    best, rec = line_projection(blocks)
    for rot in range(4):
        keep = [c for c in range(4) if c != rot]
        e, r3 = line_projection(blocks[:, :, keep])
        r = blocks.copy(); r[:, :, keep] = r3
        take = e < best
        best = np.where(take, e, best); rec[take] = r[take]
    for p in range(64):
        m = np.array([(PART2[p] >> i) & 1 for i in range(16)], bool)
        e0, r0 = line_projection(blocks, m)
        e1, r1 = line_projection(r0, ~m)                                                # (r0 keeps the other subset's texels as they were)
        take = e0 + e1 < best
        best = np.where(take, e0 + e1, best); rec[take] = r1[take]
    return best, rec


def from_blocks(b, n, h, w):
    return b.reshape(n, h // 4, w // 4, 4, 4, 4).transpose(0, 1, 3, 2, 4, 5).reshape(n, h, w, 4)


def ideal_chain(large):
    """The 8-level box chain of the shape volume with every level passed through ideal_reconstruction slice by slice (levels below 4 x 4 texels are kept:
    a single block holds them exactly enough), rounded to 8 bits -> the flat chain csky_set_noise_mips takes, and the level-0 PSNR."""
    chain = gvcd_amd.assets.build_mips(large, 8)
    out, o, psnr0 = [], 0, None
    for l in range(8):
        m = 128 >> l
        lv = chain[o:o + m * m * m * 4].reshape(m, m, m, 4); o += m * m * m * 4
        if m >= 4:
            rec = np.concatenate([ideal_reconstruction(to_blocks(lv[i:i + 16]))[1] for i in range(0, m, 16)])
            q = np.clip(np.rint(from_blocks(rec, m, m, m)), 0, 255).astype(np.uint8)
        else:
            q = lv.copy()
        if l == 0:
            psnr0 = 10 * np.log10(255.0 ** 2 / ((q.astype(np.float64) - lv) ** 2).mean())
        out.append(q.reshape(-1))
    return np.concatenate(out), psnr0


def to_blocks(vol):
    """[n, h, w, 4] uint8 slices -> [n * h/4 * w/4, 16, 4] float64"""
    n, h, w, _ = vol.shape
    return vol.reshape(n, h // 4, 4, w // 4, 4, 4).transpose(0, 1, 3, 2, 4, 5).reshape(-1, 16, 4).astype(np.float64)


def main():
    large, _, _ = gvcd_amd.assets.load_default_noise()
    if len(sys.argv) > 2 and sys.argv[1] == "--write":                                 # the whole chain, for tools/bc7_sensitivity.py's third row (minutes of numpy: run it where no GPU is waiting)
        chain, psnr0 = ideal_chain(large)
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        np.savez_compressed(sys.argv[2], chain=chain, psnr_level0=psnr0, shape_sha256=gvcd_amd.assets.sha256(large))
        print("wrote %s: level-0 PSNR of the rounded ideal reconstruction %.2f dB" % (sys.argv[2], psnr0))
        return
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 4                                # every `step`-th slice of the 128 (4: 32 slices, 32 768 blocks)
    vol = np.ascontiguousarray(large[::step])
    err = ideal_block_error(to_blocks(vol))
    mse = err.sum() / vol.size
    print("shape volume, level 0, every %d-th slice (%d blocks): NO BC7 encoder can exceed %.2f dB (unquantised line fits over the format's subset shapes)" % (
        step, err.size, 10 * np.log10(255.0 ** 2 / mse)))
    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    if os.path.exists(so):
        hs = C.CDLL(so)
        n, h, w = vol.shape[:3]
        for q in (0, 1):
            out = np.zeros((n, h // 4, w // 4, 16), np.uint8)
            hs.hostsim_bc7_encode_quality(vol.ctypes.data_as(C.c_void_p), w, h, n, q, out.ctypes.data_as(C.c_void_p))
            dec = np.stack([gvcd_amd.assets.decode_bc7(out[i], w, h) for i in range(n)]).astype(np.float64)
            print("   this library's encoder, quality %d (the per-block code of bc7enc.hip on the host): %.2f dB" % (q, 10 * np.log10(255.0 ** 2 / ((dec - vol) ** 2).mean())))


if __name__ == "__main__":
    main()
