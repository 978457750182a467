"""N > 1 path on CPU: world_size-2/3 gloo processes shard the frame's bands, gather on rank 0, and must reproduce
the single-process frame bit for bit.  The per-band renderer here is the oracle (a test stand-in for the HIP
call); the sharding / gather / interleave code under test is the product's tiling.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gvcd_amd
    from oracle import oracle as O
    large, small, weather = gvcd_amd.assets.load_default_noise()
    tex = O.OracleTextures(large, small, weather)
    sun = np.array([1, 1, 0], np.float64) / np.sqrt(2)
    sk = O.sky_lut(sun.astype(np.float32), O.transmittance_lut())
    p = O.default_params(W, H, sun)

    def render_bands(bands, out):
        br, first, stride, n = bands
        for k in range(n):
            y0 = (first + k * stride) * br
            img = O.clouds(tex, p, sk, rect=(0, y0, W, br))
            out[k * br:(k + 1) * br] = torch.from_numpy(img.view(np.int16).copy())

    frame = gvcd_amd.tiling.render_sharded(render_bands, H, W, rank, world, dist, torch.device("cpu"))
    if rank == 0:
        full = O.clouds(tex, p, sk)
        q.put(bool((frame.numpy().view(np.uint16) == full.view(np.uint16)).all()))
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 32), (3, 40)])
def test_sharded_frame_equals_single_process(world, H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, 32, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_band_partition_is_exact():
    sys.path.insert(0, ROOT)
    import gvcd_amd
    T = gvcd_amd.tiling
    for H in (8, 64, 1024, 2048, 40):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                br, first, stride, n = T.bands_for_rank(H, r, world)
                assert n <= T.max_bands(H, world)
                seen += [first + k * stride for k in range(n)]
            assert sorted(seen) == list(range(H // 8))      # every band exactly once
    with pytest.raises(ValueError):
        T.bands_for_rank(30, 0, 2)


def _group_worker(rank, world, groups, port, H, W, frames, q):
    """tiling.FrameGroups (the dealing bench.py --groups uses): frame f goes to group f % groups, the group's ranks split its bands, rank 0
    receives every frame over that group's communicator.  Synthetic bands (value = a function of frame and pixel row), two gathers in flight.
    Behind its bands every rank sends its rows index::per of a (ragged: 10 rows) sky LUT, the byte layout bench.py uses at N > 1."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gvcd_amd
    T = gvcd_amd.tiling
    fg = T.FrameGroups(rank, world, groups, dist)
    br, first, stride, n = fg.bands(H)
    mb = fg.max_bands(H)
    nbuf = 2 * (groups if rank == 0 else 1)
    LH, LW = 10, 6
    band_bytes, lut_bytes = mb * br * W * 8, fg.max_lut_rows(LH) * LW * 8
    local_b = [torch.zeros(band_bytes + lut_bytes, dtype=torch.uint8) for _ in range(nbuf)]
    local = [t[:band_bytes].view(torch.int16).view(mb * br, W, 4) for t in local_b]
    local_lut = [t[band_bytes:].view(torch.int16).view(-1, LW, 4) for t in local_b]
    gathered = [torch.empty((fg.max_members, band_bytes + lut_bytes), dtype=torch.uint8) for _ in range(nbuf)] if rank == 0 else [None] * nbuf
    pending, ok, taken = [], True, 0
    r0, rs, rn = fg.lut_rows(LH)

    def lut_value(f, rows):
        return ((f * 11 + rows.view(-1, 1, 1) * 5 + torch.arange(LW, dtype=torch.int32).view(1, LW, 1)) % 30000).to(torch.int16).expand(rows.numel(), LW, 4)

    def check(f, b):
        img, lut = fg.split(gathered[b], H, W, LH, LW)
        frame = fg.assemble(f, img, H)
        rows = torch.arange(H, dtype=torch.int32).view(H, 1, 1)
        want = ((f * 37 + rows * 3) % 30000).to(torch.int16).expand(H, W, 4)
        sky = fg.assemble_lut(f, lut, LH)
        return bool((frame == want).all()) and tuple(frame.shape) == (H, W, 4) and tuple(sky.shape) == (LH, LW, 4) and \
            bool((sky == lut_value(f, torch.arange(LH, dtype=torch.int32))).all())

    for f in range(frames):
        if not fg.takes_part(f):
            continue
        b = taken % nbuf
        taken += 1
        if fg.renders(f):
            for k in range(n):
                y0 = (first + k * stride) * br
                rows = torch.arange(y0, y0 + br, dtype=torch.int32).view(br, 1, 1)
                local[b][k * br:(k + 1) * br] = ((f * 37 + rows * 3) % 30000).to(torch.int16).expand(br, W, 4)
            local_lut[b][:rn] = lut_value(f, torch.arange(r0, LH, rs, dtype=torch.int32))
        else:
            local_b[b].fill_(255)                     # rank 0's dummy for another group's frame: must never show up
        pending.append((fg.gather(f, local_b[b], gathered[b], async_op=True), f, b))
        if len(pending) == nbuf:                      # the oldest gather ran while the younger frames were produced
            w, f0, b0 = pending.pop(0)
            w.wait()
            if rank == 0:
                ok = ok and check(f0, b0)
    for w, f0, b0 in pending:
        w.wait()
        if rank == 0:
            ok = ok and check(f0, b0)
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,groups,H", [(4, 2, 64), (3, 3, 40), (2, 1, 32), (4, 4, 16)])
def test_frame_groups_deliver_every_frame_to_rank_0(world, groups, H):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, groups, port, H, 24, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_frame_groups_partition():
    sys.path.insert(0, ROOT)
    import gvcd_amd
    T = gvcd_amd.tiling
    for world, G in ((8, 1), (8, 2), (8, 4), (8, 8), (6, 3)):
        seen = {}
        for r in range(world):
            fg = T.FrameGroups(r, world, G)
            assert fg.per == world // G and fg.max_members == (fg.per + 1 if G > 1 else fg.per)
            for f in range(2 * G):
                if fg.renders(f):
                    br, first, stride, n = fg.bands(64)
                    seen.setdefault(f, []).extend(first + k * stride for k in range(n))
                assert fg.takes_part(f) == (fg.renders(f) or r == 0)
        assert all(sorted(v) == list(range(8)) for v in seen.values()) and len(seen) == 2 * G     # every band of every frame exactly once
    for world in (1, 2, 3, 8, 128):                       # sky-LUT rows: every row of the 100 exactly once, whatever the split
        rows = []
        for r in range(world):
            r0, rs, rn = T.lut_rows_for_rank(100, r, world)
            assert rn == len(range(r0, 100, rs)) <= T.max_lut_rows(100, world)
            rows.extend(range(r0, 100, rs))
        assert sorted(rows) == list(range(100))
    with pytest.raises(ValueError):
        T.FrameGroups(0, 8, 3)
