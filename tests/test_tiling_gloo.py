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
