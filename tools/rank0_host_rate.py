#!/usr/bin/env python
"""What rank 0 of an 8-rank run has to get through per frame on the HOST side (one GPU, one process): its 1/8 share with the sky-LUT rows,
a real RCCL gather (a world of one: RCCL's call path and its copy kernel, not its wire time) into an 8-member buffer, the interleave of
bands and LUT rows -- the loop of bench.py's N > 1 path with tiling.FrameGroups told it is rank 0 of 8.  A share takes ~0.22 ms per frame
with eight frames in flight (tools/share_matrix.py): if this loop is slower than that, rank 0 is host-bound and the split is too."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import gvcd_amd
from gvcd_amd import tiling

W, H, LW, LH = 2048, 1024, 200, 100
WORLD = int(os.environ.get("FAKE_WORLD", "8"))
FIF = int(os.environ.get("FIF", "8"))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


class OneOfMany:
    """torch.distributed for a world of one, asked to gather from WORLD ranks: member 0 is gathered for real, the others keep what they hold"""
    @staticmethod
    def gather(src, gather_list=None, dst=0, group=None, async_op=False):
        return dist.gather(src, gather_list=gather_list[:1], dst=0, async_op=async_op)


s = (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).astype(np.float32)
p = np.array([W, H, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.270588, 0.188235, 0.027451, 1.0, s[0], s[1], s[2], 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.05, 0.2, 0.0], np.float32)
ctx = gvcd_amd.Context(0)
ctx.set_noise(*gvcd_amd.assets.load_default_noise())
ctx.render_transmittance(256, 64)
ctx.set_frames_in_flight(FIF)
fg = tiling.FrameGroups(0, WORLD, 1, OneOfMany)
bands, mb = fg.bands(H), fg.max_bands(H)
band_bytes, lut_bytes = mb * 8 * W * 8, fg.max_lut_rows(LH) * LW * 8
lr = fg.lut_rows(LH)
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(FIF)]
local_b = [torch.zeros(band_bytes + lut_bytes, dtype=torch.uint8, device=dev) for _ in range(FIF)]
gathered = [torch.zeros((WORLD, band_bytes + lut_bytes), dtype=torch.uint8, device=dev) for _ in range(FIF)]
frame_out = [torch.zeros((H, W, 4), dtype=torch.int16, device=dev) for _ in range(FIF)]
lut_out = [torch.zeros((LH, LW, 4), dtype=torch.int16, device=dev) for _ in range(FIF)]
pending = [None] * FIF
T = {"render": 0.0, "gather": 0.0, "finish": 0.0}
out = [None, None]


SKIP = os.environ.get("SKIP", "")          # "gather", "interleave" or "gather,interleave": leave that part out (what does rank 0's extra time consist of?)


class Done:
    def wait(self):
        pass


def finish(o):
    t = time.perf_counter()
    with torch.cuda.stream(streams[o]):
        pending[o].wait(); pending[o] = None
        if "interleave" in SKIP:
            T["finish"] += time.perf_counter() - t
            return
        if os.environ.get("TORCH_INTERLEAVE"):                   # what round 4 started with: permute + copy on the whole chip
            img, lut = fg.split(gathered[o], H, W, LH, LW)
            out[0] = fg.assemble(0, img, H)
            out[1] = fg.assemble_lut(0, lut, LH)
        else:
            fg.assemble_device(0, gathered[o], ctx, streams[o].cuda_stream, H, W, frame_out[o], LH, LW, lut_out[o])
    T["finish"] += time.perf_counter() - t


def step(k):
    b = k % FIF
    st = streams[b].cuda_stream
    t = time.perf_counter()
    ctx.render_sky_lut_rows_device(s, lr[0], lr[1], local_b[b].data_ptr() + band_bytes, lut_bytes, LW, LH, st)
    ctx.render_clouds_device(p, W, bands, local_b[b].data_ptr(), W * 8, st)
    t1 = time.perf_counter(); T["render"] += t1 - t
    with torch.cuda.stream(streams[b]):
        pending[b] = Done() if "gather" in SKIP else fg.gather(k, local_b[b], gathered[b], async_op=True)
    T["gather"] += time.perf_counter() - t1
    o = (b + 1) % FIF
    if pending[o] is not None:
        finish(o)


for k in range(40):
    step(k)
torch.cuda.synchronize()
for key in T: T[key] = 0.0
N = 400
t0 = time.perf_counter()
for k in range(N):
    step(k)
host = time.perf_counter() - t0
for o in range(FIF):
    if pending[o] is not None: finish(o)
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("%srank 0 of %d, %d frames in flight: %.3f ms per frame in all, host loop alone %.3f ms per frame (render calls %.3f, gather call %.3f, wait + interleave calls %.3f)"
      % ("[without %s] " % SKIP if SKIP else "", WORLD, FIF, total / N * 1e3, host / N * 1e3, T["render"] / N * 1e3, T["gather"] / N * 1e3, T["finish"] / N * 1e3))
dist.destroy_process_group()
