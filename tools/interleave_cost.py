import sys, time; sys.path.insert(0,'.')
import torch, gvcd_amd
from gvcd_amd import tiling
H,W,N=1024,2048,8
g=torch.zeros((N, tiling.max_bands(H,N)*8, W, 4), dtype=torch.int16, device='cuda')
for _ in range(5): f=tiling.interleave(g,H,N)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): f=tiling.interleave(g,H,N)
torch.cuda.synchronize(); print("interleave of a gathered 2048x1024 frame (8 ranks): %.1f us" % ((time.perf_counter()-t)/200*1e6))
src=torch.zeros((N-1, 2*1024*1024), dtype=torch.uint8, device='cuda'); dst=torch.empty_like(src)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(200): dst.copy_(src)
torch.cuda.synchronize(); print("local copy of 7 x 2 MiB (what the gather writes on rank 0): %.1f us" % ((time.perf_counter()-t)/200*1e6))
