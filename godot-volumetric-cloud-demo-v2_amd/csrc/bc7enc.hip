// bc7enc.hip -- the BC7 (BPTC) encoder on the GPU: one 4 x 4 block per lane (bc7enc_core.h).  Not on the per-frame path: it runs when a host
// asks for its textures "as the engine's importer would store them" (compress/mode=2: weather.bmp.import:19-20, worlnoise.bmp.import:19-20,
// perlworlnoise.tga.import:19-20), a few hundred thousand independent blocks per call.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "bc7enc_core.h"

namespace csky {

// n_img images of w x h RGBA8 texels back to back -> n_img x ceil(h/4) x ceil(w/4) blocks of 16 bytes, row-major per image
__global__ __launch_bounds__(64) void bc7_encode_kernel(const uint8_t* __restrict__ img, int w, int h, int n_img, int quality, uint4* __restrict__ out) {
    const int bw = (w + 3) >> 2, bh = (h + 3) >> 2;
    const size_t per = (size_t)bw * bh, i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= per * (size_t)n_img) return;
    const size_t im = i / per, b = i - im * per;
    unsigned char px[16][4];
    bc7_gather_block(img + im * (size_t)w * h * 4, w, h, (int)(b % bw), (int)(b / bw), px);
    uint32_t blk[4];
    bc7_encode_block(px, blk, quality);
    out[i] = make_uint4(blk[0], blk[1], blk[2], blk[3]);
}
hipError_t launch_bc7_encode(const uint8_t* d_img, int w, int h, int n_img, int quality, uint4* d_blocks, hipStream_t s) {
    const size_t total = (size_t)((w + 3) >> 2) * ((h + 3) >> 2) * (size_t)n_img;
    if (total) bc7_encode_kernel<<<(unsigned)((total + 63) / 64), 64, 0, s>>>(d_img, w, h, n_img, quality, d_blocks);
    return hipGetLastError();
}

}  // namespace csky
