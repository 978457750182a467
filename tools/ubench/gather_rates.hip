// gather_rates.hip -- what the vector memory path of gfx950 sustains for the cloud kernel's access shape: one 16-byte
// gather per lane (global_load_dwordx4, 64 independent addresses per wave instruction), by footprint (which cache level
// serves it) and by how many distinct 128-byte lines a wave instruction touches.  The guide states no L1 bytes/clk for
// gathers, so the "L1-gather" roof of bench.py is MEASURED here, not assumed.
//   rate = wave-level load instructions per second (x 64 lanes x 16 B = bytes/s the lanes receive)
// Patterns (per wave instruction):
//   coalesced : lane l reads base + 16 l                      ( 8 lines, 1 KiB contiguous: the streaming shape)
//   same-line : all lanes inside ONE 128-byte line            ( 1 line : a wavefront looking at one texel cell row)
//   lines-K   : K distinct random lines, lanes spread over them (K = 4, 16)
//   random    : every lane an independent random 16-byte slot  (~64 lines: horizon rays, texels hundreds apart)
// Footprints: 16 KiB (L1), 1 MiB (L2 of every XCD), 32 MiB (beyond one XCD's 4 MiB L2: Infinity Cache), 1 GiB (HBM).
// 8 waves per SIMD, 4 independent loads in flight per lane per iteration (addresses do not depend on loaded data), plus a
// dependent-chain run (1 wave per SIMD, next address from the loaded value) for the latency.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rates.hip -o gather_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xs(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

// mode 0 coalesced, 1 same-line, 2 random, 3 lines-K (K = kparam)
template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, unsigned slot_mask, int iters, int kparam, uint4* __restrict__ out,
                                              unsigned long long* __restrict__ clk) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned rw = xs(wave * 2654435761u + 12345u);          // wave-uniform stream (same value in all lanes)
    unsigned rl = xs((wave * 64 + lane) * 747796405u + 1u);  // per-lane stream
    uint4 acc = make_uint4(0, 0, 0, 0);
    unsigned long long t0 = 0, r0 = 0;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned slot;                                   // index in 16-byte units
            if (MODE == 0) { rw = xs(rw); slot = ((rw & slot_mask) & ~63u) + lane; }
            else if (MODE == 1) { rw = xs(rw); slot = ((rw & slot_mask) & ~7u) + (lane & 7); }
            else if (MODE == 2) { rl = xs(rl); slot = rl & slot_mask; }
            else { rw = xs(rw); const unsigned line = xs(rw + (lane % (unsigned)kparam) * 0x9E3779B9u); slot = ((line & slot_mask) & ~7u) + ((lane / (unsigned)kparam) & 7); }
            const uint4 v = buf[slot];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
        clk[0] = t1 - t0; clk[1] = r1 - r0;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = acc;
}

// dependent chain: the next slot comes from the value just loaded (buffer pre-filled with a random permutation-ish table)
__global__ __launch_bounds__(256) void chase(const uint4* __restrict__ buf, unsigned slot_mask, int iters, uint4* __restrict__ out, unsigned long long* __restrict__ clk) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned slot = xs((wave * 64 + lane) * 747796405u + 1u) & slot_mask;
    unsigned long long t0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    uint4 v = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) { v = buf[slot]; slot = v.x & slot_mask; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t1; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); clk[0] = t1 - t0; }
    if (v.y == 0x12345678u) out[threadIdx.x] = v;
}

template <int MODE> int run(const char* name, const uint4* d, size_t bytes, int kparam, uint4* d_out, unsigned long long* d_clk, int cus) {
    const unsigned slot_mask = (unsigned)(bytes / 16 - 1);
    const int blocks = cus * 8;                                 // 8 waves per SIMD
    const int iters = bytes > (64u << 20) ? 200 : 1000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    gather<MODE><<<blocks, 256>>>(d, slot_mask, 20, kparam, d_out, nullptr);     // warm the caches
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    gather<MODE><<<blocks, 256>>>(d, slot_mask, iters, kparam, d_out, d_clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk[2]; CHK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
    const double mhz = (double)clk[0] / (double)clk[1] * 100.0;
    const double winstr = (double)blocks * 4 * iters * 4;       // wave-level load instructions
    const double per_cu_cycles = ms * 1e-3 * mhz * 1e6 / (winstr / cus);
    printf("%-10s %8.0f KiB  %8.3f ms  %8.1f Gload-instr/s  %8.2f TB/s to lanes  %6.1f cycles per wave-load per CU  (%4.0f MHz)\n", name, bytes / 1024.0, ms,
           winstr / (ms * 1e-3) / 1e9, winstr * 1024.0 / (ms * 1e-3) / 1e12, per_cu_cycles, mhz);
    return 0;
}

int main(int argc, char** argv) {
    const bool only_big = argc > 1 && argv[1][0] == 'b';      // "big": the 32 MiB and 1 GiB footprints only (FETCH_SIZE calibration pass), no latency chase
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxb = (size_t)1 << 30;
    uint4* d; uint4* d_out; unsigned long long* d_clk;
    CHK(hipMalloc(&d, maxb)); CHK(hipMalloc(&d_out, 4096)); CHK(hipMalloc(&d_clk, 64));
    {   // fill: x = pseudo-random slot index (for the chase), rest arbitrary
        std::vector<uint4> h(maxb / 16);
        unsigned r = 88172645u;
        for (size_t i = 0; i < h.size(); i++) { r ^= r << 13; r ^= r >> 17; r ^= r << 5; h[i] = make_uint4(r, (unsigned)i, r * 3u, ~r); }
        CHK(hipMemcpy(d, h.data(), maxb, hipMemcpyHostToDevice));
    }
    printf("# %s, %d CUs; 16-byte gathers (global_load_dwordx4), 8 waves/SIMD, 4 independent loads per lane per iteration\n", prop.gcnArchName, cus);
    const size_t sizes[] = {(size_t)16 << 10, (size_t)1 << 20, (size_t)32 << 20, (size_t)1 << 30};
    for (size_t b : sizes) {
        if (only_big && b < ((size_t)32 << 20)) continue;
        if (run<0>("coalesced", d, b, 0, d_out, d_clk, cus)) return 1;
        if (run<1>("same-line", d, b, 0, d_out, d_clk, cus)) return 1;
        if (run<3>("lines-4", d, b, 4, d_out, d_clk, cus)) return 1;
        if (run<3>("lines-16", d, b, 16, d_out, d_clk, cus)) return 1;
        if (run<2>("random", d, b, 0, d_out, d_clk, cus)) return 1;
    }
    if (only_big) return 0;
    for (size_t b : sizes) {   // latency: one wave per SIMD, dependent loads
        const unsigned slot_mask = (unsigned)(b / 16 - 1);
        chase<<<cus, 256>>>(d, slot_mask, 50, d_out, d_clk);
        CHK(hipDeviceSynchronize());
        chase<<<cus, 256>>>(d, slot_mask, 2000, d_out, d_clk);
        CHK(hipDeviceSynchronize());
        unsigned long long c; CHK(hipMemcpy(&c, d_clk, 8, hipMemcpyDeviceToHost));
        printf("dependent random 16-byte gather, %8.0f KiB footprint, 1 wave/SIMD: %6.0f cycles per load\n", b / 1024.0, (double)c / 2000.0);
    }
    return 0;
}
