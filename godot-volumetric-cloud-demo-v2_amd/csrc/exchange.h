// exchange.h -- light-march packet exchange between the wavefronts of ONE persistent launch (device code, gfx950).
//
// Why: a launch of the cloud kernel ends with its heaviest tiles while the rest of the chip idles (the solo
// launch drains for the last ~27 % of its span; a rank's 1/8 share is bound by its slowest wavefront).  Five workgroup
// ORDERS lost against the static one; this moves WORK instead, at the granularity the compact march already has: a flush =
// the light march (clouds.glsl:186-199) + the state-independent shading terms (:178, :202-209) of 64 queued in-cloud
// samples, 7 floats in and 5 floats out per sample, independent of every other sample and of the owner ray's running state.
//
//   * A wavefront that finds every tile sequence of clouds_kernel_exchange empty becomes a HELPER of its XCD: it takes the next ticket of the
//     XCD's queue (one returning add), waits at that ticket's ring entry, runs the same light_march_terms() the owner would have run for the
//     packet published there, writes the five results back and takes the next ticket.  It leaves when it reads an EXIT entry.
//   * An OWNER (a wavefront marching a tile) whose XCD has helpers publishes its flushes instead of executing them, keeps marching primary
//     samples, and replays the results IN ORDER later (clouds.glsl:207-210 is the only part that depends on the ray's state); up to XK packets
//     are outstanding per wavefront, it waits for the oldest when all are.  Per-ray arithmetic and its order are those of march_compact, so
//     frames are byte-identical whoever executed a packet.
//   * The wavefront that finishes the launch's LAST tile writes EXIT into the first 64 unserved tickets of every queue; a helper that reads
//     EXIT at ticket tail + i passes it on to tickets tail + 64 (i + 1) + lane, so 64 + 64^2 + ... waiting helpers leave within three hops.
//
// No shared word is ever polled: a helper spins on ITS ticket's ring entry (16 entries per 128-byte line), an owner on its own packet slot.
// The shared words see one returning add per packet on the owners' side (tail) and one on the helpers' side (head), on different lines.
// (Builds that lost, on MI355X, C3 frame 2.04 ms without the exchange: helpers scanning eight {head, tail} pairs every 0.85 us: 145 ms;
//  one scan per 14 us with compare-exchange pops: 65 ms -- hundreds of exchanges in flight on one word, one winner per round trip;
//  add-and-wait tickets limited to 16 ahead of the tail with owners reading {head, tail, helpers} once per flush: 5.9 ms.)
//
// Transport (MI355X: eight XCDs, private L2s, per-CU L1 never refreshed by other CUs' stores; cdna_hip_programming.md §6 G16):
// every shared word is an 8-byte {tag, value} granule written by ONE relaxed agent-scope atomic store (sc1, write-through)
// and read by relaxed agent-scope atomic loads (sc1, L1 bypassed) until the tag matches: the data IS the flag, so no
// fence, no drain and no L1 invalidate (the L1 holds the texture cells the march lives on) on either side.  Tags carry a
// per-launch epoch (a plain kernel argument: these launches are never graph-captured) so that nothing has to be zeroed
// between launches; the host zeroes the pool when the epoch wraps (every 1023 launches of a ring slot).
//   packet slot (private to one wavefront, XK per wavefront):  in[7][64] granules | out[5][64] granules | rec[64] x 16 B
//       in  = position (3), density t, height fraction, the owner ray's step length and phase value; t == 0 marks an unused lane
//       out = D.rgb, 1 / max(1e-7, t), dt   (shade_terms)
//       rec = the owner's step records {mask lo, mask hi, first slot, steps << 8 | samples}: plain stores and loads of the
//             SAME lane of the SAME wavefront (never read by anyone else)
//   queue entry = {epoch << 22 | ticket + 1, wavefront << 19 | packet number}; the ticket comes from one returning add on
//       the queue's tail and the entry is written one primary step LATER (the owner never waits for the add's round trip).
#pragma once
#include "csky_common.h"
#include "exchange_args.h"

namespace csky {

typedef __attribute__((address_space(1))) unsigned long long xg64;
typedef __attribute__((address_space(1))) uint32_t xg32;
CSKY_D unsigned long long xld64(const unsigned long long* p) { return __hip_atomic_load((xg64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CSKY_D void xst64(unsigned long long* p, unsigned long long v) { __hip_atomic_store((xg64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CSKY_D uint32_t xld32(const uint32_t* p) { return __hip_atomic_load((xg32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CSKY_D uint32_t xadd32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add((xg32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CSKY_D uint32_t xrfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
CSKY_D unsigned long long xgranule(uint32_t tag, float v) { return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v); }

// Diagnostic build (-DCSKY_XCH_TIMING, tools/exchange_ab.py): sums of 80 ns units (100 MHz ticks >> 3) per phase in counters 5..15
#ifdef CSKY_XCH_TIMING
#define XT_NOW() wall_clock64()
#define XT_ADD(diag, k, t0) do { if ((diag) && (threadIdx.x & 63) == 0) (void)xadd32((diag) + (k) * XDIAG_STRIDE, (uint32_t)((wall_clock64() - (t0)) >> 3)); } while (0)
#else
#define XT_NOW() 0ull
#define XT_ADD(diag, k, t0) do { (void)(t0); } while (0)
#endif

// an owner wavefront's side of the exchange; every member is wave-uniform except tkv / anyv (meaningful in lane 0)
struct XchWave {
    unsigned long long* slot0;                          // this wavefront's XK packet slots
    uint32_t* ctl;                                      // the launch's control words
    unsigned long long* ring;                           // its XCD's queue
    uint32_t* diag;
    uint32_t* tnote;                                    // CSKY_XCH_TIMING: publish times of the outstanding packets (LDS, XK words)
    uint32_t xcc, wave_id, epoch;
    uint32_t pub, rep;                                  // packets published / replayed by this wavefront in this launch
    uint32_t pend;                                      // packet pub-1 has a ticket in flight and no queue entry yet
    uint32_t tkv;                                       // lane 0: that ticket
    uint32_t have, asked, nfl;                          // this XCD has helpers | a read of its flag is in flight (anyv) | flushes since the last read
    uint32_t anyv;                                      // lane 0: the flag as read
};

// Does this wavefront's XCD have helpers yet?  One read of a write-once word at the start of a tile and every 2nd flush until it says yes
// (loads return in order: a slow read would stall the next primary step's texture fetch, so it is not issued per flush).
CSKY_D void xch_ask(XchWave& x, bool lane0, bool force) {
    if (!x.have && !x.asked && (force || (++x.nfl & 1u) == 0u)) {
        if (lane0) x.anyv = xld32(x.ctl + XCTL_ANY + x.xcc);
        x.asked = 1u;
    }
}
CSKY_D bool xch_should_publish(XchWave& x) {
#ifdef CSKY_XCH_NOPUB
    return false;                                       // experiment build: the exchange kernel's own cost without a single publication
#endif
    if (!x.have && x.asked) { x.have = xrfl(x.anyv) != 0u ? 1u : 0u; x.asked = 0u; }
    return x.have && x.pub < XSEQ_MAX;
}
// write the queue entry of the packet whose ticket is in flight (called one primary step after the publication, and before any wait)
CSKY_D void xch_flush_entry(XchWave& x, bool lane0) {
    if (x.pend) {
        if (lane0) {
            const uint32_t tk = x.tkv;
            xst64(x.ring + (tk & (uint32_t)(XQ_N - 1)), ((unsigned long long)((x.epoch << 22) | ((tk + 1u) & 0x3fffffu)) << 32) | (unsigned long long)((x.wave_id << 19) | (x.pub - 1u)));
        }
        x.pend = 0u;
    }
}
// N granule planes of a packet slot: sweep until every lane's tag matches.  The spin is bounded (about a second): a protocol fault shows up as a
// counted time-out (diag[4], csky_exchange_counters) and a wrong frame, never as a hung GPU.
constexpr uint32_t XCH_SPIN_LIMIT = 1u << 20;
template <int N>
CSKY_D void xch_sweep(const unsigned long long* planes, int lane, uint32_t tag, float (&v)[N], uint32_t* diag) {
    for (uint32_t spins = 0;; spins++) {
        bool ok = true;
#pragma unroll
        for (int p = 0; p < N; p++) {
            const unsigned long long g = xld64(planes + p * 64 + lane);
            v[p] = __uint_as_float((uint32_t)g);
            ok = ok && (uint32_t)(g >> 32) == tag;
        }
        if (__all(ok)) return;
        if (spins == 0u && diag && lane == 0) (void)xadd32(diag + 2 * XDIAG_STRIDE, 1u);
        if (spins >= XCH_SPIN_LIMIT) { if (diag && lane == 0) (void)xadd32(diag + 4 * XDIAG_STRIDE, 1u); return; }
        __builtin_amdgcn_s_sleep(4);
    }
}

}  // namespace csky
