// exchange_args.h -- sizes and the kernel argument of the light-march packet exchange (protocol: exchange.h).  Plain C++: api.cpp allocates by these.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace csky {

#ifndef CSKY_XCH_K
#define CSKY_XCH_K 4
#endif
constexpr int XK = CSKY_XCH_K;                          // packets a wavefront may have outstanding
constexpr int XQ_N = 8192;                              // ring entries per XCD queue (a power of two; unpopped entries are bounded by the publishing policy)
constexpr uint32_t XQ_TICKET_MAX = (1u << 22) - 4096u;  // tickets per queue per launch that fit the entry tag
constexpr uint32_t XSEQ_MAX = (1u << 19) - 2u;          // packets per wavefront per launch that fit the granule tag
constexpr int XS_U64 = 12 * 64 + 2 * 64;                // one packet slot in 8-byte units (7168 bytes)
// control words (uint32), each group on its own 128-byte line.  Nothing here is polled in a loop by more than one wavefront at a time.
constexpr int XCTL_TAIL = 0;                            // [XCTL_TAIL + 32 y]: tickets handed to owners of XCD y (one returning add per published packet)
constexpr int XCTL_HEAD = 8 * 32;                       // [XCTL_HEAD + 32 y]: tickets taken by helpers of XCD y (one returning add per packet served)
constexpr int XCTL_ANY = 16 * 32;                       // [XCTL_ANY + y]: XCD y has helpers (written once per launch, read by its owners now and then)
constexpr int XCTL_DONE = 17 * 32;                      // finished tiles of the launch (one returning add per tile)
constexpr int XCTL_WORDS = 18 * 32;
constexpr int XCH_MAX_WAVES = 8192;                     // wavefront ids fit 13 bits of the queue entry
constexpr uint32_t XCH_EXIT = 0xffffffffu;              // low word of a queue entry that tells the ticket's holder to leave

constexpr int XDIAG_N = 16, XDIAG_STRIDE = 32;
struct XchArgs {                                        // kernel argument of clouds_kernel_exchange
    uint32_t* ctl;                                      // [XCTL_WORDS], all zero at launch; the last wavefront out zeroes it again
    unsigned long long* ring;                           // [8][XQ_N]
    unsigned long long* slots;                          // [resident wavefronts][XK][XS_U64]
    uint32_t* diag;                                     // optional counters (csky_exchange_counters), counter k at diag[k * XDIAG_STRIDE] (a 128-byte line each)
    uint32_t epoch;                                     // 1..1023
    uint32_t total_tiles;                               // 4 x entries of the launch order
};

}  // namespace csky
