// rj_xplan.hpp — who sends what to whom in the exchange step of a sharded join, and how what
// arrives feeds the next radix pass.  Host-only arithmetic over the all-gathered count tensor:
// no HIP, no RCCL (unit-tested on CPU through rj_exchange_plan, include/rj.h).
// There is no reference counterpart (the reference is one CPU process).
//
// Stage A of a sharded join partitions a rank's tuples by (owner rank, first local digit) in ONE
// radix pass; its output is laid out owner-major: [owner 0: digit 0 .. S-1][owner 1: ...] ...
// The count tensor every rank all-gathers is
//     cnt[(src * world + dst) * subs + sub] = tuples rank `src` holds for owner `dst`, digit `sub`.
// Rank `me` sends owner d's slice (one contiguous range) to rank d and receives one contiguous
// range from every source, source-major.  Inside a received range the digits still lie one behind
// the other, so the receive buffer is a set of world * subs runs; the next pass takes them as
// input segments in DIGIT-major order (all runs of digit 0, then digit 1, ...), `world` input
// segments feeding one output segment (the rank's first-level partition of that digit).
#pragma once
#include <cstdint>
#include <vector>

namespace rj {

struct ExchangePlan {
    uint64_t              n_recv = 0;          // tuples this rank receives
    std::vector<uint64_t> send_off, send_cnt;  // [world] tuples: slice of my stage-A output for rank d
    std::vector<uint64_t> recv_off, recv_cnt;  // [world] tuples: where rank s's slice lands
    std::vector<uint32_t> seg_begin, seg_end;  // [subs * world] run (sub, src) at index sub * world + src
    std::vector<uint32_t> part_off;            // [subs + 1] prefix of the first-level partition sizes
};

// most tuples a rank may hold (positions are 32-bit)
constexpr uint64_t XPLAN_MAX_TUPLES = 0xfffffff0ull;

// The first rank of the WORLD whose receive total exceeds XPLAN_MAX_TUPLES, or -1.  Every rank
// evaluates this over the same tensor, so all of them take the same decision BEFORE any
// collective moves data (a rank that gave up alone would leave its peers blocked).
int exchange_first_overflow(uint32_t world, uint32_t subs, const uint64_t* cnt);

// Rank `me`'s half of the exchange.  Throws rj::Error(RJ_ERR_ARG) on a malformed request and
// RJ_ERR_UNSUPPORTED when some rank's receive total overflows.
void exchange_plan(uint32_t world, uint32_t subs, uint32_t me, const uint64_t* cnt, ExchangePlan& out);

}  // namespace rj
