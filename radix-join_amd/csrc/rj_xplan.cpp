// rj_xplan.cpp — exchange layout of a sharded join (see rj_xplan.hpp).  Host only.
#include "rj_xplan.hpp"

#include "rj_internal.hpp"

namespace rj {

int exchange_first_overflow(uint32_t world, uint32_t subs, const uint64_t* cnt) {
    for (uint32_t d = 0; d < world; ++d) {
        uint64_t tot = 0;
        for (uint32_t s = 0; s < world; ++s)
            for (uint32_t k = 0; k < subs; ++k) {
                const uint64_t c = cnt[((size_t)s * world + d) * subs + k];
                if (c > XPLAN_MAX_TUPLES || (tot += c) > XPLAN_MAX_TUPLES) return (int)d;
            }
    }
    return -1;
}

void exchange_plan(uint32_t world, uint32_t subs, uint32_t me, const uint64_t* cnt, ExchangePlan& out) {
    if (!world || !subs || me >= world || !cnt) throw_fmt(RJ_ERR_ARG, "exchange plan: bad world / digit count / rank");
    const int over = exchange_first_overflow(world, subs, cnt);
    if (over >= 0)
        throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 tuples on one rank (rank %d would receive them)", over);
    auto at = [&](uint32_t s, uint32_t d, uint32_t k) { return cnt[((size_t)s * world + d) * subs + k]; };
    out.send_off.assign(world, 0);
    out.send_cnt.assign(world, 0);
    out.recv_off.assign(world, 0);
    out.recv_cnt.assign(world, 0);
    out.seg_begin.assign((size_t)subs * world, 0);
    out.seg_end.assign((size_t)subs * world, 0);
    out.part_off.assign((size_t)subs + 1, 0);
    // what I send: my stage-A output is owner-major, so owner d's slice starts where the slices
    // of the owners before it end
    uint64_t pos = 0;
    for (uint32_t d = 0; d < world; ++d) {
        uint64_t n = 0;
        for (uint32_t k = 0; k < subs; ++k) n += at(me, d, k);
        out.send_off[d] = pos;
        out.send_cnt[d] = n;
        pos += n;
    }
    if (pos > XPLAN_MAX_TUPLES) throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 tuples on one rank (rank %u holds them)", me);
    // what I receive: source-major ranges; inside source s's range its digits 0 .. subs-1
    pos = 0;
    for (uint32_t s = 0; s < world; ++s) {
        out.recv_off[s] = pos;
        for (uint32_t k = 0; k < subs; ++k) {
            const uint64_t c = at(s, me, k);
            out.seg_begin[(size_t)k * world + s] = (uint32_t)pos;
            out.seg_end[(size_t)k * world + s] = (uint32_t)(pos + c);
            out.part_off[k + 1] += (uint32_t)c;
            pos += c;
        }
        out.recv_cnt[s] = pos - out.recv_off[s];
    }
    out.n_recv = pos;
    for (uint32_t k = 0; k < subs; ++k) out.part_off[k + 1] += out.part_off[k];
}

}  // namespace rj
