// sweep_deal.cpp -- the destination side of the K1s layout (csrc/spmm.hip: build_blocked_sweep): which position each
// row (or piece of a split row) takes.  Pure host code so that it can be tested without a GPU (dory_sweep_deal).
//
// A sweep = `sweep_tiles` workgroups of 32 lane groups per XCD that move in step, so what a sweep costs is set by the
// rows per group of its workgroups, and a last sweep that is only partly occupied costs as much as a full one (Reddit:
// 91 workgroups of 320 rows per XCD = 2.84 sweeps, paid as 3).  The positions are therefore laid out for whole sweeps --
// 8 XCDs x S sweeps x sweep_tiles x 32 groups x R positions -- and the groups of the last sweep get fewer rows instead
// (their other positions stay empty): 10 + 10 + 9 rows per group instead of 3 x 10.  Groups are numbered as the kernel
// walks them: XCD, workgroup, group.  Items arrive sorted by descending weight (edges) and are dealt in bands over the
// groups that still have room, alternate bands in reverse (serpentine), so that groups with the same number of rows
// carry nearly the same number of edges.
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/dorylus_host.h"

namespace dory {

// returns false on overflow; cap[g] = rows of group g, npos = T * R.
// loader_relief: rows per sweep kept off lane groups 0 and 1 of every workgroup (the two groups of the workgroup's wave 0,
// which also copies the next step's entries for the other fifteen waves -- csrc/spmm.hip, LOADER); the other groups take
// them.  In the (shorter) last sweep the relief shrinks in proportion.
bool sweep_deal_plan(uint32_t nl, uint32_t R, uint32_t sweep_tiles, std::vector<uint32_t> *cap, uint32_t *npos,
                     uint32_t loader_relief) {
    if (R == 0) return false;
    loader_relief = std::min(loader_relief, R / 2);                    // (a layout of few rows per group: little or nothing to take off)
    const uint32_t tiles = std::max<uint32_t>(1, sweep_tiles);
    const uint32_t GS = tiles * 32u;                                   // groups per sweep and XCD
    const uint32_t n_x = (nl + 7) / 8;
    // rows of an ordinary group summed over the sweeps: the smallest `need` whose capacity (loader groups carry less) holds n_x
    auto relief_of = [&](uint32_t rows) { return (uint32_t)(((uint64_t)loader_relief * rows + R / 2) / R); };   // of a sweep with `rows` per group
    auto capacity = [&](uint32_t need) -> uint64_t {
        const uint32_t S = (need + R - 1) / R, last = need - (S - 1) * R;
        const uint64_t full = (uint64_t)GS * R - (uint64_t)2 * tiles * relief_of(R);
        const uint64_t tail = (uint64_t)GS * last - (uint64_t)2 * tiles * std::min(relief_of(last), last);
        return (uint64_t)(S - 1) * full + tail;
    };
    uint32_t need = std::max<uint32_t>(1, (n_x + GS - 1) / GS);
    const uint32_t S_plain = (need + R - 1) / R;                       // sweeps without any relief
    for (;; --loader_relief) {                                         // never pay for the relief with one more sweep
        need = std::max<uint32_t>(1, (n_x + GS - 1) / GS);
        while (capacity(need) < n_x) ++need;
        if ((need + R - 1) / R == S_plain || loader_relief == 0) break;
    }
    const uint32_t S = (need + R - 1) / R;
    if ((uint64_t)8 * S * GS * R > 0xFFFFFFF0ull) return false;
    const uint32_t T = 8u * S * GS;
    const uint32_t last = need - (S - 1) * R;
    cap->assign(T, R);
    for (uint32_t x = 0; x < 8; ++x)
        for (uint32_t sw = 0; sw < S; ++sw)
            for (uint32_t g = 0; g < GS; ++g) {
                const uint32_t rows = sw + 1 == S ? last : R;
                const uint32_t rel = (g % 32u) < 2u ? std::min(relief_of(rows), rows) : 0u;
                (*cap)[(size_t)x * S * GS + (size_t)sw * GS + g] = rows - rel;
            }
    // make the capacity exact: the surplus comes off groups spread evenly over the last sweeps of all XCDs
    uint64_t total = (uint64_t)8 * capacity(need);
    const uint32_t L = 8u * GS;
    for (uint32_t sw = S; sw-- > 0 && total > nl;) {
        while (total > nl) {
            const uint64_t surplus = std::min<uint64_t>(total - nl, L);
            bool any = false;
            for (uint64_t k = 0; k < surplus; ++k) {
                uint32_t j = (uint32_t)(k * L / surplus);                // j-th group of sweep sw, counted over the XCDs
                for (uint32_t probe = 0; probe < L; ++probe, j = (j + 1) % L) {   // (a group without rows: the next one that has some)
                    uint32_t &cg = (*cap)[(size_t)(j / GS) * S * GS + (size_t)sw * GS + j % GS];
                    if (cg) { --cg; --total; any = true; break; }
                }
            }
            if (!any) break;
        }
    }
    if (total != nl) return false;
    *npos = std::max<uint32_t>(T * R, 8);
    return true;
}

// pos[i] = position of item i (items sorted by descending weight).  Band b = the b-th heaviest row of every group that
// takes part in it.  An ordinary group takes part in bands [0, cap); a loader group (lane groups 0 and 1 of a workgroup,
// dealt fewer rows) sits out `loader_lo` of the HEAVIEST bands and the rest of its relief at the light end -- on a graph
// with skewed degrees the light bands alone would take rows off it but hardly any edges (build_blocked_sweep chooses
// loader_lo from the band weights).
bool sweep_deal_positions(uint32_t nl, uint32_t R, const std::vector<uint32_t> &cap, uint32_t *pos, uint32_t loader_lo) {
    const uint32_t T = (uint32_t)cap.size();
    auto takes = [&](uint32_t g, uint32_t band) {
        const uint32_t lo = (g % 32u) < 2u ? std::min(loader_lo, R - std::min(R, cap[g])) : 0u;
        return band >= lo && band < lo + cap[g];
    };
    uint32_t i = 0;
    for (uint32_t band = 0; band < R; ++band) {
        if (!(band & 1u)) { for (uint32_t g = 0; g < T; ++g) if (takes(g, band)) { if (i >= nl) return false; pos[i++] = g * R + band; } }
        else { for (uint32_t g = T; g-- > 0;) if (takes(g, band)) { if (i >= nl) return false; pos[i++] = g * R + band; } }
    }
    return i == nl;
}

// The deal by weight: item i (sorted by descending weight w[i]) goes to the group with the smallest load per row slot
// (load / cap) among those with a free slot -- longest-processing-time-first with row capacities.  Used when the groups'
// capacities differ on purpose (loader groups): every group then carries edges in proportion to its rows whatever the
// degree distribution, which skipping whole bands cannot do on a skewed graph.  Ties go to the lower group index: a pure
// function of (weights, capacities).
bool sweep_deal_balanced(uint32_t nl, uint32_t R, const std::vector<uint32_t> &cap, const uint64_t *w, uint32_t *pos) {
    const uint32_t T = (uint32_t)cap.size();
    struct Node { uint64_t load; uint32_t cap, g; };
    auto worse = [](const Node &a, const Node &b) {     // a after b in the queue: a.load / a.cap > b.load / b.cap, then higher g
        const unsigned __int128 l = (unsigned __int128)a.load * b.cap, r = (unsigned __int128)b.load * a.cap;
        return l != r ? l > r : a.g > b.g;
    };
    std::vector<Node> heap;
    heap.reserve(T);
    for (uint32_t g = 0; g < T; ++g)
        if (cap[g]) heap.push_back({0, cap[g], g});
    std::make_heap(heap.begin(), heap.end(), worse);
    std::vector<uint32_t> used(T, 0);
    for (uint32_t i = 0; i < nl; ++i) {
        if (heap.empty()) return false;
        std::pop_heap(heap.begin(), heap.end(), worse);
        Node n = heap.back();
        heap.pop_back();
        if (used[n.g] >= R) return false;
        pos[i] = n.g * R + used[n.g]++;
        n.load += w[i];
        if (used[n.g] < cap[n.g]) {
            heap.push_back(n);
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    }
    return heap.empty();
}

}  // namespace dory

extern "C" int dory_sweep_deal(uint32_t items, uint32_t rows_per_group, uint32_t sweep_tiles, uint32_t *positions_out,
                               uint32_t *group_rows_out, uint32_t *item_position) {
    std::vector<uint32_t> cap;
    uint32_t npos = 0;
    if (!dory::sweep_deal_plan(items, rows_per_group, sweep_tiles, &cap, &npos, 0)) return 1;
    if (positions_out) *positions_out = npos;
    if (group_rows_out) std::copy(cap.begin(), cap.end(), group_rows_out);
    if (item_position && !dory::sweep_deal_positions(items, rows_per_group, cap, item_position, 0)) return 2;
    return 0;
}

extern "C" int dory_sweep_deal_weighted(uint32_t items, const uint64_t *weights, uint32_t rows_per_group, uint32_t sweep_tiles,
                                        uint32_t loader_relief, uint32_t *positions_out, uint32_t *group_rows_out,
                                        uint32_t *item_position) {
    if (!weights) return 1;
    std::vector<uint32_t> cap;
    uint32_t npos = 0;
    if (!dory::sweep_deal_plan(items, rows_per_group, sweep_tiles, &cap, &npos, loader_relief)) return 1;
    if (positions_out) *positions_out = npos;
    if (group_rows_out) std::copy(cap.begin(), cap.end(), group_rows_out);
    if (item_position && !dory::sweep_deal_balanced(items, rows_per_group, cap, weights, item_position)) return 2;
    return 0;
}
