// sweep_core.hpp -- the skeleton of K1s, the register-accumulating gated sweep (description: spmm.hip, "K1s"), as a
// device function template over an OP policy, so that the plain SpMM (spmm.hip: spmm_sweep_kernel) and the multi-head GAT
// edge passes (gat_mh_sweep.hip) run the same gates, the same loader wave and the same even layout (build_blocked_sweep).
// What an OP supplies: the per-row accumulators (Row), per-(row, step) constants (RowC), what one gathered entry does to a
// row (entry), an optional workgroup prologue (tables in LDS) and the store of a finished row or piece of a split row.
// The skeleton owns everything else: workgroup -> (XCD, slab, sweep, tile), the per-XCD gates, the LDS-DMA loader wave,
// the staging of the (idx, val) entries, the batches of four gathers through a buffer resource and the predicated tail.
#ifndef DORY_SWEEP_CORE_HPP
#define DORY_SWEEP_CORE_HPP
#include "spmm_common.hpp"

namespace dory {

constexpr int SWEEP_NT = 1024;
constexpr int SWEEP_C = 128;             // staged (idx,val) pairs per lane group and pass
constexpr int SWEEP_U = 4;               // gathers per batch
constexpr int SWEEP_SLACK = 1;           // start step b when all finished b - SWEEP_SLACK - 1
// Polling is bounded.  One poll (eight serialised L1-bypassing loads + a short sleep) is 4-5 us.  A gate on a step of the
// workgroup's own sweep waits for peers that are co-resident by assumption and at most a few steps behind: ~3 ms of
// polls is two orders above any legitimate wait.  A gate on the PREVIOUS sweep (the first SWEEP_SLACK + 1 steps) may
// legitimately wait for most of a sweep (surplus workgroups resident beside a sweep that left CUs to RCCL kernels):
// its limit grows with the number of steps.  After a timeout the launch finishes ungated and the context keeps its gates
// off for the next SWEEP_BACKOFF launches (the cause -- another process's kernels, a CU mask, RCCL holding more CUs than
// reserved -- rarely goes away within one launch), then tries again.
constexpr uint32_t SWEEP_SPIN_SHORT = 600, SWEEP_SPIN_PER_STEP = 40;
constexpr uint32_t SWEEP_BACKOFF = 16;

struct SweepArgs {
    uint32_t rpx;        // destination rows per XCD
    uint32_t tiles_x;    // workgroups per XCD and slab
    uint32_t G;          // workgroups per sweep (= CUs per XCD)
    uint32_t nsweeps;    // slabs * ceil(tiles_x / G)
    uint32_t b_lo, b_hi; // source blocks of this launch
    uint32_t *done;      // [8][nsweeps][b_hi - b_lo][32] arrival words + 1 "gates off" flag
    uint32_t flags;      // 2: second launch of an aggregation (pieces of split rows add to their slots); 8: no gates (diagnostic);
                         // 32: launched beside an exchange in flight (CUs left to its kernels): a back-off class of its own
    float *split_partial;// [B.nslots][ld]: bare sums of the pieces of split rows
    uint32_t *stat;      // per context, never reset by a launch: [0] gate timeouts, [1] first launch number that gates again,
                         // [2] launches that ran (partly) ungated after a timeout, [3] as [1] for the launches beside an exchange
                         // (a timeout there -- RCCL holding more CUs than reserved -- must not switch off the gates of
                         // the launches that run alone, and the other way round), [4] the context's K1s launch number,
                         // [5] workgroups of the launch in flight that have left.  The launch number lives on the device
                         // (the last workgroup to leave bumps it) so that a launch recorded into a hipGraph advances it
                         // on every replay: a kernel argument would be frozen at recording time and a back-off could
                         // never end.
};

// ---- the gate and the arrival, shared by the sweep kernels ------------------------------------------------------------
// gate: step sb may start once every workgroup of the sweep has finished step sb - SWEEP_SLACK - 1 (for the first steps:
// of the previous sweep of this XCD).  *lds_allowed = number of steps this workgroup may start.
template <int SLACK = SWEEP_SLACK>
__device__ __forceinline__ void sweep_gate(const SweepArgs &w, uint32_t seq, bool gated, uint32_t sb, uint32_t q, uint32_t nbs, uint32_t *dq,
                                           uint32_t cnt_q, uint32_t cnt_p, uint32_t *gates_off, uint32_t *lds_allowed,
                                           uint32_t *lds_lock, int lane) {
    const int bb = (int)sb - SLACK - 1;
    const bool prev = bb < 0;
    const uint32_t *word = !prev ? dq + (size_t)bb * 32
                                 : (q > 0 && (int)nbs + bb >= 0 ? dq - (size_t)nbs * 32 + (size_t)((int)nbs + bb) * 32 : nullptr);
    const uint32_t need = !prev ? cnt_q : cnt_p;
    if (word && lane == 0 && gated) {
        const uint32_t limit = prev ? SWEEP_SPIN_SHORT + SWEEP_SPIN_PER_STEP * nbs : SWEEP_SPIN_SHORT;
        while (__hip_atomic_load(lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= sb) {
            if (__hip_atomic_exchange(lds_lock, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                uint32_t spins = 0;      // this wave polls for the workgroup
                while (__hip_atomic_load(lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= sb) {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    uint32_t sum = 0;
#pragma unroll
                    for (int i = 0; i < 32; i += 4) {
                        u4 v;
                        const uint32_t *p = word + i;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                        sum += v.x + v.y + v.z + v.w;
                    }
                    if (sum >= need || __hip_atomic_load(gates_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > limit) {
                        // the sweep's workgroups are not co-resident (or not on this XCD): the rest of this launch
                        // and the context's next SWEEP_BACKOFF launches run ungated -- same results, counted
                        if (__hip_atomic_exchange(gates_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __hip_atomic_fetch_add(w.stat + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_max(w.stat + ((w.flags & 32u) ? 3 : 1), seq + 1u + SWEEP_BACKOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        break;
                    }
                }
                // (a wave at a later step may have raised it meanwhile: never lower it)
                __hip_atomic_fetch_max(lds_allowed, sb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(lds_lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
}

// the last wave of a workgroup to finish step sb reports it in the workgroup's own word of the (sweep, step) line
__device__ __forceinline__ void sweep_arrive(uint32_t sb, uint32_t t, uint32_t *dq, uint32_t *lds_cnt, int lane, int nwaves,
                                             uint32_t *lds_wgdone = nullptr /* LOADER: steps every wave of the workgroup has finished */) {
    if (lane == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&lds_cnt[sb & 7], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == (uint32_t)nwaves - 1) {
            __hip_atomic_store(&lds_cnt[sb & 7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lds_wgdone) __hip_atomic_store(lds_wgdone, sb + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dq + (size_t)sb * 32 + (t & 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// LOADER (32-lane groups): wave 0 of the workgroup copies every lane group's entries and the workgroup's row offsets of
// the NEXT step into LDS (LDS-DMA, three buffers) while all sixteen waves work on this one.  Why: a wave's vector loads
// return in order, so a wave that requests its own entries -- they stream from HBM, 2-3 us -- cannot get its next gathers
// back before them: 8-11 % of a launch with the gates on (profiles/r03_experiments.txt items 4, 9).  The wait is per
// wave: taken by one wave (whose lane groups the layout deals fewer rows: option spmm_sweep_loader_relief,
// host/sweep_deal.cpp) it is off the path of the other fifteen, which issue nothing but gathers.

#ifndef SWEEP_DMA_AUX
#define SWEEP_DMA_AUX 2   // cache policy of the loader's copies: nt (the entry stream is read once: it must not push the window out of L2)
#endif
template <int GROUP, int R, bool PAIR, bool LOADER, class OP>
__device__ __forceinline__ void sweep_run(const SpmmArgs &a, const BlockedAdj &B, const SweepArgs &w, OP &op) {
    constexpr bool UNIT = OP::UNIT_W;
    static_assert(!PAIR || OP::PLAIN, "rows in pairs: the plain SpMM only");
    constexpr int GPW = 64 / GROUP;
    constexpr int NGRP = SWEEP_NT / GROUP;
    constexpr int NW = SWEEP_NT / 64;
    constexpr int RW = NGRP * R;
    // staged (idx, val) pairs per lane group and pass.  With the loader wave a 16-lane launch (64 lane groups) gets 512-byte slots:
    // three buffers of 64 x 1 KB would not fit the LDS, and its groups hold at most a few rows (tens of entries per step)
    constexpr int C = (LOADER && GROUP == 16) ? SWEEP_C / 2 : SWEEP_C;
    constexpr int U = OP::BATCH, CQ = C / (2 * GROUP), CE = C - 1;   // CQ 16-byte loads of two entries per lane; a pass holds CE entries (its first may be the odd one of a pair)
    constexpr int NBUF = LOADER ? 3 : 1;
    constexpr int OFFB = (RW + 1 + 63) / 64 * 64;           // LOADER: the block's base (lo, hi) sits behind the copied offsets
    static_assert(!LOADER || (GROUP == 32 && C == 128) || (GROUP == 16 && C == 64), "the loader copies 1 KB per instruction: one slot of a 32-lane group, two of 16-lane groups");
    __shared__ uint2 stage[NBUF * NGRP][C];
    __shared__ uint32_t o_lds[LOADER ? 1 : NGRP][R + 2];
    __shared__ uint32_t offl[LOADER ? NBUF : 1][LOADER ? OFFB + 2 : 1];
    __shared__ uint32_t lds_allowed, lds_lock, lds_cnt[8], lds_ready, lds_wgdone, lds_seq;
    if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 8) lds_allowed = 0;
    if (threadIdx.x == 9) lds_lock = 0;
    if (threadIdx.x == 10) lds_ready = 0;
    if (threadIdx.x == 11) lds_wgdone = 0;
    if (threadIdx.x == 12) lds_seq = __hip_atomic_load(w.stat + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t seq = lds_seq;                            // launch number of the context (same for every workgroup: see leave)
    // leaving: the last workgroup of the launch to get here advances the context's launch number (nobody can still be
    // about to read it: every workgroup has read it before it counts itself out)
    auto leave = [&]() {
        if (threadIdx.x == 0 &&
            __hip_atomic_fetch_add(w.stat + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(w.stat + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(w.stat + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const uint32_t id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const uint32_t spp = (w.tiles_x + w.G - 1) / w.G;        // sweeps per slab
    const uint32_t tiles_pad = spp * w.G;
    const uint32_t slab = k / tiles_pad, t = k % tiles_pad;
    if (t >= w.tiles_x) { leave(); return; }                 // padding workgroup of a slab's last sweep
    const uint32_t q = k / w.G;                              // sweep, global over the slabs
    const uint32_t cnt_q = min(w.G, w.tiles_x - (q % spp) * w.G);
    const uint32_t cnt_p = q % spp == 0 ? min(w.G, w.tiles_x - (spp - 1) * w.G) : w.G;   // size of sweep q-1
    const uint32_t nbs = w.b_hi - w.b_lo;
    uint32_t *dq = w.done + ((size_t)xcd * w.nsweeps + q) * nbs * 32;
    uint32_t *gates_off = w.done + (size_t)8 * w.nsweeps * nbs * 32;
    // gates of this launch: off by request (diagnostic), or while the context backs off after a timeout
    const bool gated = !(w.flags & 8u) && !(seq < __hip_atomic_load(w.stat + ((w.flags & 32u) ? 3 : 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (!gated && blockIdx.x == 0 && threadIdx.x == 0 && !(w.flags & 8u))
        __hip_atomic_fetch_add(w.stat + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane % GROUP, gi = lane / GROUP;
    const int g = wave * GPW + gi;
    const uint32_t xend = min((xcd + 1) * w.rpx, B.npos);   // positions (= rows without B.perm)
    const uint32_t v0 = min(xcd * w.rpx + t * RW + (uint32_t)g * R, xend);
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl) + ccol;
    uint2 *st = stage[g];
    const uint32_t *ol = LOADER ? offl[0] : o_lds[g];
    const uint32_t orow = min(v0 + (uint32_t)min(li, R), xend);   // lane li <= R holds the offset of row v0 + li

    typename OP::Row rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) op.init(rows[r]);

    // Source rows through a buffer resource: 32-bit byte offsets (row id x row bytes in one 24-bit multiply-add), and
    // an absent slot of a tail is an out-of-range offset -- reads zeros, no memory access, no branch.  A launch covers
    // local-source blocks or ghost blocks, never both: one base.
    const bool ghost_launch = w.b_lo >= B.nb_local;
    const uint32_t row_b = a.ld * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(ghost_launch ? a.xg : a.xl), 0, (ghost_launch ? B.nghost : a.N) * row_b, 0x00020000);
    // a lane without a column (the last slab of a row narrower than the slabs) multiplies by 0 and adds -1: always out of range,
    // at no instruction
    const uint32_t lane_b = col_ok ? ccol * 16u - (ghost_launch ? a.N : 0u) * row_b : 0xFFFFFFFFu;   // (mod 2^32) + idx * row_b = byte offset
    const uint32_t lane_m = col_ok ? row_b : 0u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    auto gather = [&](uint32_t sidx, bool on) -> float4 {
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? __umul24(sidx, lane_m) + lane_b : 0xFFFFFFFFu, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // a pass of a step's entries: [cs, min(cs + CE, o_R)) of this group.  (idx, val) pairs are interleaved in the blocked
    // copy and travel two per lane and load (the addresser spends as long on a 4-byte load as on a 16-byte one); the
    // pair that holds the pass's first entry may begin one entry early
    auto load_entries = [&](uint64_t base, uint32_t cs, uint32_t oR, u4 (&en)[CQ]) {
        const uint64_t A0 = (base + cs) & ~1ull, Aend = base + oR;
#pragma unroll
        for (int qq = 0; qq < CQ; ++qq) {
            const uint64_t aa = A0 + 2u * (uint32_t)(qq * GROUP + li);
            en[qq] = (u4){0u, 0u, 0u, 0u};
            if (aa < Aend) en[qq] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(B.bent) + (aa >> 1));
        }
    };
    auto stage_entries = [&](uint64_t base, uint32_t cs, uint32_t ce, const u4 (&en)[CQ]) {
        const int p0 = (int)(((base + cs) & ~1ull) - base) - (int)cs;      // 0 or -1: stage index of the first loaded entry
#pragma unroll
        for (int qq = 0; qq < CQ; ++qq) {
            const int i0 = p0 + 2 * (qq * GROUP + li);
            if (i0 >= 0 && (uint32_t)i0 < ce - cs) st[i0] = make_uint2(en[qq].x, en[qq].y);
            if (i0 + 1 >= 0 && (uint32_t)(i0 + 1) < ce - cs) st[i0 + 1] = make_uint2(en[qq].z, en[qq].w);
        }
    };

    if constexpr (OP::PROLOGUE) {    // the OP's tables of this workgroup's rows (LDS)
        op.prologue(a, B, xcd * w.rpx + t * (uint32_t)RW, xend, w.b_lo >= B.nb_local, col, li);
        __syncthreads();
    }
    uint32_t my_o = 0;
    u4 en_pre[CQ];
    // ---- LOADER: the copies of one step, issued by wave 0 ----
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const uint32_t pos0 = xcd * w.rpx + t * RW;              // first position of the workgroup
    // lane j < NGRP of wave 0: first offset of lane group j in step `stp` (the addresses of the entry copies come out of
    // a register: a load between two copies would have to wait for the first)
    auto group_starts = [&](uint32_t stp) -> uint32_t {
        uint32_t ln = (uint32_t)min(lane, NGRP - 1);
        asm volatile("" : "+v"(ln));                          // (opaque, as in copy_step)
        return stp < nbs ? (B.boff + (size_t)(w.b_lo + stp) * (B.npos + 1))[min(pos0 + ln * R, xend)] : 0u;
    };
    auto block_base = [&](uint32_t stp) -> uint64_t { return stp < nbs ? B.bbase[w.b_lo + stp] : 0ull; };
    auto copy_step = [&](uint32_t stp, uint32_t gstart, uint64_t base) {
        const uint32_t bs = w.b_lo + stp, buf = stp % (uint32_t)NBUF;
        if constexpr (LOADER)
            if (lane == 0) { offl[buf][OFFB] = (uint32_t)base; offl[buf][OFFB + 1] = (uint32_t)(base >> 32); }
        const uint32_t *orow_b = B.boff + (size_t)bs * (B.npos + 1);
        uint32_t ln = (uint32_t)lane;                        // (opaque: the per-lane positions are two instructions each -- not
        asm volatile("" : "+v"(ln));                         //  worth hoisted 64-bit registers that end up in scratch and are
#pragma unroll                                               //  reloaded, one dependent miss after the other, by the loader)
        for (int j = 0; j < OFFB / 64; ++j)                  // the RW + 1 row offsets of the workgroup's positions
            if ((uint32_t)j * 64u + ln <= (uint32_t)RW)
                __builtin_amdgcn_global_load_lds((gptr_t)(orow_b + min(pos0 + (uint32_t)j * 64u + ln, xend)), (lptr_t)&offl[LOADER ? buf : 0][LOADER ? j * 64 : 0], 4, 0, SWEEP_DMA_AUX);
        if constexpr (GROUP == 32) {
#pragma unroll
            for (int gg = 0; gg < NGRP; ++gg) {              // one 1 KB run of entries per lane group, from the even entry at or before its first
                const uint64_t A0 = (base + (uint32_t)__builtin_amdgcn_readlane((int)gstart, gg)) & ~1ull;
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const u4 *>(B.bent) + (A0 >> 1) + lane), (lptr_t)&stage[buf * NGRP + gg][0], 16, 0, SWEEP_DMA_AUX);
            }
        } else {
#pragma unroll
            for (int gp = 0; gp < NGRP / 2; ++gp) {          // 16-lane groups: two 512-byte runs per instruction, the wave's halves on consecutive groups' slots
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)gstart, 2 * gp), s1 = (uint32_t)__builtin_amdgcn_readlane((int)gstart, 2 * gp + 1);
                const uint64_t A0 = (base + (ln < 32u ? s0 : s1)) & ~1ull;
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const u4 *>(B.bent) + (A0 >> 1) + (ln & 31u)), (lptr_t)&stage[buf * NGRP + 2 * gp][0], 16, 0, SWEEP_DMA_AUX);
            }
        }
    };
    auto uniform64 = [&](uint64_t v) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    // the loader's look-ahead: the lane groups' first offsets and the block base of the step after the one being copied
    // (requested behind the copies, back with them)
    uint32_t gs_next = 0;
    uint64_t bb_next = 0;
    if constexpr (LOADER) {
        if (wave == 0) {
            const uint32_t gs0 = group_starts(0);
            const uint64_t bb0 = uniform64(block_base(0));
            copy_step(0, gs0, bb0);
            gs_next = group_starts(1);
            bb_next = block_base(1);
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): the copies have landed
            bb_next = uniform64(bb_next);
            if (lane == 0) __hip_atomic_store(&lds_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        my_o = (B.boff + (size_t)w.b_lo * (B.npos + 1))[orow];
        load_entries(B.bbase[w.b_lo], (uint32_t)__shfl((int)my_o, 0, GROUP), (uint32_t)__shfl((int)my_o, R, GROUP), en_pre);
    }

    for (uint32_t b = w.b_lo; b < w.b_hi; ++b) {
        const uint32_t sb = b - w.b_lo;                      // step of this launch
        sweep_gate<OP::SLACK>(w, seq, gated, sb, q, nbs, dq, cnt_q, cnt_p, gates_off, &lds_allowed, &lds_lock, lane);
        uint32_t my_o_next = 0, o0, oR;
        uint64_t base;
        if constexpr (LOADER) {
            if (wave == 0 && sb + 1 < nbs) {
                // the buffer of step sb + 1 was last read in step sb - 2: every wave must be past it
                if (sb >= 2)
                    while (__hip_atomic_load(&lds_wgdone, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < sb - 1u) __builtin_amdgcn_s_sleep(2);
                copy_step(sb + 1, gs_next, bb_next);
                gs_next = group_starts(sb + 2);
                bb_next = block_base(sb + 2);
                __builtin_amdgcn_s_waitcnt(0x0f70);          // the one in-order wait of the workgroup, taken here
                bb_next = uniform64(bb_next);
                if (lane == 0) __hip_atomic_store(&lds_ready, sb + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            while (__hip_atomic_load(&lds_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < sb + 1u) __builtin_amdgcn_s_sleep(1);
            const uint32_t buf = sb % (uint32_t)NBUF;
            ol = &offl[buf][g * R];
            o0 = ol[0]; oR = ol[R];
            base = ((uint64_t)offl[buf][OFFB + 1] << 32) | offl[buf][OFFB];
            st = &stage[buf * NGRP + g][0] + (uint32_t)((base + o0) & 1ull);   // entry e of the first pass sits at slot e - o0 + (the copy began one entry early)
        } else {
            // offsets of the next step (in flight during this one)
            if (b + 1 < w.b_hi) my_o_next = (B.boff + (size_t)(b + 1) * (B.npos + 1))[orow];
            if (li <= R) o_lds[g][li] = my_o;
            o0 = (uint32_t)__shfl((int)my_o, 0, GROUP); oR = (uint32_t)__shfl((int)my_o, R, GROUP);
            base = B.bbase[b];
        }
        for (uint32_t cs = o0; cs < oR; cs += CE) {
            const uint32_t ce = min(cs + CE, oR);
            if constexpr (LOADER) {
                if (cs != o0) {   // (rare: more than CE entries of the group in one step) the group fetches the rest itself
                    st = &stage[(sb % (uint32_t)NBUF) * NGRP + g][0];
                    load_entries(base, cs, oR, en_pre);
                    stage_entries(base, cs, ce, en_pre);
                }
            } else {
                if (cs != o0) load_entries(base, cs, oR, en_pre);      // (rare: more than CE entries of the group in one step)
                stage_entries(base, cs, ce, en_pre);
            }
            if constexpr (PAIR) {
            // two consecutive rows of a lane group as one stream of entries: one tail per two rows, and the two lane
            // groups of a wave differ less over 20 entries than over 10; an entry goes to the first or the second row's
            // accumulator by its place (both products are formed, one with weight 0)
#pragma unroll
            for (int rr = 0; rr < R / 2; ++rr) {
                const uint32_t o0 = ol[2 * rr], o1 = ol[2 * rr + 1], o2 = ol[2 * rr + 2];
                const uint32_t lo = max(o0, cs), hi = min(o2, ce), mid = o1;
                if (lo < hi) {
                    uint32_t e = lo;
                    uint2 en[U];                                 // (read a batch ahead, as in the plain walk below)
#pragma unroll
                    for (int u = 0; u < U; ++u) en[u] = st[e + u - cs];
                    for (; e + U <= hi; e += U) {
                        float4 x[U];
                        float wv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) { x[u] = gather(en[u].x, true); wv[u] = UNIT ? 1.f : __uint_as_float(en[u].y); }
#pragma unroll
                        for (int u = 0; u < U; ++u) en[u] = st[e + U + u - cs];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const bool first = e + u < mid;
                            rows[2 * rr].acc = fma4(first ? wv[u] : 0.f, x[u], rows[2 * rr].acc);
                            rows[2 * rr + 1].acc = fma4(first ? 0.f : wv[u], x[u], rows[2 * rr + 1].acc);
                        }
                    }
                    if (e < hi) {
                        const uint32_t n = hi - e;
                        float4 x[U - 1];
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) x[u] = gather(en[u].x, (uint32_t)u < n);
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) {
                            const float wv = (uint32_t)u < n ? (UNIT ? 1.f : __uint_as_float(en[u].y)) : 0.f;
                            const bool first = e + u < mid;
                            rows[2 * rr].acc = fma4(first ? wv : 0.f, x[u], rows[2 * rr].acc);
                            rows[2 * rr + 1].acc = fma4(first ? 0.f : wv, x[u], rows[2 * rr + 1].acc);
                        }
                    }
                }
            }
            } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t rlo = ol[r], rhi = ol[r + 1];
                const uint32_t lo = max(rlo, cs), hi = min(rhi, ce);
                if (lo < hi) {
                    const typename OP::RowC rc = op.row_const((uint32_t)(g * R + r));
                    // the entries of a batch are read while the batch before it is in flight (the LDS round trip is off
                    // the chain LDS -> gathers -> sums that a wave repeats ~30 times per step); slots past the row's end
                    // read whatever is staged behind it (inside the LDS allocation) and are switched off below
                    uint32_t e = lo;
                    uint2 en[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) en[u] = st[e + u - cs];
                    for (; e + U <= hi; e += U) {               // full batches: nothing predicated
                        float4 x[U];
                        typename OP::Aux ax[U];                  // the entry's weight bits, or the OP's second gather
#pragma unroll
                        for (int u = 0; u < U; ++u) x[u] = gather(en[u].x, true);
                        if constexpr (OP::AUX_BATCH) {           // one second-table gather for the whole batch
                            op.template aux_batch<U>(&st[e - cs], (uint32_t)U, ax);
                        } else {
#pragma unroll
                            for (int u = 0; u < U; ++u) ax[u] = op.aux(en[u].x, en[u].y, true);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) en[u] = st[e + U + u - cs];
#pragma unroll
                        for (int u = 0; u < U; ++u) op.template entry<true>(rows[r], rc, x[u], ax[u], true);
                    }
                    if (e < hi) {                                // tail: 1 .. U-1 edges
                        const uint32_t n = hi - e;
                        float4 x[U - 1];
                        typename OP::Aux ax[U - 1];
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) x[u] = gather(en[u].x, (uint32_t)u < n);
                        if constexpr (OP::AUX_BATCH) {
                            op.template aux_batch<U - 1>(&st[e - cs], n, ax);
                        } else {
#pragma unroll
                            for (int u = 0; u < U - 1; ++u) ax[u] = op.aux(en[u].x, en[u].y, (uint32_t)u < n);
                        }
#pragma unroll
                        for (int u = 0; u < U - 1; ++u)
                            op.template entry<false>(rows[r], rc, x[u], ax[u], (uint32_t)u < n);
                    }
                }
            }
            }
        }
        if constexpr (!LOADER) {
            // first pass of the next step's entries: in flight across the gate
            my_o = my_o_next;
            if (b + 1 < w.b_hi)
                load_entries(B.bbase[b + 1], (uint32_t)__shfl((int)my_o, 0, GROUP), (uint32_t)__shfl((int)my_o, R, GROUP), en_pre);
        }
        sweep_arrive(sb, t, dq, lds_cnt, lane, NW, LOADER ? &lds_wgdone : nullptr);
    }

    // a finished row, or a piece of a split row (its bare sum goes to slot tgt & 0x7FFFFFFF of a side buffer; a combine
    // kernel of the OP's family finishes those rows)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t pos = v0 + r;
        const uint32_t v = pos < xend ? (B.perm ? B.perm[pos] : pos) : 0xFFFFFFFFu;
        if (v != 0xFFFFFFFFu && col_ok) {
            const uint32_t tgt = B.otgt ? B.otgt[pos] : v;
            op.store(rows[r], a, w, v, (tgt & 0x80000000u) != 0, tgt & 0x7FFFFFFFu, col, nchunk, xl4);
        }
    }
    leave();
}

}  // namespace dory
#endif
