// gat_mh.hpp -- shared by gat_mh.hip (row-wise kernels, launchers) and gat_mh_blocked.hip (source-blocked kernels)
#ifndef DORY_GAT_MH_HPP
#define DORY_GAT_MH_HPP
#include "ctx.hpp"

namespace dory {

constexpr float GATMH_SLOPE = 0.2f;
constexpr int GATMH_MAXC = 4;   // K*D <= 256 (row-wise kernels are instantiated for 1, 2 or 4 chunks of 64 floats)

__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : GATMH_SLOPE * x; }

struct GatMhArgs {
    uint32_t N, K, D, ld /*of z, o, do, dz*/, ldk /*of el, er, m, den, t, del, der*/;
    const uint64_t *ptr;   // CSC (forward / dst pass) or CSR (src pass)
    const uint32_t *idx;
};

// K heads of D features fit the kernels: K*D <= 256, D a power of two <= 64 unless there is a single head
inline bool gatmh_shape_ok(uint32_t K, uint32_t D) {
    if (K == 0 || D == 0 || K > 64 || (uint64_t)K * D > 64 * GATMH_MAXC) return false;
    if (K == 1) return true;
    return (D & (D - 1)) == 0 && D <= 64;   // per-head reductions are xor-shuffles inside D lanes
}

}  // namespace dory
#endif
