// iiv_wave.h -- wave64 building blocks shared by the greedy kernels (iiv_greedy.hip,
// iiv_team.hip): fused-DPP reductions, the packed min-plus combine of the split store table.
#pragma once

#include "iiv_stream.h"

namespace iiv {

#define IIV_SGPR(x) __builtin_amdgcn_readfirstlane((int)(x))

// v_min_i32_dpp: `old` is the identity, so the mov folds into the min (one VALU op)
template <int CTRL> __device__ static inline int min_dpp(int v)
{
    int o = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, 0xf, 0xf, false);
    return o < v ? o : v;
}

// signed minimum over the wave, returned in an SGPR
__device__ static inline int wave_min_i32(int v)
{
    v = min_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = min_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = min_dpp<0x141>(v);  // row_half_mirror
    v = min_dpp<0x140>(v);  // row_mirror: every lane holds its row's minimum
    v = min_dpp<0x142>(v);  // row_bcast:15
    v = min_dpp<0x143>(v);  // row_bcast:31: lane 63 holds the wave's
    return __builtin_amdgcn_readlane(v, 63);
}

template <int CTRL> __device__ static inline uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}

// merge this lane's sorted pair (k1 <= k2) with the pair held by its DPP partner
template <int CTRL> __device__ static inline void top2_step(uint32_t &k1, uint32_t &k2)
{
    uint32_t o1 = dpp_u32<CTRL>(k1), o2 = dpp_u32<CTRL>(k2);
    uint32_t lo = k1 < o1 ? k1 : o1, hi = k1 < o1 ? o1 : k1;
    uint32_t m2 = k2 < o2 ? k2 : o2;
    k1 = lo;
    k2 = hi < m2 ? hi : m2;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// min(l0 + r0, l1 + r1) of two packed halves: v_pk_add_u16 + one min
__device__ static inline uint32_t combine(uint32_t l, uint32_t r)
{
    const u16x2 s = __builtin_bit_cast(u16x2, l) + __builtin_bit_cast(u16x2, r);
    const uint32_t a = s.x, b = s.y;
    return a < b ? a : b;
}

}  // namespace iiv
