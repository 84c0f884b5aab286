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

// The wave's TWO smallest values in one pass.  In: this lane's own two smallest, k1 <= k2 (values that are equal across lanes
// must be ones the caller does not care about: a value present in two lanes would be counted twice).  A step merges a lane's
// pair with its DPP partner's: three fused-DPP ops (min / max of the firsts, min of the seconds) and one min, each reading a
// register written at least two instructions earlier -- no s_nop between them, where two separate minima (the second needs
// the first's result to find the lane that gives up its key) cost 2 x (6 DPP + 6 s_nop) and a select chain in between.
// Out: (k1, k2) of lane 63 = the wave's two smallest, in SGPRs.
// (Written as one asm block: from C++ the compiler fuses only two of a step's three DPP reads and adds a v_mov, a v_mov_dpp and
// an INT_MAX materialisation per step -- six instructions instead of four.  Hazards: a DPP op may read a VGPR two instructions
// after the VALU op that wrote it; every read below keeps that distance, the s_nop covers the producers in front of the block.
// A lane whose DPP source does not exist (row_bcast into row 0, or rows 0 / 1) keeps what its destination held: nonsense that
// only such lanes ever read.)
__device__ static inline void wave_top2_i32(int &k1, int &k2)
{
    int a, h, m;
#define IIV_TOP2_STEP(dst, src, ctrl)                                              \
    "v_min_i32_dpp " dst ", " src ", " src " " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_max_i32_dpp %3, " src ", " src " " ctrl " row_mask:0xf bank_mask:0xf\n\t"      \
    "v_min_i32_dpp %4, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                \
    "v_min_i32_e32 %1, %3, %4\n\t"
    asm volatile("s_nop 1\n\t"
                 IIV_TOP2_STEP("%2", "%0", "quad_perm:[1,0,3,2]")
                 IIV_TOP2_STEP("%0", "%2", "quad_perm:[2,3,0,1]")
                 IIV_TOP2_STEP("%2", "%0", "row_half_mirror")
                 IIV_TOP2_STEP("%0", "%2", "row_mirror")       // every lane holds its row's pair
                 IIV_TOP2_STEP("%2", "%0", "row_bcast:15")     // row r + row r - 1
                 IIV_TOP2_STEP("%0", "%2", "row_bcast:31")     // lane 63: all four rows, each once
                 : "+v"(k1), "+v"(k2), "=&v"(a), "=&v"(h), "=&v"(m));
#undef IIV_TOP2_STEP
    k1 = __builtin_amdgcn_readlane(k1, 63);
    k2 = __builtin_amdgcn_readlane(k2, 63);
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
