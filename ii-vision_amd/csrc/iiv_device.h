// iiv_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Screen-model arithmetic restated for one wave64 lane per screen byte.  The
// reference keeps a packed (32,128) u64 "column" array (transcoder/screen.py:
// 169-226) and masks 13/14-bit windows out of it; here the same window is built
// directly from the three screen bytes that determine it (the byte itself and
// its two neighbours in dot order), which is what lets the kernels keep only
// raw memory-map bytes in LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iiv {

constexpr int kHGR = 0;   // video_mode.py:7
constexpr int kDHGR = 1;  // video_mode.py:8

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<kHGR> {
    static constexpr int kBits = 14;         // MASKED_BITS  screen.py:615
    static constexpr int kDots = 18;         // MASKED_DOTS  screen.py:624
    static constexpr int kOffsets = 2;       // len(BYTE_MASKS)
    static constexpr int kContentBits = 8;   // palette bit is part of the store
    static constexpr int kBanks = 1;
};
template <> struct ModeTraits<kDHGR> {
    static constexpr int kBits = 13;         // screen.py:886
    static constexpr int kDots = 10;         // screen.py:890
    static constexpr int kOffsets = 4;
    static constexpr int kContentBits = 7;   // palette bit masked (screen.py:939-941)
    static constexpr int kBanks = 2;
};

__host__ __device__ inline int masked_bits(int mode) { return mode == kDHGR ? 13 : 14; }
__host__ __device__ inline int masked_dots(int mode) { return mode == kDHGR ? 10 : 18; }
__host__ __device__ inline int num_offsets(int mode) { return mode == kDHGR ? 4 : 2; }
__host__ __device__ inline int content_bits(int mode) { return mode == kDHGR ? 7 : 8; }

// PHASES (screen.py:645, 919)
__host__ __device__ inline int phase_of(int mode, int o)
{
    return mode == kDHGR ? ((0x2301 >> (4 * o)) & 0xf)  // [1,0,3,2]
                         : (o == 0 ? 1 : 3);            // [1,3]
}

// SCREEN_HOLES (screen.py:42-62): offsets 120..127 and 248..255 of every page.
__host__ __device__ inline bool is_hole(int offset) { return (offset & 127) >= 120; }

// byte_offset (screen.py:694-700, 956-969): index of a screen byte inside its
// packed column.
template <int MODE> __host__ __device__ inline int byte_offset(int page_offset, int is_aux)
{
    int odd = page_offset & 1;
    if (MODE == kDHGR) return is_aux ? (odd ? 2 : 0) : (odd ? 3 : 1);
    return odd;
}

// The masked window (mask_and_shift_data of the packed column, screen.py:369-378)
// of the byte `own` at parity `odd`, given its predecessor and successor in dot
// order (0 beyond either end of the 256-byte page row, which is exactly where
// _pack zeroes header/footer, screen.py:217,224).
template <int MODE>
__host__ __device__ inline uint32_t masked_window(uint32_t prev, uint32_t own, uint32_t next, int odd)
{
    if (MODE == kDHGR) {
        // header = top 3 of the previous 7-bit byte, body, footer = low 3 of the next
        return ((prev & 0x7f) >> 4) | ((own & 0x7f) << 3) | ((next & 7) << 10);
    }
    // HGR (screen.py:566-569): ffFbbbbbbbBAaaaaaaaHhh.  Low 3 bits = data bits 5,6
    // and palette bit of the previous byte; high 3 = palette bit, data bits 0,1 of
    // the next byte; the byte itself sits in between, palette bit next to the
    // neighbouring column's.
    uint32_t lo = ((prev >> 5) & 3) | ((prev >> 7) << 2);
    uint32_t hi = ((next >> 7) << 11) | ((next & 3) << 12);
    uint32_t mid = odd ? ((((own & 0x7f) << 1) | (own >> 7)) << 3) : (own << 3);
    return lo | mid | hi;
}

// masked_update applied inside the window (screen.py:792-816, 993-1007).
template <int MODE> __host__ __device__ inline uint32_t poke_window(uint32_t m, uint32_t content, int odd)
{
    if (MODE == kDHGR) return (m & ~(0x7fu << 3)) | ((content & 0x7f) << 3);
    uint32_t mid = odd ? (((content & 0x7f) << 1) | (content >> 7)) : content;
    return (m & ~(0xffu << 3)) | (mid << 3);
}

// Neighbour bytes of screen byte (page row, offset y) of bank is_aux, in dot order.
//   DHGR columns interleave aux[2c], main[2c], aux[2c+1], main[2c+1] (screen.py:822-826)
//   HGR  columns are main[2c], main[2c+1].
// own_row / oth_row point at the 256-byte page row of the bank itself / the other bank.
template <int MODE, typename P>
__device__ inline void neighbours(P own_row, P oth_row, int y, int is_aux, uint32_t &prev, uint32_t &next)
{
    if (MODE == kDHGR) {
        if (is_aux) {
            prev = y > 0 ? oth_row[y - 1] : 0u;
            next = oth_row[y];
        } else {
            prev = oth_row[y];
            next = y < 255 ? oth_row[y + 1] : 0u;
        }
    } else {
        prev = y > 0 ? own_row[y - 1] : 0u;
        next = y < 255 ? own_row[y + 1] : 0u;
    }
}

// ---- MT19937 (CPython _random / numpy legacy RandomState) -----------------

__host__ __device__ inline uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__host__ __device__ inline uint32_t mt_mix(uint32_t a, uint32_t b)
{
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// dst = next 624-word block after src (dst != src), cooperatively by NT threads.
// Three dependency phases: [0,227) reads only src; [227,454) and [454,624) read
// dst words produced one phase earlier.  Ends with a barrier.
template <int NT> __device__ inline void mt_twist(const uint32_t *src, uint32_t *dst, int tid)
{
    for (int i = tid; i < 227; i += NT) dst[i] = src[i + 397] ^ mt_mix(src[i], src[i + 1]);
    __syncthreads();
    for (int i = 227 + tid; i < 454; i += NT) dst[i] = dst[i - 227] ^ mt_mix(src[i], src[i + 1]);
    __syncthreads();
    for (int i = 454 + tid; i < 624; i += NT) {
        uint32_t nx = (i == 623) ? dst[0] : src[i + 1];
        dst[i] = dst[i - 227] ^ mt_mix(src[i], nx);
    }
    __syncthreads();
}

// ---- wave64 helpers ---------------------------------------------------------

__device__ inline int lane_id() { return threadIdx.x & 63; }

// number of set bits of `mask` strictly below this lane
__device__ inline int prefix_popc(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

}  // namespace iiv
