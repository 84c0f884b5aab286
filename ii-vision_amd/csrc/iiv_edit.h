// iiv_edit.h -- colour strings and the edit-distance recurrence, shared by the
// table builder (iiv_tables.hip) and the encoder prologue (iiv_prologue.hip).
// Reference: transcoder/screen.py:712-789, 983-990; colours.py:100-148;
// make_data_tables.py:30-41, 92-108.
#pragma once

#include "iiv_device.h"

namespace iiv {

// ------------------------------------------------------------------ colour strings

// HGRBitmap._double_pixels (screen.py:712-739)
__host__ __device__ inline uint32_t double_pixels(uint32_t v)
{
    uint32_t r = 0;
    for (int k = 0; k < 7; k++)
        if (v & (1u << k)) r |= 3u << (2 * k);
    if (v & 0x40) r |= 1u << 14;
    return r;
}

// to_dots (screen.py:743-789; DHGR: identity, screen.py:983-990)
template <int MODE> __host__ __device__ inline uint32_t to_dots(uint32_t m, int o)
{
    if (MODE == kDHGR) return m;
    uint32_t h = (m & 7) << 5;
    uint32_t hp = (h & 0x80) >> 7;
    uint32_t res = double_pixels(h & 0x7f) >> (11 - hp);
    uint32_t b, bp;
    if (o == 0) {
        b = (m >> 3) & 0xff;
        bp = (b & 0x80) >> 7;
    } else {
        bp = (m >> 3) & 1;
        b = ((m >> 4) & 0x7f) ^ (bp << 7);
    }
    res &= ~(0x3fffu << (3 + bp));
    res ^= double_pixels(b & 0x7f) << (3 + bp);
    uint32_t f = ((m >> 12) & 3) ^ (((m >> 11) & 1) << 7);
    uint32_t fp = (f & 0x80) >> 7;
    res &= ~(0xfu << (17 + fp));
    res ^= double_pixels(f & 0x7f) << (17 + fp);
    return res & ((1u << 21) - 1);
}

// Colour value of pixel k = rol4(dots[k..k+3], (phase + k) & 3) (colours.py:100-134);
// the string is returned as packed nibbles, pixel k in nibble k (lo: 0..15, hi: 16..).
template <int MODE> __host__ __device__ inline void colour_string(uint32_t m, int o, uint64_t &lo, uint32_t &hi)
{
    const uint32_t dots = to_dots<MODE>(m, o);
    const int ph = phase_of(MODE, o);
    // 32-bit accumulators of 8 pixels each; rol4(w, r) = ((w | w << 4) >> (4 - r)) & 15
    uint32_t w[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < ModeTraits<MODE>::kDots; k++) {
        const uint32_t win = (dots >> k) & 0xf;
        const uint32_t c = ((win | (win << 4)) >> (4 - ((ph + k) & 3))) & 0xf;
        w[k >> 3] |= c << (4 * (k & 7));
    }
    lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    hi = w[2];
}

template <int K> __device__ inline uint32_t nib(uint64_t lo, uint32_t hi)
{
    if (K < 8) return ((uint32_t)lo >> (4 * K)) & 0xf;
    if (K < 16) return ((uint32_t)(lo >> 32) >> (4 * (K - 8))) & 0xf;
    return (hi >> (4 * (K - 16))) & 0xf;
}

// weighted Damerau-Levenshtein between two equal-length colour strings with
// insert/delete cost 1e5 and transpose cost 1 (make_data_tables.py:30-41,98-104)
// reduces to E[k] = min(E[k-1] + sub(a_k,b_k), E[k-2] + 1 if a_{k-1}a_k == b_k b_{k-1}).
// lut = 16x16 substitute costs (u16) in LDS, diagonal forced to 0 (see load_cost_lut).
// stage the 16x16 costs in LDS with a zero diagonal (call with >= 256 threads, then barrier)
__device__ inline void load_cost_lut(uint16_t *lut, const uint16_t *sub, int tid)
{
    if (tid < 256) lut[tid] = ((tid >> 4) == (tid & 15)) ? (uint16_t)0 : sub[tid];
}

template <int N, int K> struct EditStep {
    __device__ static inline void run(uint64_t alo, uint32_t ahi, uint64_t blo, uint32_t bhi,
                                      const uint16_t *lut, uint32_t &e1, uint32_t &e2)
    {
        uint32_t a = nib<K>(alo, ahi), b = nib<K>(blo, bhi);
        // lut's diagonal is zeroed by whoever loads it (equal pixels cost nothing,
        // make_data_tables.py / weighted_levenshtein), so the read needs no branch
        uint32_t s = (uint32_t)lut[a * 16 + b];
        uint32_t e = e1 + s;
        if (K >= 1) {
            uint32_t ap = nib<(K >= 1 ? K - 1 : 0)>(alo, ahi), bp = nib<(K >= 1 ? K - 1 : 0)>(blo, bhi);
            uint32_t t = e2 + 1;
            if (ap == b && a == bp && t < e) e = t;
        }
        e2 = e1;
        e1 = e;
        EditStep<N, K + 1>::run(alo, ahi, blo, bhi, lut, e1, e2);
    }
};
template <int N> struct EditStep<N, N> {
    __device__ static inline void run(uint64_t, uint32_t, uint64_t, uint32_t, const uint16_t *, uint32_t &, uint32_t &) {}
};

// ---- HGR windows as dots: two small lookups (the prologue's diff weights, iiv_prologue.hip) ----------------
// to_dots<kHGR> of a 14-bit window is the OR of what its low seven bits and its high seven bits paint, each taken
// together with the palette bit of the byte itself (bit 10 of an even byte's window, bit 3 of an odd one's), which
// decides where the body's doubled dots start (checked for every window of both parities by hgr_dot_lut_check,
// iiv_tables.hip, at table build time).  Table layout, u32 entries holding `dots << 1` (bit 0 = "dot -1" = 0, as the
// pair-term table wants it, iiv_tables.hip: dw_piece_kernel):
//     even bytes: [0, 256) low part, index = bits 0..6 | palette bit << 7;   [256, 384) high part, index = bits 7..13
//     odd bytes:  [384, 512) low part, index = bits 0..6;   [512, 768) high part, index = bits 7..13 | palette bit << 7
constexpr int kHgrDotLutEntries = 768;
__host__ __device__ inline uint32_t hgr_dot_slot_lo(uint32_t m, int odd)
{
    return odd ? 384u + (m & 0x7fu) : ((m & 0x7fu) | ((m >> 3) & 0x80u));
}
__host__ __device__ inline uint32_t hgr_dot_slot_hi(uint32_t m, int odd)
{
    return odd ? 512u + ((m >> 7) | ((m << 4) & 0x80u)) : 256u + (m >> 7);
}
// a window that has slot `t`'s bits and no others (what the table's entry is computed from), and its parity
__host__ __device__ inline uint32_t hgr_dot_slot_window(int t, int &odd)
{
    const uint32_t u = (uint32_t)t;
    odd = t >= 384;
    if (t < 256) return (u & 0x7fu) | ((u & 0x80u) << 3);
    if (t < 384) return (u - 256u) << 7;
    if (t < 512) return u - 384u;
    return (((u - 512u) & 0x7fu) << 7) | (((u - 512u) & 0x80u) >> 4);
}

template <int N>
__device__ inline uint32_t edit_distance(uint64_t alo, uint32_t ahi, uint64_t blo, uint32_t bhi, const uint16_t *lut)
{
    uint32_t e1 = 0, e2 = 0;
    EditStep<N, 0>::run(alo, ahi, blo, bhi, lut, e1, e2);
    return e1;
}


}  // namespace iiv
