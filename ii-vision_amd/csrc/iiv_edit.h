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

// The same recurrence for strings delivered one pixel per byte and already paired:
// byte j of xs[g] = a << 4 | b of pixel 4g + j (exactly the index into lut), so a step
// needs one byte extract for the lookup, and its transposition test -- a_{k-1} == b_k and
// a_k == b_{k-1} -- is "this byte equals the previous byte with its nibbles swapped".
// G = number of 32-bit groups, LAST = pixels in the last group.
template <int K, int N> struct EditStepBytes {
    template <typename X>
    __device__ static inline void run(const X &xs, const X &ys, const uint16_t *lut, uint32_t &e1, uint32_t &e2)
    {
        const uint32_t idx = (xs[K >> 2] >> (8 * (K & 3))) & 0xffu;
        uint32_t e = e1 + (uint32_t)lut[idx];
        if (K >= 1) {
            const uint32_t prev_swapped = (ys[(K - 1) >> 2] >> (8 * ((K - 1) & 3))) & 0xffu;
            const uint32_t t = e2 + 1;
            if (idx == prev_swapped && t < e) e = t;
        }
        e2 = e1;
        e1 = e;
        EditStepBytes<K + 1, N>::run(xs, ys, lut, e1, e2);
    }
};
template <int N> struct EditStepBytes<N, N> {
    template <typename X>
    __device__ static inline void run(const X &, const X &, const uint16_t *, uint32_t &, uint32_t &) {}
};

// pixel p of the string sits in byte p % 4 of group p / 4 (groups may be partly filled: the
// caller lays the pixels out so that N consecutive byte slots are used)
template <int N, int G>
__device__ inline uint32_t edit_distance_bytes(const uint32_t (&src)[G], const uint32_t (&tgt)[G], const uint16_t *lut)
{
    uint32_t xs[G], ys[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        xs[g] = (src[g] << 4) | tgt[g];
        ys[g] = (tgt[g] << 4) | src[g];  // xs with the nibbles of every byte swapped (one v_lshl_or each)
    }
    uint32_t e1 = 0, e2 = 0;
    EditStepBytes<0, N>::run(xs, ys, lut, e1, e2);
    return e1;
}

// Two distances at once in packed 16-bit arithmetic (every value of the recurrence is <= 18 x 113 < 2^11): the
// two strings' steps share the v_pk_add / v_pk_min, only the two cost lookups stay separate.  The transposition
// test becomes arithmetic: x = (this byte) ^ (previous byte with its nibbles swapped) is 0 exactly where the
// transposition exists, and its cost is then 1 + 0x3fff * min(x, 1) -- a value no path through a missing
// transposition can undercut (e2 + 0x4000 > e1 + any substitution cost, and nothing overflows 16 bits).
typedef unsigned short edit_u16x2 __attribute__((ext_vector_type(2)));
// byte KB of string a in bits 0..7, byte KB of string b in bits 16..23 (v_perm_b32: bytes 0..3 of the second
// operand are selectors 0..3, of the first 4..7; 0x0c = the constant 0)
template <int KB> __device__ static inline uint32_t edit_pair_bytes(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_perm(b, a, 0x0c000c00u | (uint32_t)KB | ((uint32_t)(4 + KB) << 16));
}
template <int K, int N> struct EditStepPair {
    template <typename X>
    __device__ static inline void run(const X &xa, const X &ya, const X &xb, const X &yb, const uint16_t *lut, uint32_t prev_swapped,
                                      edit_u16x2 &e1, edit_u16x2 &e2)
    {
        const uint32_t idx = edit_pair_bytes<(K & 3)>(xa[K >> 2], xb[K >> 2]);   // the two lookup indices, one per half
        const uint32_t s = (uint32_t)lut[idx & 0xffu] | ((uint32_t)lut[idx >> 16] << 16);
        edit_u16x2 e = e1 + __builtin_bit_cast(edit_u16x2, s);
        if (K >= 1) {
            // (inline assembly: written as min / multiply-add the compiler turns it into two compares, two
            // selects and a repack)
            uint32_t tc;
            asm("v_pk_min_u16 %0, %1, %2\n\tv_pk_mad_u16 %0, %0, %3, %2" : "=&v"(tc) : "v"(idx ^ prev_swapped), "v"(0x00010001u), "v"(0x3fff3fffu));
            e = __builtin_elementwise_min(e, e2 + __builtin_bit_cast(edit_u16x2, tc));
        }
        e2 = e1;
        e1 = e;
        EditStepPair<K + 1, N>::run(xa, ya, xb, yb, lut, K + 1 < N ? edit_pair_bytes<(K & 3)>(ya[K >> 2], yb[K >> 2]) : 0u, e1, e2);
    }
};
template <int N> struct EditStepPair<N, N> {
    template <typename X>
    __device__ static inline void run(const X &, const X &, const X &, const X &, const uint16_t *, uint32_t, edit_u16x2 &, edit_u16x2 &) {}
};
template <int N, int G>
__device__ inline void edit_distance_bytes_pair(const uint32_t (&src_a)[G], const uint32_t (&tgt_a)[G], const uint32_t (&src_b)[G],
                                                const uint32_t (&tgt_b)[G], const uint16_t *lut, uint32_t &d_a, uint32_t &d_b)
{
    uint32_t xa[G], ya[G], xb[G], yb[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        xa[g] = (src_a[g] << 4) | tgt_a[g];
        ya[g] = (tgt_a[g] << 4) | src_a[g];
        xb[g] = (src_b[g] << 4) | tgt_b[g];
        yb[g] = (tgt_b[g] << 4) | src_b[g];
    }
    edit_u16x2 e1 = {0, 0}, e2 = {0, 0};
    EditStepPair<0, N>::run(xa, ya, xb, yb, lut, 0u, e1, e2);
    d_a = e1.x;
    d_b = e1.y;
}

// HGR colour strings from three small lookups: pixels 0..5, 6..11 and 12..17 of a window's string
// depend on only 7, 6 and 7 of its 14 bits (found by brute force over all 2 x 2^14 windows: flip a
// bit, see which pixels can change -- the palette bit of the byte itself reaches every pixel of
// its group, a neighbour's only the pixels next to it):
//     even byte:  bits 0..5 and 10 | bits 4..8 and 10 | bits 7..13
//     odd byte:   bits 0..6        | bits 3 and 5..9  | bits 3 and 8..13
// hgr_group_index compresses a window to the index of group g; hgr_group_window is its inverse
// with every other bit zero (a representative window for building the table).
constexpr int kHgrGroupEntries = 320;  // 128 + 64 + 128 per parity
__host__ __device__ constexpr int hgr_group_base(int g) { return g == 0 ? 0 : g == 1 ? 128 : 192; }
__host__ __device__ inline uint32_t hgr_group_index(uint32_t m, int g, int odd)
{
    if (!odd) {
        if (g == 0) return (m & 0x3fu) | ((m >> 4) & 0x40u);
        if (g == 1) return ((m >> 4) & 0x1fu) | ((m >> 5) & 0x20u);
        return m >> 7;
    }
    if (g == 0) return m & 0x7fu;
    if (g == 1) return ((m >> 3) & 1u) | ((m >> 4) & 0x3eu);
    return ((m >> 3) & 1u) | ((m >> 7) & 0x7eu);
}
__host__ __device__ inline uint32_t hgr_group_window(uint32_t idx, int g, int odd)
{
    if (!odd) {
        if (g == 0) return (idx & 0x3fu) | ((idx & 0x40u) << 4);
        if (g == 1) return ((idx & 0x1fu) << 4) | ((idx & 0x20u) << 5);
        return idx << 7;
    }
    if (g == 0) return idx;
    if (g == 1) return ((idx & 1u) << 3) | ((idx & 0x3eu) << 4);
    return ((idx & 1u) << 3) | ((idx & 0x7eu) << 7);
}

template <int N>
__device__ inline uint32_t edit_distance(uint64_t alo, uint32_t ahi, uint64_t blo, uint32_t bhi, const uint16_t *lut)
{
    uint32_t e1 = 0, e2 = 0;
    EditStep<N, 0>::run(alo, ahi, blo, bhi, lut, e1, e2);
    return e1;
}


}  // namespace iiv
