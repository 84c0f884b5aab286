// iiv_ingest.hip -- RGB frames -> HGR / DHGR memory maps on gfx950.
//
// The reference shells out to an external C tool, /usr/local/bin/bmp2dhr, for this step
// (transcoder/frame_grabber.py:68-115; README.md:217-221 wishes for a "direct image
// encoding").  That tool is not part of the reference's source and is absent here, so this
// conversion has no reference output to match: it is specified, in integer arithmetic, in
// include/iivision.h (iiv_frames_to_memory_maps), and the tests hold these kernels to a CPU
// restatement of that specification bit for bit.
//
// Round 5: the conversion is now sized for the encoder's rate (3.4 M frames/s: 550 GB/s of RGB in).
//   * The nearest palette colour is an arg-min of LINEAR forms: 2 dr^2 + 4 dg^2 + 3 db^2 =
//     (2 r^2 + 4 g^2 + 3 b^2) + K_c - (4 R_c r + 8 G_c g + 6 B_c b), the first term common to all colours, so
//     key_c = 16 (K_c - 4 R_c r - 8 G_c g - 6 B_c b) + c is three 24-bit multiply-adds per colour and the
//     winner (ties to the lower colour value, as the specification says) one signed minimum.
//   * Ordered dither: one thread per SEVEN colour pixels -- 28 DHGR dots = four screen bytes (aux, main, aux,
//     main), 14 HGR dots = two bytes with everything their palette-bit decisions need -- so no pixel is
//     evaluated twice and no thread exchanges anything; its 42 source bytes are eleven aligned dword loads
//     and a funnel shift, its output two 16-bit stores.  (It was one thread per screen byte: every DHGR pixel
//     evaluated 1.75 times, every HGR pixel twice for both palette bits.)
//   * Error diffusion: lanes are rows, three frames of twenty rows per wave, seven pixels apart, instead of 192
//     threads and a workgroup barrier per pixel step (ingest_diffusion_kernel below): what a row hands to the
//     row below moves to the next lane by DPP, every lane is at the same phase of the 7-pixel / 2- or 4-byte
//     period, so HGR's palette-bit decisions and the stores are uniform and phase-indexed state is static.
//   * No allocation, no synchronisation: the palette's linear forms travel as a kernel argument, the screen
//     holes are zeroed by a kernel on the same stream.
#include "iiv_host.h"
#include "iiv_stream.h"
#include <stdlib.h>
#include <type_traits>

namespace iiv {

__device__ __host__ static inline int y_to_offset(int y)  // y_to_base_addr(y, 0) - 0x2000 (screen.py:16-22)
{
    return 1024 * (y % 8) + 128 * ((y % 64) / 8) + 40 * (y / 64);
}

// The palette as the kernels use it (built on the host per call, passed by value).
struct IngestPalette {
    int32_t k[16];        // 16 (2 R^2 + 4 G^2 + 3 B^2) + colour value
    int32_t a[16], b[16], c[16];   // -16 * 4 R, -16 * 8 G, -16 * 6 B
    uint32_t bc[16];      // the last two as a pair of 16-bit halves (both fit: 128 x 255, 96 x 255 < 2^15): v_dot2c_i32_i16's operand
    uint32_t rgb[16];     // R | G << 8 | B << 16
    int32_t dither[16];   // ordered-dither offset of (y & 3) * 4 + (k & 3): floor((2 Bayer - 15) * amplitude / 16)
};

static IngestPalette make_palette(const uint8_t pal[48], int dither)
{
    static const int bayer[16] = {0, 8, 2, 10, 12, 4, 14, 6, 3, 11, 1, 9, 15, 7, 13, 5};
    IngestPalette p;
    for (int c = 0; c < 16; c++) {
        const int R = pal[3 * c], G = pal[3 * c + 1], B = pal[3 * c + 2];
        p.k[c] = 16 * (2 * R * R + 4 * G * G + 3 * B * B) + c;
        p.a[c] = -64 * R;
        p.b[c] = -128 * G;
        p.c[c] = -96 * B;
        p.bc[c] = (uint32_t)(uint16_t)(int16_t)(-128 * G) | ((uint32_t)(uint16_t)(int16_t)(-96 * B) << 16);
        p.rgb[c] = (uint32_t)R | ((uint32_t)G << 8) | ((uint32_t)B << 16);
        p.dither[c] = dither == IIV_DITHER_DIFFUSION ? 0 : ((2 * bayer[c] - 15) * dither + 16 * 256) / 16 - 256;
    }
    return p;
}

// The sixteen K_c in vector registers: a VOP3 instruction reads ONE scalar register, so mad(r, A_c, K_c) with both constants
// in SGPRs costs a v_mov besides -- with K_c in a VGPR a colour is exactly three v_mad_i32_i24 (the compiler otherwise
// builds it from two multiplies, a multiply-add and a three-operand add plus the moves: 4.5 instructions and 2.6 moves)
struct IngestKv {
    int k[16];
};
__device__ static inline IngestKv ingest_kv(const IngestPalette &P)
{
    IngestKv v;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        v.k[c] = P.k[c];
        asm volatile("" : "+v"(v.k[c]));
    }
    return v;
}

// key of colour c for the pixel (r, g, b) = 16 * (distance - the term common to all colours) + c
//     = (K_c + r * (-64 R_c)) + (g, b) . (-128 G_c, -96 B_c):
// a v_mad_i32_i24 into a fresh register and a v_dot2c_i32_i16 that accumulates in place (16-bit pairs, exact 32-bit sum; g, b
// are clamped to 0..255, the coefficients fit 16 bits) -- two instructions instead of three multiply-adds.  (Both products
// in dot form, (r, g) . (..) + (b, 0) . (..), compile to the accumulate-in-place form too and then need a copy of K_c each:
// three again.)  DHGR: the nearest of the sixteen colours (ties to the lower colour value).
typedef short ingest_v2s __attribute__((ext_vector_type(2)));
__device__ static inline int dist_key(const IngestPalette &P, int kc, int c, int r, ingest_v2s gb)
{
    return __builtin_amdgcn_sdot2(gb, __builtin_bit_cast(ingest_v2s, P.bc[c]), __mul24(r, P.a[c]) + kc, false);
}
__device__ static inline int nearest16(const IngestPalette &P, const IngestKv &kv, int r, int g, int b)
{
    const ingest_v2s gb = __builtin_bit_cast(ingest_v2s, (uint32_t)g | ((uint32_t)b << 16));
    int m = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 16; c++) m = min(m, dist_key(P, kv.k[c], c, r, gb));
    return m & 15;
}
// The same with three v_mad_i32_i24 per colour, for the error-diffusion kernel: a lane's pixels there are one dependent chain,
// and the wait states between a v_dot2c and the minimum that reads it cost that kernel more (1.6 %, same-box A/B) than the
// sixteen instructions save.  Three rounds of sixteen independent multiply-adds with ONE compiler barrier between rounds (the
// barrier keeps the compiler from re-associating a colour's chain into mad + 2 mul + add3; one asm statement per round rather
// than per colour, because the hazard recogniser puts an s_nop behind every inline-asm statement)
__device__ static inline int nearest16_mad(const IngestPalette &P, const IngestKv &kv, int r, int g, int b)
{
    int t[16];
#pragma unroll
    for (int c = 0; c < 16; c++) t[c] = __mul24(r, P.a[c]) + kv.k[c];
    asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]),
             "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
#pragma unroll
    for (int c = 0; c < 16; c++) t[c] = __mul24(g, P.b[c]) + t[c];
    asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]),
             "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
    int m = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 16; c++) m = min(m, __mul24(b, P.c[c]) + t[c]);
    return m & 15;
}

// HGR: for both palette bits, (distance term << 2 | pattern) of the nearest of the four colours the bit allows
// (black 0, violet 3 | blue 6, green 12 | orange 9, white 15; ties to the lower pattern)
__device__ static inline void nearest4x2(const IngestPalette &P, const IngestKv &kv, int r, int g, int b, int &key0, int &key1)
{
    // the six colours HGR can show
    constexpr int col[6] = {0, 3, 12, 15, 6, 9};
    const ingest_v2s gb = __builtin_bit_cast(ingest_v2s, (uint32_t)g | ((uint32_t)b << 16));
    int t[6];
#pragma unroll
    for (int j = 0; j < 6; j++) t[j] = dist_key(P, kv.k[col[j]], col[j], r, gb) >> 4;   // (>> 4: the colour value leaves, the distance term stays exact)
    const int f0 = t[0], f3 = t[1], f12 = t[2], f15 = t[3], f6 = t[4], f9 = t[5];
    const int b0 = f0 * 4, w3 = f15 * 4 + 3;
    key0 = min(min(b0, f3 * 4 + 1), min(f12 * 4 + 2, w3));
    key1 = min(min(b0, f6 * 4 + 1), min(f9 * 4 + 2, w3));
}

__global__ __launch_bounds__(256) void ingest_holes_kernel(int n_banks, uint8_t *__restrict__ main_mem, uint8_t *__restrict__ aux_mem)
{
    // the screen holes of every page (offsets 120..127, 248..255) start as zero, as bmp2dhr's files hold them
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one 8-byte hole each: 64 per bank
    if (i >= (size_t)n_banks * 64) return;
    const size_t bank = i >> 6;
    const int h = (int)(i & 63);
    uint8_t *base = (aux_mem && (bank & 1)) ? aux_mem + (bank >> 1) * 8192 : main_mem + (aux_mem ? bank >> 1 : bank) * 8192;
    *reinterpret_cast<uint2 *>(base + (h >> 1) * 256 + ((h & 1) ? 248 : 120)) = make_uint2(0, 0);
}

// Ordered dither (or none): thread T of a frame owns colour pixels 7 g .. 7 g + 6 of row y, T = 20 y + g -- the
// frame's 161 280 source bytes are 3 840 x 42 contiguous bytes in thread order.
template <int MODE>
__global__ __launch_bounds__(256) void ingest_kernel(int n, const uint8_t *__restrict__ rgb_frames, const IngestPalette P,
                                                     uint8_t *__restrict__ main_mem, uint8_t *__restrict__ aux_mem)
{
    __shared__ int dtab[16];
    if (threadIdx.x < 16) dtab[threadIdx.x] = P.dither[threadIdx.x];
    __syncthreads();
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * 3840) return;
    const size_t f = idx / 3840;
    const int T = (int)(idx - f * 3840), y = T / 20, g = T - 20 * y;
    // 42 bytes at 42 T: aligned dwords from (42 T) & ~3, then a funnel shift by 0 or 16 bits
    const uint8_t *src = rgb_frames + f * (size_t)(192 * 280 * 3) + (size_t)(42 * T);
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(src - ((42 * T) & 2));
    uint32_t w[11];
#pragma unroll
    for (int i = 0; i < 11; i++) w[i] = w32[i];
    const uint32_t sh = ((uint32_t)(42 * T) & 2u) * 8u;
    uint32_t q[11];
#pragma unroll
    for (int i = 0; i < 10; i++) q[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
    q[10] = w[10] >> sh;
    auto byte_at = [&](int nb) -> int { return (int)((q[nb >> 2] >> (8 * (nb & 3))) & 255u); };
    const int drow = (y & 3) * 4;
    const IngestKv kv = ingest_kv(P);
    int colour[7], k0[7], k1[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const int d = dtab[drow + ((7 * g + i) & 3)];
        const int r = min(max(((byte_at(6 * i) + byte_at(6 * i + 3) + 1) >> 1) + d, 0), 255);
        const int gg = min(max(((byte_at(6 * i + 1) + byte_at(6 * i + 4) + 1) >> 1) + d, 0), 255);
        const int b = min(max(((byte_at(6 * i + 2) + byte_at(6 * i + 5) + 1) >> 1) + d, 0), 255);
        if (MODE == kDHGR)
            colour[i] = nearest16(P, kv, r, gg, b);
        else
            nearest4x2(P, kv, r, gg, b, k0[i], k1[i]);
    }
    const size_t out = f * 8192 + (size_t)(y_to_offset(y) + 2 * g);
    if (MODE == kDHGR) {
        // a colour value IS its pixel's aligned dot quad (colours.py:100-134): 28 dots, seven per byte, aux / main alternating
        uint32_t dots = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) dots |= (uint32_t)colour[i] << (4 * i);
        const uint32_t b0 = dots & 0x7fu, b1 = (dots >> 7) & 0x7fu, b2 = (dots >> 14) & 0x7fu, b3 = (dots >> 21) & 0x7fu;
        *reinterpret_cast<uint16_t *>(aux_mem + out) = (uint16_t)(b0 | (b2 << 8));
        *reinterpret_cast<uint16_t *>(main_mem + out) = (uint16_t)(b1 | (b3 << 8));
    } else {
        // byte A = dots 0..6 (pixels 0, 0, 1, 1, 2, 2, 3), byte B = dots 7..13 (pixels 3, 4, 4, 5, 5, 6, 6); per byte the palette
        // bit with the smaller summed nearest-colour error over its seven dots (the term common to all colours cancels)
        // (sums of seven terms below 2^21 in magnitude: int)
        const int eA0 = 2 * ((k0[0] >> 2) + (k0[1] >> 2) + (k0[2] >> 2)) + (k0[3] >> 2);
        const int eA1 = 2 * ((k1[0] >> 2) + (k1[1] >> 2) + (k1[2] >> 2)) + (k1[3] >> 2);
        const int eB0 = (k0[3] >> 2) + 2 * ((k0[4] >> 2) + (k0[5] >> 2) + (k0[6] >> 2));
        const int eB1 = (k1[3] >> 2) + 2 * ((k1[4] >> 2) + (k1[5] >> 2) + (k1[6] >> 2));
        const int pbA = eA1 < eA0 ? 1 : 0, pbB = eB1 < eB0 ? 1 : 0;
        auto pat = [&](int i, int pb) -> uint32_t { return (uint32_t)((pb ? k1[i] : k0[i]) & 3); };
        const uint32_t A = pat(0, pbA) | (pat(1, pbA) << 2) | (pat(2, pbA) << 4) | ((pat(3, pbA) & 1u) << 6) | ((uint32_t)pbA << 7);
        const uint32_t B = (pat(3, pbB) >> 1) | (pat(4, pbB) << 1) | (pat(5, pbB) << 3) | (pat(6, pbB) << 5) | ((uint32_t)pbB << 7);
        *reinterpret_cast<uint16_t *>(main_mem + out) = (uint16_t)(A | (B << 8));
    }
}

// dither == IIV_DITHER_DIFFUSION: Floyd-Steinberg error diffusion (include/iivision.h) without a barrier: lanes are ROWS.
// A pixel needs the errors of its left neighbour (its own lane, a step ago) and of three pixels of the row above (the lane
// above, a few steps ago): what a row hands down, D(j) = e(j - 1) + 5 e(j) + 3 e(j + 1), is final once pixel j + 1 is done and
// moves to the next lane by one wave-wide DPP shift per channel.  Integer sums commute, so the schedule changes nothing: the
// result is the oracle's raster-order definition bit for bit.
// HGR adds a screen byte's palette bit, fixed just before the first pixel whose first dot lies in the byte is quantised, from
// the nearest-colour errors of the three or four pixels that start in the byte -- their values taken with everything the row
// ABOVE sends them.  So a row must stay five pixels behind the row above, and 63 x 5 pixels of skew do not fit a 140-pixel
// row: a wave of 64 rows would idle two thirds of the time.  Instead a wave works on THREE frames, twenty rows of each at a
// time (lane = 20 x frame slot + row in the pass; 4 lanes idle), SEVEN pixels behind the lane above: 19 x 7 = 133 < 140, so a
// lane goes from row r to row r + 20 without waiting, and -- 7 colour pixels = 14 HGR dots = two screen bytes = 28 DHGR dots =
// four screen bytes -- every lane is at the same place of the 7-pixel period at every step: the palette-bit decisions (phases
// 0 and 4) and the rows' stores (phase 6) are uniform control flow, everything indexed by the phase is a static register, a
// lane's source bytes are one 42-byte group per seven steps.  What the row above hands down arrives five pixels ahead of its
// use in a seven-slot queue of registers (slot = pixel mod 7); lane 19 of a frame slot -> lane 0 (row 20 q + 19 -> 20 (q + 1))
// goes through a 16-slot LDS ring.  1 540 steps per three frames.  (Round 3's kernels -- 192 threads per frame on a skewed
// wavefront, a workgroup barrier per pixel step -- ran 0.19 M HGR frames/s; a first one-wave-per-frame DHGR form of this
// round, 64 rows two pixels apart with dynamic phases, 2.9 M; this form 5.9 M HGR.)
constexpr int kDiffWaves = 2;   // waves per block
// Round 6: the source rows come through LDS.  A lane walks its row 42 bytes per seven steps, and sixty lanes of a wave
// walk sixty rows: every load instruction touched sixty cache lines, each line was fetched again for each of the ~3
// groups it holds (a CU's waves keep 150 KB of lines in use, its L1 holds 32), and the kernel ran at half the rate it has
// when every lane reads the same row (4.0 against 7.6 M DHGR frames/s: profiles/r06_diffusion_experiments.txt).  Now a
// lane's row is a stream of 64-byte blocks (aligned relative to the frame, whose size is a multiple of 64): three of
// them -- the one its next group starts in and the two behind it -- are in LDS, filled by LDS-DMA loads
// (global_load_lds_dwordx4: sixteen bytes per lane and instruction straight into LDS, no registers), a block requested
// whole and at once, at least one seven-step iteration before the first group that needs it.  One instruction's
// destination is wave-uniform (base + lane x 16), so the layout is [slot 0..2][16-byte chunk 0..3][lane][16 B] and the
// lanes that fill the same slot in an iteration go together under an execution mask.  12 KiB per wave.
#ifndef IIV_DIFF_STAGE
#define IIV_DIFF_STAGE 1
#endif
// waves per SIMD the error-diffusion kernel's registers are held to: DHGR runs 11 % faster at five (96 VGPRs, six dwords
// spilled) than at the four it gets unasked (110 VGPRs); HGR does not (-1 %); six (80 VGPRs) costs both a quarter
#ifndef IIV_DIFF_OCC
#define IIV_DIFF_OCC(MODE) (IIV_DIFF_STAGE ? 3 : (MODE) == kDHGR ? 5 : 4)   // (staged: 12 KiB of LDS per wave hold it to three anyway)
#endif
template <int MODE>
__global__ __launch_bounds__(64 * kDiffWaves, IIV_DIFF_OCC(MODE)) void ingest_diffusion_kernel(int n, const uint8_t *__restrict__ rgb_frames, const IngestPalette P,
                                                                           uint8_t *__restrict__ main_mem, uint8_t *__restrict__ aux_mem)
{
    __shared__ int ring_s[kDiffWaves][3][16][4];
#if IIV_DIFF_STAGE
    __shared__ __attribute__((aligned(16))) uint32_t stage_s[kDiffWaves][12][64][4];   // [slot x 4 + chunk][lane][16 B]
#endif
    __shared__ uint32_t pal_s[16];    // DHGR: R | G << 8 | B << 16 of the sixteen colour values
    __shared__ uint32_t rgb_s[8];     // R | G << 8 | B << 16 of colour4[pb][pattern]: black, violet | blue, green | orange, white
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 8) {
        constexpr int colour4[8] = {0, 3, 12, 15, 0, 6, 9, 15};
        rgb_s[threadIdx.x] = P.rgb[colour4[threadIdx.x]];
    }
    if (threadIdx.x < 16) pal_s[threadIdx.x] = P.rgb[threadIdx.x];
    __syncthreads();
    const int slot3 = lane / 20, i = lane - 20 * slot3;          // frame slot 0..2 (3: idle lanes 60..63), row in the pass
    const size_t f = ((size_t)blockIdx.x * kDiffWaves + wv) * 3 + (size_t)(slot3 < 3 ? slot3 : 0);
    const bool lane_ok = slot3 < 3 && f < (size_t)n;
    int (*ring)[4] = ring_s[wv][slot3 < 3 ? slot3 : 0];
    const uint8_t *frame = rgb_frames + (lane_ok ? f : 0) * (size_t)(192 * 280 * 3);
    const IngestKv kv = MODE == kDHGR ? ingest_kv(P) : IngestKv{};
    // HGR: the six colours it can show, their K in vector registers (see IngestKv)
    int kv0 = P.k[0], kv3 = P.k[3], kv12 = P.k[12], kv15 = P.k[15], kv6 = P.k[6], kv9 = P.k[9];
    asm volatile("" : "+v"(kv0), "+v"(kv3), "+v"(kv12), "+v"(kv15), "+v"(kv6), "+v"(kv9));
    struct F6 {
        int f0, f3, f12, f15, f6, f9;
    };
    auto f6 = [&](int r, int g, int b) -> F6 {   // the distance terms of the six colours (dist_key)
        const ingest_v2s gb = __builtin_bit_cast(ingest_v2s, (uint32_t)g | ((uint32_t)b << 16));
        return F6{dist_key(P, kv0, 0, r, gb) >> 4, dist_key(P, kv3, 3, r, gb) >> 4, dist_key(P, kv12, 12, r, gb) >> 4,
                  dist_key(P, kv15, 15, r, gb) >> 4, dist_key(P, kv6, 6, r, gb) >> 4, dist_key(P, kv9, 9, r, gb) >> 4};
    };
    // per-lane sequence: group tt = T - i of this lane's rows (20 groups of 7 pixels per row; rows i, 20 + i, ...)
    int Dq[7][3];                         // D from the row above for the pixels of the period, slot = pixel mod 7
#pragma unroll
    for (int j = 0; j < 7; j++) Dq[j][0] = Dq[j][1] = Dq[j][2] = 0;
    int inr = 0, ing = 0, inb = 0;        // what arrived at the end of the previous step: D(pixel + 5)
    int e0r = 0, e0g = 0, e0b = 0;        // error of the previous pixel of the row (0 in front of a row)
    int hr = 0, hg = 0, hb = 0;           // e(k - 2) + 5 e(k - 1)
    uint32_t cur[11], nxt[11];            // the source bytes of the current / next 7-pixel group, funnel-shifted to start at byte 0
    int pbA = 0, pbB = 0;
    uint32_t bytesAB = 0;                 // HGR: the two screen bytes of the group, as they fill; DHGR: its 28 dots
    // source group of sequence position tt (clamped into the frame)
    [[maybe_unused]] auto load_group = [&](int tt, uint32_t (&w)[11]) {
        int v = tt < 0 ? 0 : tt;
        int qq = v / 20, gg = v - 20 * qq;
        int rw = 20 * qq + i;
        if (rw > 191) rw = 191, gg = 0;
#ifdef IIV_EXP_DIFF_SAMEROW
        rw = 0;   // (timing experiment only: every lane reads row 0 -- what the kernel costs without its row-strided reads)
#endif
        const size_t off = (size_t)rw * 840 + (size_t)gg * 42;
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(frame + (off & ~(size_t)3));
        uint32_t raw[11];
#pragma unroll
        for (int j = 0; j < 11; j++) raw[j] = p32[j];
        const uint32_t sh = ((uint32_t)off & 2u) * 8u;
#pragma unroll
        for (int j = 0; j < 10; j++) w[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], sh);
        w[10] = raw[10] >> sh;
    };
    // (eleven dwords from the aligned base never leave the row: a group at an odd halfword is the row's last or has 42 bytes
    // behind it; positions beyond the frame are clamped to row 191's first group)
    // mean of the two source pixels of group pixel GP, three channels in one v_lerp_u8: bytes 6 GP .. 6 GP + 5 of cur
    auto mean_of = [&](auto GPc) -> uint32_t {
        constexpr int GP = decltype(GPc)::value;
        constexpr int oa = 6 * GP, ob = 6 * GP + 3;
        const uint32_t A = (oa & 3) ? __builtin_amdgcn_alignbit(cur[(oa >> 2) + 1], cur[oa >> 2], (oa & 3) * 8) : cur[oa >> 2];
        const uint32_t B = (ob & 3) ? __builtin_amdgcn_alignbit(cur[(ob >> 2) + 1 > 10 ? 10 : (ob >> 2) + 1], cur[ob >> 2], (ob & 3) * 8) : cur[ob >> 2];
        return __builtin_amdgcn_lerp(A, B, 0x01010101u);
    };
    auto value_of = [&](uint32_t m, int ar, int ag, int ab, int &r, int &g, int &b) {   // clamp(mean + floor(acc / 16))
        r = min(max((int)(m & 255u) + (ar >> 4), 0), 255);
        g = min(max((int)((m >> 8) & 255u) + (ag >> 4), 0), 255);
        b = min(max((int)((m >> 16) & 255u) + (ab >> 4), 0), 255);
    };
    int tt = -i;                          // this lane's group number at outer iteration T
    bool active = false;
    int row = i;
    size_t outp = 0;
#if IIV_DIFF_STAGE
    // ---- the staged source stream of this lane (byte offsets are relative to its frame; blocks = 64 bytes)
    // reader: where the group it reads next starts, and the slot of that byte's block; what it has moved past is refilled
    uint32_t (*stage)[64][4] = stage_s[wv];
    const int last_pass_row = 180 + i < 192 ? 180 + i : 160 + i;   // this lane's last row
    int rd_row = i, rd_off = i * 840, rd_slot = 0;
    // fetcher: the next block to request -- its offset, how many blocks of its row are still to come (itself included),
    // its row, its slot
    auto blocks_of_row = [](int rw) -> int { return ((rw * 840 + 839) >> 6) - ((rw * 840) >> 6) + 1; };
    int fe_row = i, fe_off = (i * 840) & ~63, fe_left = blocks_of_row(i), fe_slot = 0, fe_need = 3;
    auto fetch_round = [&]() {
        // every lane that still needs a block requests its next one: the lanes aiming at the same slot together
#pragma unroll
        for (int sl = 0; sl < 3; sl++) {
            const bool go = fe_need > 0 && fe_slot == sl;
            if (__ballot(go) == 0ull) continue;
            if (go) {
                const uint8_t *src = frame + fe_off;
#pragma unroll
                for (int c = 0; c < 4; c++) __builtin_amdgcn_global_load_lds(src + 16 * c, &stage[sl * 4 + c][0][0], 16, 0, 0);
                fe_need--;
                fe_slot = fe_slot == 2 ? 0 : fe_slot + 1;
                if (--fe_left == 0) {
                    fe_row = fe_row + 20 <= last_pass_row ? fe_row + 20 : fe_row;   // (behind its last row a lane reads that row again: never used)
                    fe_off = (fe_row * 840) & ~63;
                    fe_left = blocks_of_row(fe_row);
                } else {
                    fe_off += 64;
                }
            }
        }
    };
    // the eleven dwords of the group at rd_off, funnel-shifted to start at its first byte: four 16-byte chunks from LDS,
    // a window of twelve of their sixteen dwords chosen by the start's dword inside its chunk
    auto read_group = [&](uint32_t (&w)[11]) {
        const int o = rd_off & 63;                        // the start inside its block (even)
        const int nx_slot = rd_slot == 2 ? 0 : rd_slot + 1;
        uint32_t D[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = (o & ~15) + 16 * k;             // chunk k of the window: byte offset from the block's start (< 128)
            const int sl = t < 64 ? rd_slot : nx_slot;
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage[sl * 4 + ((t >> 4) & 3)][lane][0]);
            D[4 * k] = v.x, D[4 * k + 1] = v.y, D[4 * k + 2] = v.z, D[4 * k + 3] = v.w;
        }
        // (bitwise selects on opaque masks: written as `sd & 2 ? D[j + 2] : D[j]` the compiler stores the sixteen dwords to
        // SCRATCH and loads the window back at a dynamic offset)
        uint32_t m2 = 0u - (((uint32_t)o >> 3) & 1u), m1 = 0u - (((uint32_t)o >> 2) & 1u);
        asm volatile("" : "+v"(m2), "+v"(m1));
        uint32_t E[14], F[12];
#pragma unroll
        for (int j = 0; j < 14; j++) E[j] = (D[j + 2] & m2) | (D[j] & ~m2);   // v_bfi_b32
#pragma unroll
        for (int j = 0; j < 12; j++) F[j] = (E[j + 1] & m1) | (E[j] & ~m1);
        const uint32_t sh = ((uint32_t)o & 2u) * 8u;
#pragma unroll
        for (int j = 0; j < 10; j++) w[j] = __builtin_amdgcn_alignbit(F[j + 1], F[j], sh);
        w[10] = F[10] >> sh;
    };
    // the reader moves on to group g (>= 1) of this lane's sequence; the blocks it leaves behind become requests
    auto advance_reader = [&](int g) {
        if (g < 1) return;                                // (in front of its first group a lane reads group 0 again and again)
        const int gg = g % 20, rw = 20 * (g / 20) + i;
        if (rw > last_pass_row) return;                   // (behind its last row: it stays where it is; nothing of it is used)
        const int old_blk = rd_off >> 6;
        int d;
        if (gg == 0) {                                    // a new row: the rest of the old row's blocks, then the new row's first
            const int old_last = (rd_row * 840 + 839) >> 6;
            d = old_last - old_blk + 1;
            rd_row = rw;
            rd_off = rw * 840;
        } else {
            rd_off += 42;
            d = (rd_off >> 6) - old_blk;
        }
        rd_slot = rd_slot + d;
        rd_slot = rd_slot >= 3 ? rd_slot - 3 : rd_slot;
        fe_need += d;
    };
    auto staged_wait = []() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // (the order below keeps a block's request one whole iteration in front of its first read: a group is read where the
    // reader STANDS, then the reader moves on to the next group and what it left behind is requested -- at a row's end that
    // can be two blocks at once, and the next row's second block is then among the new requests)
    fetch_round();
    fetch_round();
    fetch_round();
    staged_wait();
    read_group(cur);
    advance_reader(tt + 1);
    fetch_round();
    fetch_round();
    staged_wait();
    read_group(nxt);
    advance_reader(tt + 2);
    fetch_round();
    fetch_round();
#else
    load_group(tt, cur);
    load_group(tt + 1, nxt);
#endif
    auto step = [&](auto Pc) {
        constexpr int PH = decltype(Pc)::value;
        // what arrived: D(pixel + 5) of the row above, into its slot; lane 0 of a frame slot takes it from the ring
        {
            // (every lane reads the ring -- a branch per step costs more than three LDS words; only lanes 0 / 20 / 40 keep them.
            // Round 6, measured and removed: the ring word requested here but taken at the start of the NEXT step -- its slot
            // of the queue is first looked at two steps later -- and no fence per step: 4.94 against 4.97 M frames/s, same box)
            const int s = 7 * tt + PH - 135;         // lane 19's sequence index of that D
            const int *slot = ring[s & 15];
            int s0 = slot[0], s1 = slot[1], s2 = slot[2];
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2));   // (read by every lane: kept out of an exec region each)
            const bool from_ring = i == 0, has = s >= 0;
            const int ar = from_ring ? (has ? s0 : 0) : inr;
            const int ag = from_ring ? (has ? s1 : 0) : ing;
            const int ab = from_ring ? (has ? s2 : 0) : inb;
            Dq[(PH + 5) % 7][0] = ar, Dq[(PH + 5) % 7][1] = ag, Dq[(PH + 5) % 7][2] = ab;
        }
        int er = 0, eg = 0, eb = 0;
        const bool first = PH == 0 && (tt % 20 == 0 || !active);    // (inactive lanes: only the flush of their last row matters)
        // to the row below: D(j) = e(j - 1) + 5 e(j) + 3 e(j + 1) for j = this pixel - 1 (j = 139 of the finished row at a row's
        // first pixel and at the flush behind a lane's last row; an idle lane hands down its h: 0, or the flush)
        int outr = hr, outg = hg, outb = hb;
        if (active) {
            int r, g, b;
            // 7 e + D as (e << 3) + (D - e): two full-rate instructions (a plain 7 * e + D becomes the quarter-rate v_mad_u64_u32)
            int tr = Dq[PH][0] - e0r, tg = Dq[PH][1] - e0g, tb = Dq[PH][2] - e0b;
            asm("" : "+v"(tr), "+v"(tg), "+v"(tb));
            value_of(mean_of(Pc), (e0r << 3) + tr, (e0g << 3) + tg, (e0b << 3) + tb, r, g, b);
            if constexpr (MODE == kDHGR) {
                // the nearest of the sixteen colours; its value IS the pixel's dot quad
                const int col = nearest16_mad(P, kv, r, g, b);
                const uint32_t prgb = pal_s[col];
                er = r - (int)(prgb & 255u);
                eg = g - (int)((prgb >> 8) & 255u);
                eb = b - (int)((prgb >> 16) & 255u);
                bytesAB |= (uint32_t)col << (4 * PH);
                if (PH == 6) {
                    const uint32_t b0 = bytesAB & 0x7fu, b1 = (bytesAB >> 7) & 0x7fu, b2 = (bytesAB >> 14) & 0x7fu, b3 = (bytesAB >> 21) & 0x7fu;
                    *reinterpret_cast<uint16_t *>(aux_mem + outp) = (uint16_t)(b0 | (b2 << 8));
                    *reinterpret_cast<uint16_t *>(main_mem + outp) = (uint16_t)(b1 | (b3 << 8));
                    outp += 2;
                    bytesAB = 0;
                }
            } else {
            const F6 fk = f6(r, g, b);
            if (PH == 0 || PH == 4) {
                // the palette bit of the byte this pixel opens: summed nearest-colour errors of the pixels that start in it
                // (weights: their dots in the byte; the term common to all colours cancels), ties to 0
                int s0 = 2 * min(min(fk.f0, fk.f3), min(fk.f12, fk.f15)), s1 = 2 * min(min(fk.f0, fk.f6), min(fk.f9, fk.f15));
                auto ahead = [&](auto Qc, int w) {
                    constexpr int Q = decltype(Qc)::value;
                    int r2, g2, b2;
                    value_of(mean_of(Qc), Dq[Q][0], Dq[Q][1], Dq[Q][2], r2, g2, b2);
                    const F6 fa = f6(r2, g2, b2);
                    s0 += __mul24(w, min(min(fa.f0, fa.f3), min(fa.f12, fa.f15)));   // (|f| < 2^21)
                    s1 += __mul24(w, min(min(fa.f0, fa.f6), min(fa.f9, fa.f15)));
                };
                if (PH == 0) {
                    ahead(std::integral_constant<int, 1>{}, 2);
                    ahead(std::integral_constant<int, 2>{}, 2);
                    ahead(std::integral_constant<int, 3>{}, 1);
                    pbA = s1 < s0 ? 1 : 0;
                } else {
                    ahead(std::integral_constant<int, 5>{}, 2);
                    ahead(std::integral_constant<int, 6>{}, 2);
                    pbB = s1 < s0 ? 1 : 0;
                }
            }
            const int pb = PH < 4 ? pbA : pbB;
            const int k0 = min(min(fk.f0 * 4, fk.f3 * 4 + 1), min(fk.f12 * 4 + 2, fk.f15 * 4 + 3));
            const int k1 = min(min(fk.f0 * 4, fk.f6 * 4 + 1), min(fk.f9 * 4 + 2, fk.f15 * 4 + 3));
            const uint32_t pat = (uint32_t)(pb ? k1 : k0) & 3u;
            const uint32_t prgb = rgb_s[pb * 4 + (int)pat];
            er = r - (int)(prgb & 255u);
            eg = g - (int)((prgb >> 8) & 255u);
            eb = b - (int)((prgb >> 16) & 255u);
            // dots 2 PH, 2 PH + 1 of the group: byte A = dots 0..6 | palette bit, byte B = dots 7..13 | palette bit
            if (PH < 3) bytesAB |= pat << (2 * PH);
            if (PH == 3) bytesAB |= ((pat & 1u) << 6) | ((uint32_t)pbA << 7) | ((pat >> 1) << 8);
            if (PH > 3) bytesAB |= pat << (2 * PH + 1);       // PH 4, 5, 6 -> bits 9, 11, 13 (byte B bits 1, 3, 5)
            if (PH == 6) {
                *reinterpret_cast<uint16_t *>(main_mem + outp) = (uint16_t)(bytesAB | ((uint32_t)pbB << 15));
                outp += 2;
                bytesAB = 0;
            }
            }   // HGR
            if (!first) outr += __mul24(3, er), outg += __mul24(3, eg), outb += __mul24(3, eb);
            hr = e0r + __mul24(5, er), hg = e0g + __mul24(5, eg), hb = e0b + __mul24(5, eb);
            e0r = er, e0g = eg, e0b = eb;
        } else if (PH == 0) {
            hr = hg = hb = 0;     // (flushed: out holds it)
        }
        {
            const int s = 7 * tt + PH - 1;            // this lane's sequence index of the emitted D
            if (i == 19 && s >= 0) {
                int *slot = ring[s & 15];
                slot[0] = outr, slot[1] = outg, slot[2] = outb;
            }
        }
        wave_lds_sync();
        inr = __builtin_amdgcn_update_dpp(0, outr, 0x138, 0xf, 0xf, false);   // wave_shr:1
        ing = __builtin_amdgcn_update_dpp(0, outg, 0x138, 0xf, 0xf, false);
        inb = __builtin_amdgcn_update_dpp(0, outb, 0x138, 0xf, 0xf, false);
    };
    // lane i starts at T = i and runs 200 groups (pass 9: rows 180 + i < 192 only); one more iteration flushes the last rows
    for (int T = 0; T < 19 + 200 + 1; T++) {
        const int qq = tt >= 0 ? tt / 20 : 0;
        row = 20 * qq + i;
        active = lane_ok && tt >= 0 && row < 192;
        if (tt >= 0 && tt % 20 == 0) {
            // a new row: nothing from the left; where its bytes go
            e0r = e0g = e0b = 0;
            outp = f * 8192 + (size_t)y_to_offset(row < 192 ? row : 191);
        }
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
#if IIV_DIFF_STAGE
        // what was requested at the end of the previous iteration has landed by now (six steps later).  HERE, not at the read
        // below: behind step 6 the wait would also cover the row's stores that step has just issued
        staged_wait();
#endif
        step(std::integral_constant<int, 6>{});
#pragma unroll
        for (int j = 0; j < 11; j++) cur[j] = nxt[j];
        tt++;
#if IIV_DIFF_STAGE
        read_group(nxt);          // group tt + 1: where the reader stands (its blocks: waited for in front of step 6)
        advance_reader(tt + 2);
        fetch_round();            // the blocks left behind: at most two per lane (a row's end)
        if (__ballot(fe_need > 0) != 0ull) fetch_round();
#else
        load_group(tt + 1, nxt);
#endif
    }
}

int frames_to_memory_maps(int mode, const uint8_t palette_rgb[48], int n, const uint8_t *d_rgb, int dither, uint8_t *d_main,
                          uint8_t *d_aux, hipStream_t st)
{
    const IngestPalette P = make_palette(palette_rgb, dither);
    const int n_banks = mode == kDHGR ? 2 * n : n;
    hipLaunchKernelGGL(ingest_holes_kernel, dim3((unsigned)(((size_t)n_banks * 64 + 255) / 256)), dim3(256), 0, st, n_banks, d_main,
                       mode == kDHGR ? d_aux : (uint8_t *)nullptr);
    int rc = hip_check(hipGetLastError(), "ingest_holes_kernel launch");
    if (rc) return rc;
    if (dither == IIV_DITHER_DIFFUSION) {
        const dim3 grid((unsigned)((n + 3 * kDiffWaves - 1) / (3 * kDiffWaves)));
        // (IIV_EXP_DIFF_LDS_PAD: timing experiments only -- extra dynamic LDS per workgroup caps the waves resident per CU)
        static const int lds_pad = getenv("IIV_EXP_DIFF_LDS_PAD") ? atoi(getenv("IIV_EXP_DIFF_LDS_PAD")) : 0;
        if (mode == kDHGR)
            hipLaunchKernelGGL(ingest_diffusion_kernel<kDHGR>, grid, dim3(64 * kDiffWaves), (size_t)lds_pad, st, n, d_rgb, P, d_main, d_aux);
        else
            hipLaunchKernelGGL(ingest_diffusion_kernel<kHGR>, grid, dim3(64 * kDiffWaves), (size_t)lds_pad, st, n, d_rgb, P, d_main, d_aux);
        return hip_check(hipGetLastError(), "ingest diffusion kernel launch");
    }
    const size_t total = (size_t)n * 3840;
    dim3 grid((unsigned)((total + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(ingest_kernel<kDHGR>, grid, dim3(256), 0, st, n, d_rgb, P, d_main, d_aux);
    else
        hipLaunchKernelGGL(ingest_kernel<kHGR>, grid, dim3(256), 0, st, n, d_rgb, P, d_main, d_aux);
    return hip_check(hipGetLastError(), "ingest_kernel launch");
}

}  // namespace iiv

extern "C" int iiv_frames_to_memory_maps(int mode, const uint8_t palette_rgb[48], int n_frames, const uint8_t *d_rgb,
                                         int dither, uint8_t *d_main, uint8_t *d_aux, void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || !palette_rgb || n_frames < 0 || !d_rgb || !d_main ||
        (mode == IIV_DHGR && !d_aux) || dither < 0 || dither > IIV_DITHER_DIFFUSION)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_frames_to_memory_maps: bad argument");
    if (((uintptr_t)d_rgb & 3) || ((uintptr_t)d_main & 7) || ((uintptr_t)d_aux & 7))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_frames_to_memory_maps: d_rgb must be 4-byte aligned, d_main / d_aux 8-byte aligned");
    if (n_frames == 0) return IIV_OK;
    return iiv::frames_to_memory_maps(mode, palette_rgb, n_frames, d_rgb, dither, d_main, d_aux, (hipStream_t)stream);
}
