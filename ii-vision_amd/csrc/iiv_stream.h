// iiv_stream.h -- per-stream encoder state in HBM and the formats shared by the
// prologue (iiv_prologue.hip), the greedy kernels (iiv_greedy.hip, iiv_team.hip, iiv_workgroup.hip) and the
// split-table builder (iiv_tables.hip).
// Reference: transcoder/video.py:16-301 (Video state), transcoder/screen.py:383-547.
#pragma once

#include "iiv_device.h"
#include "iiv_edit.h"

#include <type_traits>

namespace iiv {

constexpr int kPushedCap = 24576;  // >= 3 pushes (two, or three with the fourth-offset option) x 7680 non-hole bytes

// ---- wd[]: one word per byte of the live generator's bank, written once by the prologue
// and immutable while the generator lives:
//   bits  0..11  MINUS the diff weight of the byte (Bitmap.diff_weights, screen.py:400-449; <= 2047), modulo 4096
//   bits 12..21  byte offset of the byte's u16 inside a slice of the RIGHT half of the split store
//                table: split_row_right << 1  (bit 12 is 0)
//   bits 22..31  the same for the LEFT half: split_row_left << 1  (bit 22 is 0)
// A greedy step needs three things of a byte, one instruction each: the left offset (a shift), the
// right offset (a bit-field extract), and `(wd << 20) + offset constant` = -diff weight << 20 | y,
// the addend of the key `value * (2^20 + 2^8) + ...` = delta << 20 | value << 8 | y with
// delta = store value - diff weight (screen.py:547) in [-2047, 2047].
constexpr int kWdDwShift = 20;     // where a key carries its delta
constexpr int kWdDwBits = 12;
constexpr int kWdRightShift = 12, kWdLeftShift = 22;
__host__ __device__ inline uint32_t wd_word(uint32_t row_left, uint32_t row_right, uint32_t dw)
{
    return (row_left << (kWdLeftShift + 1)) | (row_right << (kWdRightShift + 1)) | ((0u - dw) & ((1u << kWdDwBits) - 1u));
}
__host__ __device__ inline uint32_t wd_dw(uint32_t wd) { return (0u - wd) & ((1u << kWdDwBits) - 1u); }   // the diff weight back
__host__ __device__ inline uint32_t wd_off_left(uint32_t wd) { return wd >> kWdLeftShift; }
__host__ __device__ inline uint32_t wd_off_right(uint32_t wd) { return (wd >> kWdRightShift) & 0x3ffu; }
constexpr uint32_t kDwPieceBias = 512;   // added to every entry of the prologue's pair-term table (its terms can be negative)
// An HGR diff weight as the sum of its nine pair terms (iiv_tables.hip: dw_piece_kernel): the two windows as dots (two
// lookups each, iiv_edit.h: hgr_dot_slot_lo), then per pixel pair i the six dots 2i-1 .. 2i+4 of both as the index
// cur6 * kHgrPieceStride + tgt6 into G, whose word holds the pair's two rotation classes (even bytes: class i & 1; odd
// bytes, whose phase differs by 2, the other one).  `dots` / `g` = the two tables wherever the caller keeps them (the
// prologue: LDS; the exhaustive check, iiv_tables.hip: dw_piece_check_hgr_kernel: HBM) -- one function, so that what is
// checked against every entry of the full table is what the prologue runs.
// Why the stride is 67 and not 64: HGR's dots come in equal pairs (HGRBitmap._double_pixels, screen.py:712-739), so six
// consecutive dots take 8 or 16 of their 64 values, and with rows of 64 words the LDS bank of a lookup -- word index mod
// 32 -- is five of the TARGET's dots alone: 32 lanes fell on 8 banks, 5.6 LDS cycles per 32-lane group instead of the
// 3.5 of a uniformly random gather (measured: SQ_LDS_BANK_CONFLICT 815 cycles per wave, 4 per LDS instruction).  With
// rows of 67 words the bank is (3 x cur6 + tgt6) mod 32, which spreads the doubled patterns of both windows over all
// banks (3.5 cycles, simulated over random screens for every stride from 64 to 95); the multiply-add that forms the
// address replaces a shift-or, so it costs no instruction.
constexpr uint32_t kHgrPieceStride = 67;
constexpr uint32_t kHgrPieceWords = 4288;   // >= 63 * 67 + 64, a multiple of four (copied to LDS 16 B per lane)
__device__ __forceinline__ uint32_t hgr_dw_pieces_sum(const uint32_t *__restrict__ dots, const unsigned char *__restrict__ g, uint32_t cm, uint32_t tm, int odd)
{
    const uint32_t Dc = dots[hgr_dot_slot_lo(cm, odd)] | dots[hgr_dot_slot_hi(cm, odd)];
    const uint32_t Dt = dots[hgr_dot_slot_lo(tm, odd)] | dots[hgr_dot_slot_hi(tm, odd)];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t cur6 = __builtin_amdgcn_ubfe(Dc, 2 * i, 6);
        const uint32_t tgt6x4 = (i == 0 ? Dt << 2 : i == 1 ? Dt : Dt >> (2 * i - 2)) & 0xfcu;
        acc += *reinterpret_cast<const uint16_t *>(g + (__umul24(cur6, 4u * kHgrPieceStride) + tgt6x4) + 2 * ((i & 1) ^ odd));
    }
    return acc - 9u * kDwPieceBias;
}
constexpr int kMaxValue = 2047;  // every table value and diff weight must fit 11 bits

struct StreamState {
    uint8_t mem[2][8192];     // [is_aux] Video.memory_map / aux_memory_map
    int32_t up[2][8192];      // [is_aux] Video.update_priority / aux_update_priority -- as the HOST sees it; the kernels' copy is up16 (below)
    uint32_t wd[8192];        // live generator: see above
    uint32_t order[8192];     // sorted initial entries: page << 8 | offset | target content << 16
    uint32_t nzbits[256];     // bit = update_priority != 0 (as left by the last launch)
    uint32_t pdone[256];      // bit = byte was a primary in the live generator (its diff weight is 0)
    uint32_t pushed[kPushedCap];  // (2047-p) << 21 | nonce << 13 | page << 8 | offset; ~0 = popped
    uint32_t mt_py[624];      // random's MT19937 block
    uint32_t mt_np[624];      // np.random's MT19937 block
    int32_t mt_py_idx, mt_np_idx;
    int32_t n_sorted, head, n_pushed, exhausted;
    int32_t gen_active, gen_is_aux, gen_frame, error;
    int32_t out_of_work[2];
    int32_t pad_content;      // target[0,0] of the live generator's bank (video.py:249)
    int32_t truncated;        // order[] holds only the top of the list (prefix sort)
    unsigned long long draws_py, draws_np, ops, pad_ops;
    // what the one-wave kernel saw of the input (iiv_encode.hip: kTieHeavyPercent), totals since creation: steps the nonces
    // decided, real (not padding) opcodes, launches that emitted something
    unsigned long long stat_exact, stat_ops, stat_runs;
    unsigned long long stamps[32];  // diagnostic builds only (-DIIV_STAMPS): prologue s_memtime stamps [0,16), greedy phase cycles [16,24)
    // What the kernels read and write of the priorities: 16 bits per byte, kUpBig = "the value is in up[]" (negative, or
    // >= 65535: priorities grow by a diff weight per call while a byte waits).  The prologue adds a diff weight to EVERY
    // priority in EVERY call (video.py:115-116): as int32 that is 64 of the ~142 KB a call moves through HBM, and the call is
    // within 15 % of what this box's HBM copies.  up[] is exact wherever up16 says kUpBig and is brought up to
    // date everywhere before the host looks (iiv_encode.hip: materialise_up_kernel / compact_up_kernel).
    uint16_t up16[2][8192];
};
constexpr uint32_t kUpBig = 0xffffu;
__device__ inline int32_t up_value(const StreamState &S, int b, int i)
{
    const uint32_t v = S.up16[b][i];
    return v == kUpBig ? S.up[b][i] : (int32_t)v;
}

enum { kErrNone = 0, kErrHoles = 1, kErrNegative = 2, kErrPaletteBit = 3, kErrPushedOverflow = 4, kErrNoGenerator = 5, kErrGuard = 6, kErrSortBudget = 7, kErrBankMix = 8 };

// What one stream does in one launch round (iiv_encode builds these from the segment lists).
// Shared schedule: every stream reads the same descriptor (stride 0); per-stream schedules:
// descriptor [round][stream].
struct LaunchSeg {
    int32_t frame;     // target frame index
    int32_t is_aux;    // bank
    int32_t n_ops;     // opcodes to emit in this round (0: the stream idles)
    int32_t ops_base;  // opcode index of this round's first opcode in the stream's output
    int32_t need;      // prologue: -1 = no new generator in this round; 0 = order everything;
                       // > 0 = order at least the `need` highest priorities (prefix sort)
};

// ---- the split store table -------------------------------------------------------------
// The store value S[o][content][window] = ED(string(poke(window, content)), string(window))
// (Bitmap.compute_delta_page / byte_pair_difference, screen.py:383-398, 525-547) is a
// min-plus chain over the pixels of the two colour strings (E[k] = min(E[k-1] + sub_k,
// E[k-2] + transposition_k), make_data_tables.py:92-108).  Cut after pixel M: the state
// (E[M-1], E[M]) depends only on the window bits that reach pixels 1..M, and the cost of
// finishing from either state component only on the bits that reach pixels M..N, so
//     S = min(l0 + r0, l1 + r1)
// with (l0, l1) from a LEFT table and (r0, r1) from a RIGHT table whose (content, row) slices
// are 1-2 KiB instead of 11-32 KiB: the 256 lookups of a greedy step touch 48-64 cache
// lines instead of ~135, and the whole table is 0.5 MiB instead of 8-16 MiB.
// Exactness for every (offset, content, window) is tested against iiv_build_store_table.
//   DHGR (10 pixels, M = 5): left bits 0..7 of the window, right bits 4..12
//   HGR  (18 pixels, M = 9): even byte: left bits 0..7 and 10, right bits 5..13
//                            odd byte:  left bits 0..8,        right bits 3 and 6..13
template <int MODE> struct SplitTraits;
template <> struct SplitTraits<kDHGR> {
    static constexpr int kCut = 5, kLeftRowBits = 8, kRightRowBits = 9, kLeftCBits = 5, kRightCBits = 6;
};
template <> struct SplitTraits<kHGR> {
    static constexpr int kCut = 9, kLeftRowBits = 9, kRightRowBits = 9, kLeftCBits = 6, kRightCBits = 6;
};

// window bits that feed each half (o = byte offset inside the packed column)
template <int MODE> __host__ __device__ constexpr uint32_t split_mask_left(int o)
{
    return MODE == kDHGR ? 0x00ffu : ((o & 1) ? 0x01ffu : 0x04ffu);
}
template <int MODE> __host__ __device__ constexpr uint32_t split_mask_right(int o)
{
    return MODE == kDHGR ? 0x1ff0u : ((o & 1) ? 0x3fc8u : 0x3fe0u);
}
// the window bits a store replaces (masked_update, screen.py:792-816, 993-1007)
template <int MODE> __host__ __device__ constexpr uint32_t split_mask_own() { return MODE == kDHGR ? 0x7fu << 3 : 0xffu << 3; }

// row parts (prologue): the window bits of each half, compressed (ascending bit order)
template <int MODE> __host__ __device__ inline uint32_t split_row_left(uint32_t win, int odd)
{
    if (MODE == kDHGR) return win & 0xffu;
    return odd ? (win & 0x1ffu) : ((win & 0xffu) | (((win >> 10) & 1u) << 8));
}
template <int MODE> __host__ __device__ inline uint32_t split_row_right(uint32_t win, int odd)
{
    if (MODE == kDHGR) return win >> 4;
    return odd ? (((win >> 3) & 1u) | ((win >> 6) << 1)) : (win >> 5);
}
// content parts (greedy step): the bits of the stored byte that reach each half
template <int MODE> __host__ __device__ inline uint32_t split_content_left(uint32_t c, int odd)
{
    if (MODE == kDHGR) return c & 31u;
    const uint32_t rc = ((c & 0x7fu) << 1) | (c >> 7);  // odd HGR bytes sit rotated in the column (screen.py:566-569)
    return odd ? (rc & 63u) : ((c & 31u) | ((c >> 7) << 5));
}
template <int MODE> __host__ __device__ inline uint32_t split_content_right(uint32_t c, int odd)
{
    if (MODE == kDHGR) return (c >> 1) & 63u;
    const uint32_t rc = ((c & 0x7fu) << 1) | (c >> 7);
    return odd ? ((rc & 1u) | ((rc >> 3) << 1)) : (c >> 2);
}
template <int MODE> __host__ __device__ constexpr size_t split_left_entries()
{
    return (size_t)ModeTraits<MODE>::kOffsets << (SplitTraits<MODE>::kLeftCBits + SplitTraits<MODE>::kLeftRowBits);
}
template <int MODE> __host__ __device__ constexpr size_t split_right_entries()
{
    return (size_t)ModeTraits<MODE>::kOffsets << (SplitTraits<MODE>::kRightCBits + SplitTraits<MODE>::kRightRowBits);
}
// The same cut applied to the diff weights (Bitmap.diff_weights, screen.py:400-449: distance between
// the current and the target window of a byte): both windows are arbitrary, so a half is indexed
// by the row parts of both -- DWL[o][row_left(cur)][row_left(tgt)], DWR[o][row_right(cur)][row_right(tgt)].
template <int MODE> __host__ __device__ constexpr size_t split_dw_left_entries()
{
    return (size_t)ModeTraits<MODE>::kOffsets << (2 * SplitTraits<MODE>::kLeftRowBits);
}
template <int MODE> __host__ __device__ constexpr size_t split_dw_right_entries()
{
    return (size_t)ModeTraits<MODE>::kOffsets << (2 * SplitTraits<MODE>::kRightRowBits);
}
constexpr uint32_t kSplitInf = 0x3fffu;  // "no path": finite sums stay below it, and INF + INF fits 16 bits

// S from the two packed halves (lo 16 bits = component 0, hi = component 1)
__host__ __device__ inline uint32_t split_combine(uint32_t l, uint32_t r)
{
    const uint32_t a = (l & 0xffffu) + (r & 0xffffu), b = (l >> 16) + (r >> 16);
    return a < b ? a : b;
}

// ---- the narrow form the greedy kernels read -------------------------------------------------
// A greedy step is bound by what its 256 lookups cost, and that follows the bytes its slices occupy
// (tools/gather_ceiling.hip: the access pattern alone, no arithmetic, takes 1.42 ms per 12288-stream
// launch with 4-byte entries and 0.92 ms with 2-byte ones).  So the kernels read two u16 per byte and
// add them.  Of the two paths of S = min(l0 + r0, l1 + r1), path 0 -- a transposition of the two
// pixels across the cut -- needs the left state's OTHER component, l0 = E[M-2].  But
//     l1 = E[M-1] = min(l0 + s, E[M-3] + 1 if pixels (M-2, M-1) transpose),   s = sub(a[M-1], b[M-1]),
// and a transposition of (M-2, M-1) excludes one of (M-1, M) where the two paths differ: adjacent
// pixels are 4-dot windows that share three dots, so a string cannot swap the same pixel with both
// neighbours unless the three are equal.  Hence l0 = l1 - s wherever path 0 matters, s is known to the
// RIGHT half (pixel M-1 lies inside its bits), and
//     S = L1 + RF,    L1[o][content part][row] = l1,    RF[o][content part][row] = min(r1, r0 - s)
// for EVERY (offset, content, window): no exceptions, no third table.  Round 3 read R1 = r1 instead
// and sent the 1/64 of the pairs where path 0 wins to the dense table through per-content exception
// masks -- eleven address instructions per byte instead of two.  That the folded form is exact is not
// taken on trust: iiv_encoder_create expands it for every entry and compares with the caller's store
// table (an encoder whose tables disagree runs the dense-table workgroup kernel instead), and
// tests/test_gpu_tables.py::test_narrow_store_table_is_exact does the same for both modes, both
// palettes and random diff matrices.  r0 - s may be negative for matrices that are not metrics, so
// RF is stored with a bias that the step's key constant takes out again.
constexpr uint32_t kNarrowBias = 256;   // > max(dm) (<= 204: iiv_encoder_create)
struct NarrowTables {
    const uint8_t *base;      // the allocation: L1 | RF
    uint32_t right_off;       // byte offset of RF
    int exact;                // build_narrow_tables: the expansion equals the dense store table for every entry
};
template <int MODE> __host__ __device__ constexpr uint32_t narrow_right_off() { return (uint32_t)split_left_entries<MODE>() * 2; }
template <int MODE> __host__ __device__ constexpr size_t narrow_total_bytes()
{
    return narrow_right_off<MODE>() + split_right_entries<MODE>() * 2;
}
// the narrow form with content innermost, two byte values per word (iiv_tables.hip: joint_pack_kernel)
template <int MODE> struct JointPack {
    static constexpr int kPairs = (1 << ModeTraits<MODE>::kContentBits) / 128;   // words per lane and row: DHGR 1, HGR 2
};
template <int MODE> __host__ __device__ constexpr size_t joint_left_entries()
{
    return ((size_t)ModeTraits<MODE>::kOffsets << SplitTraits<MODE>::kLeftRowBits) * JointPack<MODE>::kPairs * 64;
}
template <int MODE> __host__ __device__ constexpr size_t joint_right_entries()
{
    return ((size_t)ModeTraits<MODE>::kOffsets << SplitTraits<MODE>::kRightRowBits) * JointPack<MODE>::kPairs * 64;
}
// the window back from its two rows
template <int MODE> __host__ __device__ inline uint32_t split_window_from_rows(uint32_t row_left, uint32_t row_right, int odd)
{
    if (MODE == kDHGR) return (row_right << 4) | (row_left & 15u);
    return odd ? ((row_right >> 1) << 6) | (row_left & 63u) : (row_right << 5) | (row_left & 31u);
}

// LDS accesses of one wave execute in order: making one lane's LDS writes visible to the
// other lanes of the same wave needs no s_barrier, only that the compiler keeps the order.
__device__ static inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Next MT19937 block, one wave, fully unrolled (constant LDS offsets, no loop counters).  Word i of the new block is
// new[i - 227] ^ mix(old[i], old[i + 1]) for i >= 227: with lane l computing words l + 64 g of the first third (i < 227),
// then 227 + l + 64 g, then 454 + l + 64 g, the new word each needs is the SAME lane's own earlier result -- a register.
// Every LDS read is of the OLD block, all of them issued up front; the block costs one LDS turn-around (its words are the
// next block's inputs) instead of three.  (The thirteen blocks of a prologue call took one wave 20 k cycles with a
// write -> barrier -> read per third: longer than the rest of the workgroup needs to score.)
__device__ static inline void mt_twist_wave(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int lane)
{
    auto mt_mix = [](uint32_t x, uint32_t y) -> uint32_t {   // iiv_device.h: mt_mix, its select as bit-field extract + and
        const uint32_t v = (x & 0x80000000u) | (y & 0x7fffffffu);
        return (v >> 1) ^ ((uint32_t)__builtin_amdgcn_sbfe(y, 0, 1) & 0x9908b0dfu);
    };
    uint32_t a[4], b[4], c[3];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int j = lane + 64 * g;
        a[g] = b[g] = 0;
        if (g < 3 || j < 227) {
            a[g] = src[j + 397] ^ mt_mix(src[j], src[j + 1]);
            b[g] = a[g] ^ mt_mix(src[227 + j], src[228 + j]);
        }
    }
    const uint32_t new0 = (uint32_t)__builtin_amdgcn_readlane((int)a[0], 0);   // "old[624]" of word 623 is the new word 0
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const int j = lane + 64 * g;
        c[g] = 0;
        if (g < 2 || j < 170) {
            const uint32_t nx = (g == 2 && j == 169) ? new0 : src[(g == 2 && j == 169) ? 0 : 455 + j];
            c[g] = b[g] ^ mt_mix(src[454 + j], nx);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int j = lane + 64 * g;
        if (g < 3 || j < 227) {
            dst[j] = a[g];
            dst[227 + j] = b[g];
        }
        if (g < 2 || (g == 2 && j < 170)) dst[454 + j] = c[g];
    }
    wave_lds_sync();
}

// Blocks 1 .. n_blocks of the stream behind block 0 (`blk0`, 624 words in LDS), written to gen[0 .. n_blocks), by ONE wave that
// keeps the latest block in REGISTERS: lane l holds words l + 64 g of each third -- A: words 0..226, B: 227..453,
// C: 454..623 -- so that every operand of the next block is either the lane's own register (old[i], and new[i - 227] as in
// mt_twist_wave), its neighbour lane's (old[i + 1]: a wave-wide DPP shift by one lane, the last lane patched from the next
// register) or a fixed rotation of the lanes (old[i + 397] = chain j + 170 of B for j < 57, chain j - 57 of C beyond: two
// rotations, by 42 and by 7 lanes: five ds_bpermute).  A block is then ONE trip through the LDS crossbar and no LDS memory
// read at all (its words are still written out for the nonce lookups, but nothing waits for those writes); the form that
// re-read every block from LDS took one wave ~1.1 k cycles per block beside fifteen waves whose gathers queue in front
// of its reads -- 14 k cycles for the thirteen blocks of a prologue call, more than the other waves need to score.
// (the block update itself: mt_regs_twist below -- one wave, the block in its registers)
struct MtRegs {
    uint32_t A[4], B[4], C[3];   // lane l: words l + 64 g (A), 227 + l + 64 g (B), 454 + l + 64 g (C)
};
// the lane constants of the update: the two rotations' bpermute addresses and where their results apply
struct MtLaneConsts {
    int addr7, addr42;
    bool lt22, lt57;
    __device__ explicit MtLaneConsts(int lane) : addr7(((lane + 7) & 63) << 2), addr42(((lane + 42) & 63) << 2), lt22(lane < 22), lt57(lane < 57) {}
};
__device__ static inline void mt_regs_load(MtRegs &m, const uint32_t *__restrict__ blk, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int j = lane + 64 * g;
        m.A[g] = m.B[g] = 0;
        if (g < 3 || j < 227) {
            m.A[g] = blk[j];
            m.B[g] = blk[227 + j];
        }
        if (g < 3) {
            m.C[g] = 0;
            if (g < 2 || j < 170) m.C[g] = blk[454 + j];
        }
    }
}
__device__ static inline void mt_regs_store(const MtRegs &m, uint32_t *__restrict__ dst, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int j = lane + 64 * g;
        if (g < 3 || j < 227) {
            dst[j] = m.A[g];
            dst[227 + j] = m.B[g];
        }
        if (g < 2 || (g == 2 && j < 170)) dst[454 + j] = m.C[g];
    }
}
// the next block, in place
__device__ __forceinline__ void mt_regs_twist(MtRegs &m, const MtLaneConsts &k)
{
    auto mix = [](uint32_t x, uint32_t y) -> uint32_t {   // iiv_device.h: mt_mix
        const uint32_t v = (x & 0x80000000u) | (y & 0x7fffffffu);
        return (v >> 1) ^ ((uint32_t)__builtin_amdgcn_sbfe(y, 0, 1) & 0x9908b0dfu);
    };
    // value of the next chain: lane l gets lane l + 1's `v`; lane `last` (63, or the third's last chain) gets `next0` (a scalar)
    auto up1 = [&](uint32_t v, uint32_t next0, auto last) -> uint32_t {
        uint32_t s = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);   // wave_shl:1
        asm("v_writelane_b32 %0, %1, %2" : "+v"(s) : "s"(next0), "n"(decltype(last)::value));
        return s;
    };
    auto lane0 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, 0); };
    auto rot = [](uint32_t v, int addr) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v); };
    uint32_t(&A)[4] = m.A, (&B)[4] = m.B, (&C)[3] = m.C;
    // old[j + 397]: chain j + 170 of B (j < 57), chain j - 57 of C (beyond)
    const uint32_t rb2 = rot(B[2], k.addr42), rb3 = rot(B[3], k.addr42);
    const uint32_t rc0 = rot(C[0], k.addr7), rc1 = rot(C[1], k.addr7), rc2 = rot(C[2], k.addr7);
    uint32_t X[4];
    X[0] = k.lt22 ? rb2 : k.lt57 ? rb3 : rc0;
    X[1] = k.lt57 ? rc0 : rc1;
    X[2] = k.lt57 ? rc1 : rc2;
    X[3] = rc2;
    uint32_t nA[4], nB[4], nC[3];
    // first third: new[j] = old[j + 397] ^ mix(old[j], old[j + 1]); old[227] is B's chain 0
#pragma unroll
    for (int g = 0; g < 4; g++)
        nA[g] = X[g] ^ mix(A[g], g < 3 ? up1(A[g], lane0(A[g < 3 ? g + 1 : 0]), std::integral_constant<int, 63>{}) : up1(A[g], lane0(B[0]), std::integral_constant<int, 34>{}));
    // second third: new[227 + j] = new[j] ^ mix(old[227 + j], old[228 + j]); old[454] is C's chain 0
#pragma unroll
    for (int g = 0; g < 4; g++)
        nB[g] = nA[g] ^ mix(B[g], g < 3 ? up1(B[g], lane0(B[g < 3 ? g + 1 : 0]), std::integral_constant<int, 63>{}) : up1(B[g], lane0(C[0]), std::integral_constant<int, 34>{}));
    // third third: new[454 + j] = new[227 + j] ^ mix(old[454 + j], old[455 + j]); "old[624]" is the new word 0
#pragma unroll
    for (int g = 0; g < 3; g++)
        nC[g] = nB[g] ^ mix(C[g], g < 2 ? up1(C[g], lane0(C[g < 2 ? g + 1 : 0]), std::integral_constant<int, 63>{}) : up1(C[g], lane0(nA[0]), std::integral_constant<int, 41>{}));
#pragma unroll
    for (int g = 0; g < 4; g++) {
        A[g] = nA[g];
        B[g] = nB[g];
        if (g < 3) C[g] = nC[g];
    }
}
// word idx[lane] (0 .. 623) of the block, per lane: a gather across the wave's registers -- word i sits in register
// code(i) = 4 t + (j >> 6), t = its third, j = i - 227 t, lane j & 63 -- by one ds_bpermute (the LDS crossbar, no LDS
// memory) and one select per register; registers outside [code(lo), code(hi)] (lo <= hi: scalar bounds of the indices
// that matter) are skipped.  A lane whose index does not matter may pass anything in 0 .. 623.
__device__ static inline uint32_t mt_regs_gather(const MtRegs &m, int idx, int lo, int hi)
{
    auto code_of = [](int i, int &j) -> int {
        const int t = (i >= 227 ? 1 : 0) + (i >= 454 ? 1 : 0);
        j = i - 227 * t;
        return 4 * t + (j >> 6);
    };
    int j, jl, jh;
    const int code = code_of(idx, j);
    const int c_lo = code_of(lo, jl), c_hi = code_of(hi, jh);
    const int addr = (j & 63) << 2;
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        if (k >= c_lo && k <= c_hi) {   // (scalar branch)
            const uint32_t src = k < 4 ? m.A[k] : k < 8 ? m.B[k - 4] : m.C[k - 8];
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)src);
            r = code == k ? v : r;
        }
    }
    return r;
}

__device__ static inline void mt_generate_wave(const uint32_t *__restrict__ blk0, uint32_t *__restrict__ gen, int n_blocks, int lane)
{
    MtRegs m;
    mt_regs_load(m, blk0, lane);
    const MtLaneConsts kc(lane);
    for (int k = 0; k < n_blocks; k++) {
        mt_regs_twist(m, kc);
        mt_regs_store(m, gen + k * 624, lane);
    }
    wave_lds_sync();
}

// launchers implemented in iiv_greedy.hip
struct GreedyArgs {
    StreamState *states;
    const uint8_t *frames_main, *frames_aux;
    int n_frames, n_streams;
    const LaunchSeg *segs;   // device
    int seg_stride;          // 0: one descriptor for every stream; 1: descriptor per stream
    const uint32_t *left, *right;  // split store table (u32 pairs: the joint choice, the builders)
    NarrowTables nt;               // its narrow form (the one-wave and the team kernel)
    uint8_t *ops_out;
    size_t ops_stride;       // bytes between the outputs of consecutive streams
    int lds_pad;             // extra dynamic LDS per stream (bytes): caps the streams resident per CU
    int uniform_bank;        // 0 / 1: every stream that emits opcodes in this round works on this bank; -1: they differ
    bool shared;             // IIV_GREEDY_WAVE_SHARED: the LDS-shared form wherever it applies (DHGR, one bank per round)
    int *queue;              // device: this launch's stream counter, zero (the LDS-shared form's persistent workgroups)
    bool fourth;             // IIV_OPT_FOURTH_OFFSET: up to three extra offsets per opcode (the plain one-wave kernel only)
    bool count_stats;        // the one-wave kernel adds to the streams' stat_* fields
    // Longest first (round 5): streams differ by up to 2x in how long a launch takes them (picture-like input), and a launch
    // ends when its slowest stream does -- in its last fifth a third (S-iid) to three quarters (S-img) of the wave slots idle.
    // The one-wave kernel writes every stream's shader clocks of this launch to cost[stream]; the encoder sorts the streams
    // by that, descending, every few launches (iiv_encode.hip: order_streams_kernel), and workgroup / queue position i runs
    // stream perm[i].  Streams are independent: the order changes no byte.  Either may be NULL.
    const int *perm;
    uint32_t *cost;
    // live hand-over (iiv_encode_live; the team kernel, one stream): a queue in coherent host memory that receives every
    // opcode as one tagged 8-byte store; NULL otherwise
    unsigned long long *live = nullptr;
    uint32_t live_tag = 0;
};

int launch_greedy_wave(int mode, const GreedyArgs &a, hipStream_t st, int *form_out = nullptr);   // iiv_greedy.hip; *form_out: 0 plain form launched, 1 LDS-shared
int launch_greedy_team(int mode, const GreedyArgs &a, hipStream_t st);   // iiv_team.hip


// ---- the prologue (iiv_prologue.hip) and the workgroup greedy kernel (iiv_workgroup.hip)
constexpr int kSelNeedMax = 2048;  // the prologue's prefix selection is used when 3 (fourth offset: 4) * opcode budget <= this
constexpr int kBucketMax = 384;    // counting-sort buckets larger than this fall back to the bitonic sort
struct PrologueArgs {
    StreamState *states;
    const uint8_t *frames_main, *frames_aux;
    int n_frames, n_streams;
    const LaunchSeg *segs;   // device
    int seg_stride;
    const uint16_t *table;          // full symmetric table (IIV_DW_TABLE)
    const ulonglong2 *strings;      // colour-string LUT
    const uint16_t *sub;            // 16 x 16 substitution costs
    const uint32_t *dwl, *dwr;      // split diff-weight table (IIV_DW_SPLIT)
    const uint32_t *hgr_dots;       // HGR: windows -> dots, [kHgrDotLutEntries] (iiv_tables.hip: hgr_dot_lut_kernel)
    const uint32_t *dw_pieces;      // the diff weights' pair terms, DHGR [2 banks][4096], HGR [4096] (iiv_tables.hip: dw_piece_kernel)
};
int launch_prologue(int mode, int dw_mode, const PrologueArgs &a, hipStream_t st);
struct WorkgroupArgs {
    StreamState *states;
    const uint8_t *frames_main, *frames_aux;
    int n_frames, n_streams;
    const LaunchSeg *segs;
    int seg_stride;
    const uint16_t *store;                 // dense store table
    const uint32_t *left_t, *right_t;      // joint content choice: the narrow form packed two byte values per word (joint == 2:
                                           // joint_pack_kernel) or the two-component split table, content innermost (joint == 1)
    uint8_t *ops_out;
    size_t ops_stride;
};
// joint: 0 = the reference's content byte, 1 / 2 = the joint content choice scored from the split table / from the packed narrow form
// fourth (with joint != 0 only): IIV_OPT_FOURTH_OFFSET together with the joint choice -- three extra offsets per opcode
int launch_greedy_workgroup(int mode, int joint, bool fourth, const WorkgroupArgs &a, hipStream_t st);

}  // namespace iiv
