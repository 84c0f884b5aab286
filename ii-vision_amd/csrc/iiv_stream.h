// iiv_stream.h -- per-stream encoder state in HBM and the formats shared by the
// prologue (iiv_prologue.hip), the greedy kernels (iiv_greedy.hip, iiv_team.hip, iiv_workgroup.hip) and the
// split-table builder (iiv_tables.hip).
// Reference: transcoder/video.py:16-301 (Video state), transcoder/screen.py:383-547.
#pragma once

#include "iiv_device.h"

namespace iiv {

constexpr int kPushedCap = 24576;  // >= 3 pushes (two, or three with the fourth-offset option) x 7680 non-hole bytes

// ---- wd[]: one word per byte of the live generator's bank, written once by the prologue
// and immutable while the generator lives:
//   bits  0..8   row index into the RIGHT half of the split store table  (split_row_right);
//                its low 5 / 6 bits are also the index into the "exception" masks below, which
//                is why it sits at bit 0: a shift instruction takes them as they are
//   bits 10..18  row index into the LEFT half                            (split_row_left);
//                with bit 9 (always 0) the 10-bit field at bit 9 is the byte offset of the
//                row's u16 inside a slice
//   bits 20..30  diff weight of the byte (Bitmap.diff_weights, screen.py:400-449), <= 2047
// Keys of the greedy step are `value << 20 | offset`, so `key - (wd & kWdDwMask)` is
// `delta << 20 | offset` with delta = store value - diff weight (screen.py:547) in [-2047, 2047].
constexpr int kWdDwShift = 20;
constexpr uint32_t kWdDwMask = 0xffffffffu << kWdDwShift;
constexpr int kWdLeftShift = 10, kWdRightShift = 0;
constexpr uint32_t kWdRowMask = 0x1ffu;
__host__ __device__ inline uint32_t wd_word(uint32_t row_left, uint32_t row_right, uint32_t dw)
{
    return (row_left << kWdLeftShift) | (row_right << kWdRightShift) | (dw << kWdDwShift);
}
constexpr int kMaxValue = 2047;  // every table value and diff weight must fit 11 bits

struct StreamState {
    uint8_t mem[2][8192];     // [is_aux] Video.memory_map / aux_memory_map
    int32_t up[2][8192];      // [is_aux] Video.update_priority / aux_update_priority
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
};

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
// A greedy step is bound by the L1's rate for divergent loads, and that rate follows the bytes
// its slices occupy (tools/gather_ceiling.hip: the access pattern alone, no arithmetic, takes
// 1.42 ms per 12288-stream launch with 4-byte entries and 0.92 ms with 2-byte ones).  Of the two
// paths of S = min(l0 + r0, l1 + r1), path 0 -- a transposition of the two pixels across the
// cut -- exists only when the source's pixels (M, M+1) are the target's swapped, 1.6 % of all
// (content, window) pairs.  So the kernels read
//     L1[o][content part][row] = l1,   R1[o][content part][row] = r1      (u16 each)
// and S = l1 + r1, except where an *exception mask* says the pair may differ from that: there
// the value comes from the dense store table S[o][content][window] itself (one more table, read
// by one lane in 64).  The mask is indexed by the few bits the swap condition can depend on:
//     X[o][content part][index],  index = low 5 (DHGR) / 6 (HGR) bits of the right row
//                                         (HGR odd bytes: bits 3..8 of the left row)
// and is *built by comparison*: a bit is set iff some (content, window) it covers has
// l1 + r1 != S (iiv_tables.hip: narrow_mask_kernel), so the scheme is exact by construction
// whatever the bits chosen; choosing them well only keeps the exceptions rare.
// To keep every load of a step unconditional (the software pipeline of iiv_greedy.hip depends on
// that), an excepted byte does not branch: its RIGHT load is redirected to the dense entry and
// its LEFT load to a zero word -- L1 | zero | R1 | dense copy live in ONE allocation, so a
// redirection is a different 32-bit offset from the same slice base.
struct NarrowTables {
    const uint8_t *base;                         // the allocation (its first bytes are L1)
    uint32_t zero_off, right_off, dense_off;     // byte offsets of the zero word, R1, the dense copy
    const void *xmask;                           // DHGR: u32 [4][64]; HGR: u64 [2][64]
};
template <int MODE> __host__ __device__ constexpr uint32_t narrow_zero_off() { return (uint32_t)split_left_entries<MODE>() * 2; }
template <int MODE> __host__ __device__ constexpr uint32_t narrow_right_off() { return narrow_zero_off<MODE>() + 256; }
template <int MODE> __host__ __device__ constexpr uint32_t narrow_dense_off()
{
    return narrow_right_off<MODE>() + (uint32_t)split_right_entries<MODE>() * 2;
}
template <int MODE> __host__ __device__ constexpr size_t narrow_total_bytes()
{
    return narrow_dense_off<MODE>() + (((size_t)ModeTraits<MODE>::kOffsets << (ModeTraits<MODE>::kContentBits + ModeTraits<MODE>::kBits)) * 2);
}
// which mask word / bit a (content, rows) pair falls under
template <int MODE> __host__ __device__ inline uint32_t narrow_mask_content(uint32_t c, int odd)
{
    if (MODE == kDHGR) return split_content_right<MODE>(c, odd);
    return odd ? split_content_left<MODE>(c, 1) : split_content_right<MODE>(c, 0);
}
template <int MODE> __host__ __device__ inline uint32_t narrow_mask_index(uint32_t row_left, uint32_t row_right, int odd)
{
    if (MODE == kDHGR) return row_right & 31u;
    return odd ? (row_left >> 3) & 63u : row_right & 63u;
}
// the window back from its two rows
template <int MODE> __host__ __device__ inline uint32_t split_window_from_rows(uint32_t row_left, uint32_t row_right, int odd)
{
    if (MODE == kDHGR) return (row_right << 4) | (row_left & 15u);
    return odd ? ((row_right >> 1) << 6) | (row_left & 63u) : (row_right << 5) | (row_left & 31u);
}
// byte offsets, relative to the (offset, content) slice bases of L1 and R1, of the two u16 a
// byte's value is the sum of.  xm = the byte's mask word (u32 DHGR / u64 HGR, wave-uniform per
// parity), zrel = zero word - L1 slice base, drel = dense[o][content][0] - R1 slice base.
template <int MODE, int ODD, typename M>
__device__ static inline void narrow_offsets(uint32_t wd, M xm, uint32_t zrel, uint32_t drel, uint32_t &off_l, uint32_t &off_r)
{
    const uint32_t rr = wd & kWdRowMask;
    const uint32_t lr2 = __builtin_amdgcn_ubfe(wd, kWdLeftShift - 1, 10);   // left row << 1: bit 9 of wd is 0
    // the byte's exception bit (v_bfe_u32 takes the low 5 bits of its offset operand as they are)
    uint32_t t;
    if (MODE == kDHGR) t = __builtin_amdgcn_ubfe((uint32_t)xm, wd, 1);
    else if (!ODD) t = (uint32_t)(xm >> (wd & 63u)) & 1u;
    else t = (uint32_t)(xm >> ((wd >> (kWdLeftShift + 3)) & 63u)) & 1u;
    // (window << 1) + drel, the window put back together from its rows (split_window_from_rows)
    uint32_t dl;
    if (MODE == kDHGR) dl = (rr << 5) + ((lr2 & 30u) + drel);
    else if (!ODD) dl = (rr << 6) + ((lr2 & 62u) + drel);
    else dl = ((rr >> 1) << 7) + ((lr2 & 126u) + drel);
    // (bit selects on the exception bit spread over the word: one v_bfe_i32 and two v_bfi_b32 instead of a compare and
    // two conditional moves)
    const uint32_t m = 0u - t;
    off_l = (zrel & m) | (lr2 & ~m);
    off_r = (dl & m) | ((rr << 1) & ~m);
}

// LDS accesses of one wave execute in order: making one lane's LDS writes visible to the
// other lanes of the same wave needs no s_barrier, only that the compiler keeps the order.
__device__ static inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// next MT19937 block, one wave, fully unrolled (constant LDS offsets, no loop counters)
__device__ static inline void mt_twist_wave(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int lane)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = lane + 64 * k;
        if (k < 3 || i < 227) dst[i] = src[i + 397] ^ mt_mix(src[i], src[i + 1]);
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = 227 + lane + 64 * k;
        if (k < 3 || i < 454) dst[i] = dst[i - 227] ^ mt_mix(src[i], src[i + 1]);
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = 454 + lane + 64 * k;
        if (k < 2 || i < 624) {
            const uint32_t nx = (i == 623) ? dst[0] : src[i + 1];
            dst[i] = dst[i - 227] ^ mt_mix(src[i], nx);
        }
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
};

int launch_greedy_wave(int mode, const GreedyArgs &a, hipStream_t st);   // iiv_greedy.hip
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
    const uint2 *hgr_slut;          // HGR: three-lookup string table
};
int launch_prologue(int mode, int dw_mode, const PrologueArgs &a, hipStream_t st);
struct WorkgroupArgs {
    StreamState *states;
    const uint8_t *frames_main, *frames_aux;
    int n_frames, n_streams;
    const LaunchSeg *segs;
    int seg_stride;
    const uint16_t *store;                 // dense store table
    const uint32_t *left_t, *right_t;      // split store table, content-innermost (joint content choice)
    uint8_t *ops_out;
    size_t ops_stride;
};
int launch_greedy_workgroup(int mode, bool joint, const WorkgroupArgs &a, hipStream_t st);

}  // namespace iiv
