// iiv_greedy.hip -- the greedy selection loop of video.Video._index_changes /
// _compute_error (transcoder/video.py:121-187, 275-301) with Bitmap.apply
// (transcoder/screen.py:256-293), one 64-lane wave per video stream.
//
// Lane l owns page bytes 4l..4l+3.  Everything that is uniform per opcode (pop, validity,
// candidate counts, the two winners, RNG cursor, opcode emission) lives in SGPRs; the
// control flow is scalar branches; there is no workgroup barrier and no LDS exchange in
// the loop (LDS accesses of one wave execute in order).
//
// Store values come from the SPLIT store table (iiv_stream.h): value = min(l0 + r0,
// l1 + r1) with (l0, l1) = left[offset][content bits][row bits] and (r0, r1) from the right
// half.  A (offset, content) slice of either half is 1-2 KiB, so the 64 lanes of one load
// fall into 8-16 cache lines (they fell into ~45 of the 86 lines of a 10-bit slice of the
// dense table), and the whole table (0.5 MiB) stays in L2 in both modes.
//
// The loop over the sorted initial list is software-pipelined three deep.  What a step
// loads depends only on immutable data of the live generator (the entry's row of wd[] and
// the table slices of its content byte), never on the outcome of earlier steps -- only the
// validity bits and the priorities do.  So while entry k is scored, the eight table loads of
// entry k+1 and the row load of entry k+2 are in flight:
//     take(k+2) -> row load | row(k+1) arrived -> 8 gathers | gathers(k) arrived -> score, apply
// An entry whose priority was cleared after it was taken is skipped by the step itself
// (video.py:130), exactly as the reference's lazy deletion does.
//
// Scoring, fast form.  The reference orders candidates by (delta, nonce, offset)
// (video.py:290-301) and draws one nonce per candidate, eligible or not.  The two winners
// depend on the nonces only if two eligible candidates share the smallest or the
// second-smallest delta; otherwise the random stream just advances by the number of
// candidates.  A step reduces signed keys delta << 18 | offset with two fused-DPP wave
// minima and checks that no third eligible key shares the second delta; only on a tie
// (2.7 % of the opcodes of the bench workload) it re-scores the entry with every nonce
// materialised in reference order.  Exact either way.
#include "iiv_host.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include "iiv_stream.h"
#include "iiv_wave.h"

namespace iiv {

#ifndef IIV_WAVE_OCC
#define IIV_WAVE_OCC 6     // waves per SIMD the register allocation is held to
#endif
#ifndef IIV_STEP_PRIO
#define IIV_STEP_PRIO 1    // s_setprio of a wave while it scores and applies an entry (0 elsewhere)
#endif

// W = streams (waves) per workgroup.  W == 1: every table load goes to the L1/TA (both modes).
// W > 1 (DHGR): the workgroup's streams all work on the same bank and share that bank's whole L1 half
// (2 offsets x 32 content parts x 256 rows x 2 B = 32 KiB), copied into LDS once per launch: four of a
// step's eight table loads become ds_read_u16.  What bounds a step is the TA's rate for divergent loads
// (DESIGN.md 3.7); the LDS serves a random 2-byte gather in ~7 cycles where the TA needs 16-30
// (tools/gather_ceiling.hip variant E: 0.74 ms per 12288-stream launch against 0.98).
#ifndef IIV_SHARED_W
#define IIV_SHARED_W 8
#endif
#ifndef IIV_TAKE_CHECK_PLAIN
#define IIV_TAKE_CHECK_PLAIN 1
#endif
#ifndef IIV_TAKE_CHECK_SHARED
#define IIV_TAKE_CHECK_SHARED 0   // 1: the LDS-shared form also tests an entry's live bit when it takes it off the window (see take())
#endif
#ifndef IIV_SHARED_W_HGR
#define IIV_SHARED_W_HGR 16
#endif
struct WaveLds {                // per stream: 5568 B
    uint32_t nz[256];           // update_priority != 0
    uint32_t pdone[256];        // byte already emitted as a primary (its diff weight counts as 0)
    uint32_t mt[624 + 256];     // random's current MT19937 block + the first 256 words of the next one
};
struct WaveLdsBits {            // per stream, MT19937 in registers (HGR's LDS-shared form): 2048 B
    uint32_t nz[256];
    uint32_t pdone[256];
};
// What the LDS-shared form keeps in LDS per workgroup, and how many streams share it:
//   DHGR: the L1 halves of BOTH byte offsets of the bank (2 x 16 KiB), eight streams, two workgroups per CU;
//   HGR:  round 6: the L1 halves of BOTH byte offsets (2 x 64 KiB), sixteen streams, one workgroup per CU: four of a
//         step's eight table loads come from LDS.  HGR's step IS bound by its loads, and the ceiling of its access
//         pattern drops from 3.61 (all of them through the L1 / TA) over 3.35 (one half in LDS: rounds 3-5) to 2.14 ms
//         per 14336-stream launch this way (tools/gather_ceiling D 14336 HGR with IIV_GATHER_E=1, variant E2).  128 KiB of
//         the CU's 160 leave 2 KiB per stream -- the two bitmaps -- so random's MT19937 block lives in eleven
//         REGISTERS per lane there (kMtRegs below).
//         (IIV_SHARED_HGR_HALVES=1: round 5's form, the even offset's half only and MT19937 in LDS.)
#ifndef IIV_SHARED_HGR_HALVES
#define IIV_SHARED_HGR_HALVES 2
#endif
template <int MODE> struct SharedCfg {
    static constexpr int kOffsets = MODE == kDHGR ? 2 : IIV_SHARED_HGR_HALVES;              // byte offsets whose L1 lives in LDS
    static constexpr int kHalfBytes = 2 << (SplitTraits<MODE>::kLeftCBits + SplitTraits<MODE>::kLeftRowBits);   // one offset's L1
    static constexpr int kL1Bytes = kOffsets * kHalfBytes;
    static constexpr int kPad = kL1Bytes;
    static constexpr int kW = MODE == kDHGR ? IIV_SHARED_W : IIV_SHARED_W_HGR;
    static constexpr bool kMtRegs = MODE == kHGR && kOffsets == 2;                          // MT19937 in registers, not in LDS
    static constexpr int kWaveBytes = kMtRegs ? (int)sizeof(WaveLdsBits) : (int)sizeof(WaveLds);
    static constexpr int kLds = kPad + kW * kWaveBytes;
    static_assert(kLds <= 160 * 1024, "a workgroup's LDS");
};

__device__ static inline WaveLds *own_wave_lds()
{
    __shared__ WaveLds w;
    return &w;
}

// FOUR (f4, IIV_OPT_FOURTH_OFFSET; NOT the reference's behaviour): up to three extra offsets per opcode instead of two
// and a copy of the first -- the reference's exit test `len(offsets) == 3` (video.py:180-181) read as the 4 that
// video.py:146 announces; defined by oracle/iiv_oracle.c: orc_video_set_fourth_offset, pinned against the reference
// run with that literal changed (tests/golden/g8_fourth_offset.npz).
template <int MODE, int W, bool FOUR = false>
__global__ __launch_bounds__(64 * W, W == 1 ? IIV_WAVE_OCC : MODE == kHGR ? (W + 3) / 4 : (2 * W + 3) / 4 + (W % 4 ? 1 : 0)) void greedy_wave_kernel(StreamState *__restrict__ states,
                                                                   const uint8_t *__restrict__ frames_main,
                                                                   const uint8_t *__restrict__ frames_aux, int n_frames,
                                                                   const LaunchSeg *__restrict__ segs, int seg_stride,
                                                                   const NarrowTables nt,
                                                                   uint8_t *__restrict__ ops_out, size_t ops_stride, int n_streams, int bank, int *__restrict__ queue,
                                                                   int count_stats, const int *__restrict__ perm, uint32_t *__restrict__ cost)
{
    using T = SplitTraits<MODE>;
    constexpr uint32_t INF = 0xffffffffu;
    typedef uint32_t __attribute__((aligned(2))) u32_a2;
    using SC = SharedCfg<MODE>;
    extern __shared__ uint32_t dyn_lds[];
    const int lane0 = W == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    const int wave = W == 1 ? 0 : IIV_SGPR(threadIdx.x >> 6);
    // W == 1: static (constant LDS offsets); W > 1: carved from dyn_lds behind the shared table
    // kMtRegs: random's MT19937 block in eleven registers per lane instead of LDS (HGR's LDS-shared form, SharedCfg)
    constexpr bool kMtRegs = W > 1 && SC::kMtRegs;
    uint32_t *nz, *pdone, *mt;
    if constexpr (W == 1) {
        __shared__ uint32_t nz_s[256], pdone_s[256], mt_s[624 + 256];
        nz = nz_s, pdone = pdone_s, mt = mt_s;
    } else if constexpr (kMtRegs) {
        WaveLdsBits *wl = reinterpret_cast<WaveLdsBits *>(reinterpret_cast<char *>(dyn_lds) + SC::kPad) + wave;
        nz = wl->nz, pdone = wl->pdone, mt = nullptr;
    } else {
        WaveLds *wl = reinterpret_cast<WaveLds *>(reinterpret_cast<char *>(dyn_lds) + SC::kPad) + wave;
        nz = wl->nz, pdone = wl->pdone, mt = wl->mt;
    }
    const char *const l1_lds = reinterpret_cast<const char *>(dyn_lds);
    if (W > 1) {
        // (`bank`: the host launches this kernel only when every stream that emits opcodes in this round works
        // on the same bank, and says which)
        uint4 *dst = reinterpret_cast<uint4 *>(dyn_lds);
        constexpr int kQuads = SC::kHalfBytes / 16;
#pragma unroll
        for (int h = 0; h < SC::kOffsets; h++) {
            const int o = byte_offset<MODE>(h, bank);
            const uint4 *src = reinterpret_cast<const uint4 *>(nt.base + ((size_t)o << (T::kLeftCBits + T::kLeftRowBits + 1)));
            for (int i = threadIdx.x; i < kQuads; i += 64 * W) dst[h * kQuads + i] = src[i];
        }
        __syncthreads();
    }

    // One stream from start to end (the whole body of the W == 1 kernel).  W > 1: the workgroups are PERSISTENT --
    // as many as fit the GPU at once -- and every wave takes the next stream off a queue (one atomic counter per
    // launch) as soon as it is done with one: a workgroup's LDS slot would otherwise idle until the slowest of its
    // W streams ends, and a launch's last round would run partly empty (streams differ by +-20 % in how long a
    // launch takes them; measured with s_memrealtime stamps: 13 of 16-20 wave slots per CU busy on average).
    auto run_stream = [&](const int stream, const int lane) {
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
    StreamState &S = states[stream];
    const LaunchSeg g = segs[(size_t)stream * seg_stride];
    const int n_ops = IIV_SGPR(g.n_ops), is_aux = IIV_SGPR(g.is_aux), frame = IIV_SGPR(g.frame);
    if (n_ops <= 0) return;
    if (W > 1 && is_aux != bank) {
        if (lane == 0 && !S.error) S.error = kErrBankMix;
        return;
    }
    uint8_t *out = ops_out + (size_t)stream * ops_stride + (size_t)IIV_SGPR(g.ops_base) * 6;

    if (!S.gen_active || S.error) {
        if (lane == 0 && !S.error) S.error = kErrNoGenerator;
        return;
    }
    for (int i = lane; i < 256; i += 64) {
        nz[i] = S.nzbits[i];
        pdone[i] = S.pdone[i];
    }
    if constexpr (!kMtRegs)
        for (int i = lane; i < 624; i += 64) mt[i] = S.mt_py[i];
    if (W == 1) __syncthreads(); else wave_lds_sync();
    // A step reads nonces at mt_idx + t, t <= 256 (one per candidate, then <= 2 for the
    // re-queued bytes), so it can run at most 256 words into the next block: only that much
    // of it is kept ahead (`ahead`, computable from the current block alone plus itself).
    // When the current block is used up, the head moves down, the other 368 words are
    // generated in place (word i needs the old words i, i + 1 and the new word i - 227) and
    // a new head is generated.
    uint32_t *ahead = mt + 624;
    auto gen_ahead = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = lane + 64 * k;
            if (k < 3 || i < 227) ahead[i] = mt[i + 397] ^ mt_mix(mt[i], mt[i + 1]);
        }
        wave_lds_sync();
        {
            const int i = 192 + lane;
            if (i >= 227) ahead[i] = ahead[i - 227] ^ mt_mix(mt[i], mt[i + 1]);
        }
        wave_lds_sync();
    };
    auto gen_rest = [&]() {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = 256 + 64 * k + lane;
            if (k < 5 || i < 624) {
                const uint32_t nv = i < 483 ? ahead[i - 227] : mt[i - 227];
                const uint32_t nx = i == 623 ? ahead[0] : mt[i + 1];
                const uint32_t v = nv ^ mt_mix(mt[i], nx);
                wave_lds_sync();  // every lane has read its old words before any lane overwrites one
                mt[i] = v;
            }
            wave_lds_sync();
        }
    };
    auto move_head = [&]() {
#pragma unroll
        for (int k = 0; k < 4; k++) mt[lane + 64 * k] = ahead[lane + 64 * k];
        wave_lds_sync();
    };
    int mt_idx = IIV_SGPR(S.mt_py_idx);
    // ---- kMtRegs: the block in registers (iiv_stream.h: MtRegs), nothing of it in LDS.
    // The fast path of a step does not LOOK at a nonce: it only counts draws, and the nonce of a re-queued entry is needed
    // once its key is read back (phase B, a later launch).  So a re-queued entry waits in a register queue -- lane q of
    // (q_key, q_idx): its key without the nonce, and which draw its nonce is, counted from the start of the block the
    // REGISTERS hold -- and the registers do not even follow the draws: `mt_behind` counts the blocks mt_idx has moved on
    // since (a step that uses up a block adds 624 to nothing but that counter).  mt_service catches up: it finishes the
    // queued entries of the registers' block (one gather across the registers, coalesced stores of the keys), twists,
    // and so on, block by block -- when the queue is nearly full (every ~25 steps on S-iid: eight or nine blocks in one
    // go), before the exact path reads nonces, before phase B and at the end.  Draws are in order, so the queue is sorted
    // by q_idx and what is resolved is always a prefix.  The exact path is the one place that reads nonces at once -- up
    // to 256, straight from the registers; if they reach into the next block it takes the current block's share first
    // and has mt_service move the registers one block AHEAD (mt_behind = -1) for the rest; that same step's draws then
    // carry mt_idx past 624 and the count back to 0.
    MtRegs mtr;
    const MtLaneConsts mtk(lane);
    uint32_t q_key = 0;
    int q_idx = 0, n_pend = 0, mt_behind = 0;
    constexpr int kQueueTrash = 63, kQueueHigh = 56;   // lane 63 takes the writes of bytes that are not re-queued
    if constexpr (kMtRegs) {
        mt_regs_load(mtr, S.mt_py, lane);
        if (mt_idx >= 624) {
            mt_regs_twist(mtr, mtk);
            mt_idx -= 624;
        }
    } else {
        gen_ahead();
        if (mt_idx >= 624) {
            move_head();
            gen_rest();
            gen_ahead();
            mt_idx -= 624;
        }
    }
    // after a block switch only words 0..255 of the current block are in place until twist_now()
    bool twist_pending = false;
    auto twist_now = [&]() {
        if constexpr (!kMtRegs) {
            if (__builtin_expect(twist_pending, 0)) {
                gen_rest();
                gen_ahead();
                twist_pending = false;
            }
        }
    };

    const int n_sorted = IIV_SGPR(S.n_sorted);
    const int truncated = IIV_SGPR(S.truncated);
    int head = IIV_SGPR(S.head), n_pushed = IIV_SGPR(S.n_pushed), exhausted = IIV_SGPR(S.exhausted);
    int done = 0, err = 0;
    uint32_t draws = 0;
    unsigned long long pad_ops = 0;
    const uint32_t pad_content = (uint32_t)IIV_SGPR(S.pad_content);

    // table slices of the even / odd page bytes of this bank (narrow form, iiv_stream.h): byte
    // offsets, inside the one allocation, of L1[o] and RF[o]
    const int o_e = byte_offset<MODE>(0, is_aux), o_d = byte_offset<MODE>(1, is_aux);
    const uint32_t l1_e = (uint32_t)o_e << (T::kLeftCBits + T::kLeftRowBits + 1), l1_d = (uint32_t)o_d << (T::kLeftCBits + T::kLeftRowBits + 1);
    const uint32_t r1_e = nt.right_off + ((uint32_t)o_e << (T::kRightCBits + T::kRightRowBits + 1));
    const uint32_t r1_d = nt.right_off + ((uint32_t)o_d << (T::kRightCBits + T::kRightRowBits + 1));
    // this stream's state as a raw buffer, for the stores of a step
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void *)&S, 0, (int)sizeof(StreamState), 0x00020000);
    const int up_off = (int)offsetof(StreamState, up16) + is_aux * 8192 * 2, mem_off = (int)offsetof(StreamState, mem) + is_aux * 8192;
    // the allocation as a raw buffer (no bounds: every offset formed below lies inside it by construction)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)nt.base, 0, 0x7fffffff, 0x00020000);
    const uint8_t *tgt_frames = (MODE == kDHGR && is_aux ? frames_aux : frames_main) +
                                ((size_t)stream * n_frames + frame) * 8192;
    const uint4 *wd_rows = reinterpret_cast<const uint4 *>(S.wd);
    const int wsel = lane >> 3;          // this lane's word inside a page's 8 bitmap words
    const int sh0 = (4 * lane) & 31;     // its 4 bits inside that word
    const uint32_t y0 = 4u * (uint32_t)lane;
    // A key is value * kKeyMul + (wd << 20) + y = delta << 20 | value << 8 | y (iiv_stream.h: wd carries minus the diff
    // weight in its low bits); RF's bias leaves through the per-byte constant that also brings the offset y in
    constexpr uint32_t kKeyMul = (1u << kWdDwShift) | (1u << 8);
    const uint32_t ycst0 = y0 - kNarrowBias * kKeyMul;

    // What a step needs of one entry: its eight table words and the diff-weight fields of its
    // row.  Two such sets alternate (one is scored while the other is being loaded); the loop
    // below is written out twice so that no in-flight register is ever copied -- a copy
    // would make the compiler wait for the load right there.
    struct Loaded {
        uint32_t gl[4], gr[4], kb[4];   // the two table words per byte; -diff weight << 20 | y (less the bias term)
    };
    // the eight table loads of one entry (content c, row w): a lane's bytes 0, 2 are even page
    // offsets, 1, 3 odd ones; both lookups of a slice back to back (the second finds the
    // slice's lines in L1)
    auto gather8 = [&](const uint4 &w, uint32_t c, Loaded &L) {
        // slice bases of this content byte
        // (W > 1: byte offsets into the LDS copy -- the bank's even-offset half, then its odd-offset half)
        const uint32_t sl_e = (W == 1 ? l1_e : 0u) + (split_content_left<MODE>(c, 0) << (T::kLeftRowBits + 1));
        constexpr bool kOddInLds = W > 1 && SC::kOffsets == 2;   // (HGR: the odd bytes' L1 stays with the L1 / TA)
        const uint32_t sl_d = (kOddInLds ? (uint32_t)SC::kHalfBytes : l1_d) + (split_content_left<MODE>(c, 1) << (T::kLeftRowBits + 1));
        const uint32_t sr_e = r1_e + (split_content_right<MODE>(c, 0) << (T::kRightRowBits + 1));
        const uint32_t sr_d = r1_d + (split_content_right<MODE>(c, 1) << (T::kRightRowBits + 1));
        const uint32_t wr[4] = {w.x, w.y, w.z, w.w};
        uint32_t ol[4], orr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            ol[r] = wd_off_left(wr[r]);     // one shift
            orr[r] = wd_off_right(wr[r]);   // one bit-field extract
        }
        // buffer loads: address = allocation + scalar slice offset + lane offset, so a slice base is
        // one 32-bit SGPR instead of a 64-bit pointer formed per opcode (-8 scalar instructions, +1 %)
        if (W == 1) {
            L.gl[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[0], (int)sl_e, 0);
            L.gl[2] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[2], (int)sl_e, 0);
            L.gl[1] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[1], (int)sl_d, 0);
            L.gl[3] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[3], (int)sl_d, 0);
        } else {
            L.gl[0] = *reinterpret_cast<const uint16_t *>(l1_lds + (sl_e + ol[0]));
            L.gl[2] = *reinterpret_cast<const uint16_t *>(l1_lds + (sl_e + ol[2]));
            if (kOddInLds) {
                L.gl[1] = *reinterpret_cast<const uint16_t *>(l1_lds + (sl_d + ol[1]));
                L.gl[3] = *reinterpret_cast<const uint16_t *>(l1_lds + (sl_d + ol[3]));
            } else {
                L.gl[1] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[1], (int)sl_d, 0);
                L.gl[3] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)ol[3], (int)sl_d, 0);
            }
        }
        L.gr[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)orr[0], (int)sr_e, 0);
        L.gr[2] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)orr[2], (int)sr_e, 0);
        L.gr[1] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)orr[1], (int)sr_d, 0);
        L.gr[3] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)orr[3], (int)sr_d, 0);
        // (the empty asm keeps the compiler from sinking these four shift-adds to the scoring two
        // half-iterations later, which would keep the whole row alive until then)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            L.kb[r] = (wr[r] << kWdDwShift) + (ycst0 + (uint32_t)r);
            asm volatile("" : "+v"(L.kb[r]));
        }
    };

    // opcodes emitted but not yet written: opcode ob_base + l sits in lane l of (ob0, ob1, ob2[, ob3]) -- as the step has
    // them: its entry word (page << 8 | x | content << 16) and its winners (store value << 8 | offset); the six bytes of
    // (page + 32, content, x, y1, y2, x) are formed by the 64 lanes at once when the buffer leaves, not by scalar
    // instructions in every step (round 5: not even the packing of the offsets)
    uint32_t ob0 = 0, ob1 = 0, ob2 = 0, ob3 = 0;
    int ob_base = 0, ob_n = 0;   // ob_n = done - ob_base, kept as a counter of its own
    auto flush_ops = [&]() {
        if (lane < ob_n) {
            uint8_t *q = out + (size_t)(ob_base + lane) * 6;
            const uint32_t pg = (ob0 >> 8) & 31u, xx = ob0 & 255u, cc = (ob0 >> 16) & 255u;
            *reinterpret_cast<u32_a2 *>(q) = (pg + 32u) | (cc << 8) | (xx << 16) | (ob1 << 24);
            *reinterpret_cast<uint16_t *>(q + 4) = (uint16_t)((ob2 & 255u) | ((FOUR ? ob3 & 255u : xx) << 8));
        }
        ob_base = done;
        ob_n = 0;
    };

    // video.py:140-144, 170-187; screen.py:256-293.  Lanes 0..2 (FOUR: 0..3) carry (x, y1, y2[, y3]); a missing secondary
    // repeats the primary's stores.  W1..W3: a winner as `store value << 8 | offset` (0 | x for a missing one).
    // Round 5: a third of a step's instructions were here.  Nothing below branches on what a LANE holds: a byte whose store
    // leaves no error clears its priority bit with an LDS AND whose mask is all ones for the others, a byte that is
    // re-queued stores its entry at an offset that is out of the stream's buffer range for the others (the hardware drops
    // it) -- no exec juggling, no second code path; the winners travel as one packed word per lane; the opcode's bytes are
    // formed at flush time (flush_ops); the pushed-entry capacity cannot be exceeded (iiv_stream.h: kPushedCap) and is
    // checked against a margin with one compare.
    // (phase B keeps per-lane minima of the re-queued bag up to date: `track` hands it the keys a step pushes)
#ifdef IIV_STAMPS
    int n_ties = 0, n_tie_members = 0, n_ties_small = 0;
    // (what the nonces have to order at a tie: the bytes sharing the smallest delta if there are two or more of them, else
    // those sharing the second -- how many, and how many a LANE holds at most)
    int n_rel = 0, n_rel_single = 0, n_rel_double = 0;
#endif
    bool prev_tie = false;        // the previous step's two winners shared their delta: expect the same of this one
    int n_exact = 0;              // steps of this launch that took the exact-nonce path (tie_stats: what the host picks the kernel form by)
    uint32_t pkey_v = 0;          // lanes 1, 2 (FOUR: and 3): the keys pushed by the latest step (track only)
    int push_f1 = 0, push_f2 = 0, push_f3 = 0, push_base = 0;
    static_assert(kPushedCap >= 3 * 7680 + 8, "a generator's steps have distinct primaries: <= 7680 steps x <= 3 pushes");
    // kMtRegs: queue entries [a, b) finished -- their nonces are words q_idx - off of the block in the registers -- and
    // stored (entry q belongs to slot n_pushed - n_pend + q of pushed[]); returns the keys, lane q = entry q
    auto mt_resolve = [&](int a, int b, int off) -> uint32_t {
        const int lo = __builtin_amdgcn_readlane(q_idx, a) - off, hi = __builtin_amdgcn_readlane(q_idx, b - 1) - off;
        const bool in = lane >= a && lane < b;
        const uint32_t w = mt_regs_gather(mtr, in ? q_idx - off : lo, lo, hi);
        const uint32_t key = q_key | ((mt_temper(w) >> 24) << 13);
        const uint32_t slot = in ? (uint32_t)(n_pushed - n_pend + lane) * 4u : 0x7ffffff0u;
        __builtin_amdgcn_raw_buffer_store_b32(key, rsrc_s, (int)slot, (int)offsetof(StreamState, pushed), 0);
        return key;
    };
    // lane k's entry (keys: the winners' lanes) -> queue lane n_pend + pos (pos = the re-queued entries in front of it), or
    // the trash lane if it is not re-queued; its nonce is draw first + pos of the current block
    auto mt_enqueue = [&](uint32_t keys, int k, int fk, int pos, int first) {
        const uint32_t key0 = (uint32_t)__builtin_amdgcn_readlane((int)keys, k);
        const int at = fk ? n_pend + pos : kQueueTrash;
        asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
            : "+v"(q_key), "+v"(q_idx)
            : "s"(key0), "s"(first + pos), "s"(at));
    };
    // catch up until the registers hold block `target` relative to the one mt_idx counts in (0: that block; -1: the one
    // after it), finishing every queued entry of the blocks passed and of the block arrived at.  Returns the keys finished
    // LAST (lane q = entry q of the queue as it then stood: what phase B's bookkeeping reads right after a step).
    auto mt_service = [&](int target) -> uint32_t {
        uint32_t keys = 0;
        for (;;) {
            const int r = (int)__popcll(__ballot(lane < n_pend && q_idx < 624));
            if (r > 0) {
                keys = mt_resolve(0, r, 0);
                if (r < n_pend) {   // the entries left over move down to lane 0
                    const int from = ((lane + r) & 63) << 2;
                    q_key = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)q_key);
                    q_idx = __builtin_amdgcn_ds_bpermute(from, q_idx);
                }
                n_pend -= r;
            }
            if (mt_behind <= target) break;
            mt_regs_twist(mtr, mtk);
            mt_behind--;
            q_idx -= 624;
        }
        return keys;
    };
    auto apply = [&](auto track, uint32_t e, uint32_t W1, uint32_t W2, uint32_t W3, int C) {
        const int p = (e >> 8) & 31, x = e & 255;
        const uint32_t c = (e >> 16) & 0xffu;
        // (a winner's store value is non-zero: it is re-queued.  As s_min_u32: written as min(v, 1) or v != 0 the flag takes
        // a v_cndmask / v_readfirstlane round trip through a vector register)
        auto flag = [](uint32_t w) -> int {
            int f;
            asm("s_min_u32 %0, %1, 1" : "=s"(f) : "s"(w >> 8) : "scc");   // (s_min writes SCC)
            return f;
        };
        const int f1 = flag(W1), f2 = flag(W2), f3 = FOUR ? flag(W3) : 0;
        if (__builtin_expect(n_pushed > kPushedCap - 4, 0)) {
            err = kErrPushedOverflow;
            return;
        }
        if (__builtin_expect(twist_pending && mt_idx + C + (FOUR ? 3 : 2) >= 256, 0)) twist_now();
        // lane 0 = x | 0 << 8, lane 1 = W1, lane 2 = W2 (lane 3 = W3); where a re-queued entry goes: lane 1 slot 0, lane 2
        // slot f1 (lane 3 slot f1 + f2)
        // (scalars written into lanes of one register each: a `lane == k ? a : b` chain compiles to selects on
        // loop-invariant lane masks, which the allocator then spills and reloads on every step)
        uint32_t w_v = (uint32_t)x, k_v = 0u;
        asm("v_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %1, %4, 2"
            : "+v"(w_v), "+v"(k_v)
            : "s"(W1), "s"(W2), "s"(f1));   // (wave-uniform by construction: no v_readfirstlane round trip for a flag)
        if (FOUR)
            asm("v_writelane_b32 %0, %2, 3\n\tv_writelane_b32 %1, %3, 3" : "+v"(w_v), "+v"(k_v) : "s"(W3), "s"(f1 + f2));
        // the opcode: the entry word and the winners go into lane ob_n of three registers; 64 of them leave in two
        // coalesced stores (flush_ops).  (Here, not at the end: the winners' scalar registers are free from here on)
        // (gfx9 VOP3 reads one SGPR only; v_writelane may take its lane select from m0 besides)
        asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0"
            : "+v"(ob0), "+v"(ob1), "+v"(ob2)
            : "s"(IIV_SGPR(e)), "s"(ob_n), "s"(W1), "s"(W2));
        if (FOUR) asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(ob3) : "s"(W3), "s"(ob_n));
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (keeps `lane < 3` from becoming one more hoisted, spilled mask)
        uint32_t pk0_v = 0;            // kMtRegs: lanes 1..3, the re-queued entry's key without its nonce
        if (ln < (FOUR ? 4 : 3)) {
            const uint32_t off = w_v & 255u, val = w_v >> 8;
            // (buffer stores into this stream's state: field offsets in scalar registers instead of 64-bit pointers added per lane)
            const uint32_t loc = (uint32_t)(p * 256) | off;
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)val, rsrc_s, (int)(loc * 2u), up_off, 0);   // byte_pair_difference == store-table value (screen.py:383-398); <= 2047: the 16-bit copy
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)c, rsrc_s, (int)loc, mem_off, 0);
            // priority 0 -> out of the live set; all ones for a byte that keeps a priority
            const uint32_t gone_bit = (val == 0u ? 1u : 0u) << (off & 31u);
            atomicAnd(&nz[loc >> 5], ~gone_bit);
            atomicOr(&pdone[p * 8 + (x >> 5)], 1u << (x & 31));   // (the primary's diff weight counts as 0 from here on; from all lanes: idempotent)
            // the re-queued entry (video.py:178), its nonce the next word behind the candidates'; a byte that is not
            // re-queued aims beyond the buffer
            if constexpr (!kMtRegs) {
                const uint32_t nonce = mt_temper(mt[mt_idx + C + (int)k_v]) >> 24;
                const uint32_t pkey = ((2047u - val) << 21) | (nonce << 13) | loc;
                const uint32_t slot = val != 0u ? (uint32_t)(n_pushed + (int)k_v) * 4u : 0x7ffffff0u;
                __builtin_amdgcn_raw_buffer_store_b32(pkey, rsrc_s, (int)slot, (int)offsetof(StreamState, pushed), 0);
                if (decltype(track)::value) pkey_v = pkey;
            } else {
                pk0_v = ((2047u - val) << 21) | loc;   // the nonce later: the entry goes into the register queue below
            }
        }
        if constexpr (kMtRegs) {
            const int first = mt_behind * 624 + mt_idx + C;   // (counted from the block in the registers)
            mt_enqueue(pk0_v, 1, f1, 0, first);
            mt_enqueue(pk0_v, 2, f2, f1, first);
            if (FOUR) mt_enqueue(pk0_v, 3, f3, f1 + f2, first);
            n_pend += f1 + f2 + f3;
        }
        if (decltype(track)::value) push_f1 = f1, push_f2 = f2, push_f3 = f3, push_base = n_pushed;
        mt_idx += C + f1 + f2 + f3;
        draws += (uint32_t)(C + f1 + f2 + f3);
        n_pushed += f1 + f2 + f3;
        if constexpr (kMtRegs) {
            if (decltype(track)::value) {
                // phase B wants the finished keys now (lane q = this step's q-th push; its queue holds nothing else): the
                // current block's, and if the draws reached beyond it, one block on for the others
                pkey_v = mt_service(0);
                if (n_pend > 0) {
                    const int r = (FOUR ? f1 + f2 + f3 : f1 + f2) - n_pend;   // finished in the first round
                    const uint32_t k2 = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - r) & 63) << 2, (int)mt_service(-1));
                    pkey_v = lane < r ? pkey_v : k2;
                }
            } else if (__builtin_expect(n_pend > kQueueHigh, 0)) {
                (void)mt_service(0);
            }
        }
        done++;
        ob_n++;
        if (__builtin_expect(ob_n == 64, 0)) flush_ops();
        if (__builtin_expect(mt_idx >= 624, 0)) {
            if constexpr (kMtRegs) {
                mt_idx -= 624;   // (the registers follow when somebody needs them: mt_service)
                mt_behind++;
            } else {
                // the next block becomes the current one: its head moves down now, the rest of it
                // and the new head are generated later, while table loads are in flight
                // (twist_now), at the latest before a step reads past word 255
                move_head();
                mt_idx -= 624;
                twist_pending = true;
            }
        }
    };

    // One greedy step on list entry e = page << 8 | offset | content << 16 with what was loaded
    // for it.  Returns false if the entry's priority is gone (video.py:130: nothing happens);
    // otherwise an opcode is emitted or err is set.
    // (nzw, pdw: this lane's words of the page's two bitmaps, read by the caller as early as the previous step's updates allow --
    // the step's first decision hangs on them, and an LDS round trip at its start is a stall at raised priority)
    auto step = [&](auto track, uint32_t e, const Loaded &L, uint32_t nzw, uint32_t pdw) -> bool {
        const int x = e & 255;
        const uint32_t c = (e >> 16) & 0xffu;  // video.py:134
        const uint32_t xword = (uint32_t)__builtin_amdgcn_readlane((int)nzw, (x >> 5) * 8);
        if (__builtin_expect(!((xword >> (x & 31)) & 1u), 0)) return false;
        if (MODE == kDHGR && c >= 0x80) {  // video.py:137
            err = kErrPaletteBit;
            return true;
        }
        // x itself leaves both sets before the page is scored (video.py:140-141)
        const uint32_t xbit = wsel == (x >> 5) ? 1u << (x & 31) : 0u;
        nzw &= ~xbit;
        pdw |= xbit;
        // per byte: d = delta << 20 | y (screen.py:547); kt = d for bytes whose diff weight
        // still counts (not yet a primary, video.py:141), ke = d for bytes that may still be
        // chosen (update_priority != 0, video.py:159); >= 0 otherwise
        uint32_t nd[4];
        int ke[4];
        int C = 0;
        unsigned long long cand[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            nd[r] = L.gl[r] + L.gr[r];   // L1 + RF = store value + kNarrowBias (iiv_stream.h: narrow form)
            // delta << 20 | store value << 8 | offset: the value (11 bits) rides in the key's spare bits, so a winner's key
            // brings it along (one multiply-add: nd * (2^20 + 2^8) + kb, kb = -diff weight << 20 | y less the bias term)
            const int d = (int)(__umul24(nd[r], kKeyMul) + L.kb[r]);
            const int gone = __builtin_amdgcn_sbfe((int)pdw, sh0 + r, 1);
            const int live = __builtin_amdgcn_sbfe((int)nzw, sh0 + r, 1);
            ke[r] = d & live;
            cand[r] = __ballot((d & ~gone) < 0);  // video.py:283
            C += (int)__popcll(cand[r]);           // one nonce each (video.py:290-293)
        }
        // two smallest eligible keys: in the lane, then across the wave
        const int a0 = ke[0] < ke[1] ? ke[0] : ke[1], b0 = ke[0] < ke[1] ? ke[1] : ke[0];
        const int a1 = ke[2] < ke[3] ? ke[2] : ke[3], b1 = ke[2] < ke[3] ? ke[3] : ke[2];
        const int k1 = a0 < a1 ? a0 : a1, hi01 = a0 < a1 ? a1 : a0, mb = b0 < b1 ? b0 : b1;
        const int k2 = hi01 < mb ? hi01 : mb;
        const int k3 = hi01 < mb ? mb : hi01;   // (FOUR: the lane's third smallest)
        // (a fast-path key carries its byte's store value in bits 8..18; the exact path's keys have no room for it: there it is
        // read out of the lanes, packed in pairs)
        auto nd_of = [&](int y) -> uint32_t {
            const uint32_t nd01 = nd[0] | (nd[1] << 16), nd23 = nd[2] | (nd[3] << 16);
            const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)nd01, y >> 2);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)nd23, y >> 2);
            return ((((y & 2) ? pb : pa) >> ((y & 1) * 16)) & 0xffffu) - kNarrowBias;
        };
        // The exact path below is complete by itself (it orders every eligible byte by (delta, nonce, offset)); the fast
        // path in front of it only pays where ties are rare.  On picture-like input they are the rule (96 % of the
        // steps), so a step that follows a tie goes straight to the exact path.
        // the winners as `store value << 8 | offset`: bits 0..18 of a fast-path key; a missing one repeats the primary (x, value 0)
        uint32_t W1 = (uint32_t)x, W2 = (uint32_t)x, W3 = (uint32_t)x;
        bool tie = prev_tie;
        // does another eligible byte share the delta of the last winner K?  (then the nonces decide.)  x = key ^ K is 0 for K
        // itself, below 2^20 for a byte with K's delta, and has its sign bit set for a byte that is not eligible: one v_xad_u32
        // (x - 1) per byte, a minimum, one compare -- no ballots to count
        auto shared_delta = [&](int K) -> bool {
            uint32_t m = ((uint32_t)(ke[0] ^ K)) - 1u;
#pragma unroll
            for (int r = 1; r < 4; r++) {
                const uint32_t x1 = ((uint32_t)(ke[r] ^ K)) - 1u;
                m = x1 < m ? x1 : m;
            }
            return __ballot(m < (1u << kWdDwShift) - 1u) != 0ull;
        };
#ifdef IIV_STAMPS
        {   // diagnostic build: how often the nonces decide, and among how many bytes -- computed beside the step, whatever path it takes
            int D1 = k1, D2 = k2;
            wave_top2_i32(D1, D2);
            if (D1 < 0 && D2 < 0 && ((D1 >> kWdDwShift) == (D2 >> kWdDwShift) || shared_delta(D2))) {
                int n1 = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) n1 += (int)__popcll(__ballot(ke[r] < 0 && (ke[r] >> kWdDwShift) == (D1 >> kWdDwShift)));
                n_ties++;
                n_tie_members += n1;
                if (n1 <= 2) n_ties_small++;
                // what the nonces have to order: the bytes at the smallest delta if two or more share it, else those at the second
                const int dR = n1 >= 2 ? (D1 >> kWdDwShift) : (D2 >> kWdDwShift);
                int lane_cnt = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) lane_cnt += (ke[r] < 0 && (ke[r] >> kWdDwShift) == dR) ? 1 : 0;
                const unsigned long long b1 = __ballot(lane_cnt >= 1), b2 = __ballot(lane_cnt >= 2), b3 = __ballot(lane_cnt >= 3), b4 = __ballot(lane_cnt >= 4);
                n_rel += IIV_SGPR(__popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4));
                n_rel_single += IIV_SGPR(b2 == 0ull ? 1 : 0);
                n_rel_double += IIV_SGPR((b2 != 0ull && b3 == 0ull) ? 1 : 0);
            }
        }
#endif
        // the wave's two smallest eligible keys in one fused-DPP pass (iiv_wave.h)
        int K1 = k1, K2 = k2;
        if (!tie) wave_top2_i32(K1, K2); else K1 = K2 = 0;
        if (__builtin_expect(K1 < 0, 1)) {
            W1 = (uint32_t)K1 & 0x7ffffu;
            if (__builtin_expect(K2 < 0, 1)) {
                W2 = (uint32_t)K2 & 0x7ffffu;
                tie = (K1 >> kWdDwShift) == (K2 >> kWdDwShift);
                if constexpr (!FOUR) {
                    if (!tie) tie = shared_delta(K2);
                } else if (!tie) {
                    // the third winner: every lane's smallest key above K2 (a lane may have held K1, K2 or both)
                    const int c3 = k1 > K2 ? k1 : (k2 > K2 ? k2 : k3);
                    const int K3 = wave_min_i32(c3);
                    if (K3 < 0) {
                        W3 = (uint32_t)K3 & 0x7ffffu;
                        tie = (K2 >> kWdDwShift) == (K3 >> kWdDwShift) || shared_delta(K3);
                    }
                }
            }
        }
        // (the LDS-shared form is dispatched to input whose steps the nonces rarely decide: lay the exact path out of line there)
        if (W > 1 ? __builtin_expect(tie, 0) : tie) {
            // the reference's (delta, nonce, offset) heap order with every candidate's nonce
            // materialised: one random.getrandbits(8) per candidate in ascending offset
            // (video.py:290-293)
            twist_now();
            n_exact++;
            // (on picture-like input this is the normal path: 96 % of the opcodes of S-img tie, with 11 bytes sharing the
            // smallest delta; on random input 2.5 %)
            // keys: delta (signed, top 16 bits) | nonce | offset.  An eligible byte's delta is negative, so its key is; the
            // others keep whatever non-negative key their bits make -- no bias, no select: signed minima find the eligible ones
            // first, and a winner that is not negative is "none"
            static_assert(kWdDwShift == 20, "the key's delta field is taken from bits 20..31 of an eligible key");
            int key[4];
            // candidates in lower lanes draw first, then this lane's bytes in ascending order
            int run = mt_idx;
#pragma unroll
            for (int q = 0; q < 4; q++) run += prefix_popc(cand[q]);
            uint32_t word[4];
            if constexpr (kMtRegs) {
                // the candidates' draws mt_idx .. mt_idx + C - 1 out of the registers: the current block's share, then -- if they
                // reach beyond it -- the queue resolved, the next block made, and its share
                int at[4];
                bool mine[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    at[r] = run;
                    mine[r] = (cand[r] >> lane) & 1ull;
                    run += mine[r] ? 1 : 0;
                }
                const int last = mt_idx + (C > 0 ? C - 1 : 0), last0 = last < 623 ? last : 623;
                if (mt_behind != 0) (void)mt_service(0);
#pragma unroll
                for (int r = 0; r < 4; r++) word[r] = mt_regs_gather(mtr, mine[r] && at[r] < 624 ? at[r] : mt_idx, mt_idx, last0);
                if (__builtin_expect(last >= 624, 0)) {
                    (void)mt_service(-1);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t w2 = mt_regs_gather(mtr, mine[r] && at[r] >= 624 ? at[r] - 624 : 0, 0, last - 624);
                        word[r] = at[r] >= 624 ? w2 : word[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    word[r] = mt[run];
                    run += (int)((cand[r] >> lane) & 1ull);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t nonce = mt_temper(word[r]) >> 24;
                key[r] = (int)(((uint32_t)(ke[r] >> 4) & 0xffff0000u) | (nonce << 8) | (y0 + r));   // video.py:159
            }
            // the two smallest (delta, nonce, offset): in the lane, then the one-pass fused-DPP top-2 of the fast path
            // (keys are unique -- they end in the offset)
            const int ta0 = key[0] < key[1] ? key[0] : key[1], tb0 = key[0] < key[1] ? key[1] : key[0];
            const int ta1 = key[2] < key[3] ? key[2] : key[3], tb1 = key[2] < key[3] ? key[3] : key[2];
            const int t1 = ta0 < ta1 ? ta0 : ta1;
            const int thi = ta0 < ta1 ? ta1 : ta0, tmb = tb0 < tb1 ? tb0 : tb1;
            const int t2 = thi < tmb ? thi : tmb;
            int T1 = t1, T2 = t2;
            wave_top2_i32(T1, T2);   // (keys are unique: they end in the offset)
            W1 = T1 < 0 ? (nd_of(T1 & 255) << 8) | (uint32_t)(T1 & 255) : (uint32_t)x;
            W2 = T2 < 0 ? (nd_of(T2 & 255) << 8) | (uint32_t)(T2 & 255) : (uint32_t)x;
            prev_tie = T2 < 0 && (T1 >> 16) == (T2 >> 16);   // (a prediction only: either path is exact)
            if constexpr (FOUR) {
                const int t3 = thi < tmb ? tmb : thi;
                const int T3 = wave_min_i32(t1 > T2 ? t1 : (t2 > T2 ? t2 : t3));   // every lane's smallest key above T2
                W3 = T3 < 0 ? (nd_of(T3 & 255) << 8) | (uint32_t)(T3 & 255) : (uint32_t)x;
                prev_tie = prev_tie || (T3 < 0 && (T2 >> 16) == (T3 >> 16));
            }
        }
        apply(track, e, W1, W2, W3, C);
        return true;
    };

    int guard = n_ops + 8192 + 2 * kPushedCap + 64;
#ifdef IIV_STAMPS
    // diagnostic build: shader clocks of this wave per pipeline stage, and its start / end time
    unsigned long long ph[4] = {0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
    const unsigned long long wave_t0 = ph_t, wave_r0 = __builtin_amdgcn_s_memrealtime();
    int n_steps = 0;
#define IIV_PHASE(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph[i] += now_ - ph_t;                                         \
        ph_t = now_;                                                  \
    } while (0)
#else
#define IIV_PHASE(i) do { } while (0)
#endif

    // ---- phase A: the sorted initial list (video.py:121-131), pipelined.
    // The list is read 64 entries at a time (the next window's words are requested one window
    // ahead); the entries of a window whose priority is still non-zero are compacted into one
    // register (lane k = k-th live entry, through a 256 B LDS scatter), so that taking the
    // next entry is one v_readlane with a scalar index.
    // window entries: page << 8 | offset | content << 16 | (list position - win_base) << 24
    if (head < n_sorted && !exhausted) {
        int win_base = head, win_end = head;  // list positions [win_base, win_end) are in the window
        int n_dense = 0, qi = 0;               // live entries of the window, next one to hand out
        uint32_t dense_e = 0;
        uint32_t ord_next = S.order[(head + lane) & 8191];  // the words of [win_end, win_end + 64)
        auto refill = [&]() {
            const int start = win_end;
            const uint32_t e = ord_next;
            const int idx = start + lane;
            ord_next = S.order[(idx + 64) & 8191];   // (unconditional, see below; positions >= n_sorted are never used)
            const uint32_t loc = e & 0x1fffu;
            const bool v = idx < n_sorted && ((nz[loc >> 5] >> (loc & 31)) & 1u);
            const unsigned long long mask = __ballot(v);
            n_dense = (int)__popcll(mask);
            qi = 0;
            // compaction across lanes (no LDS memory): live entries go to lanes 0 .. n_dense - 1 in list order,
            // the others fill the lanes behind them
            const int dst = v ? prefix_popc(mask) : n_dense + prefix_popc(~mask);
            dense_e = (uint32_t)__builtin_amdgcn_ds_permute(dst << 2, (int)((e & 0x00ffffffu) | ((uint32_t)lane << 24)));
            // (the permute's result is waited for HERE, once per window: left pending, the compiler puts an lgkmcnt(0) in front of
            // every take's v_readlane of this register -- behind the table reads a half-iteration has just issued)
            asm volatile("" : "+v"(dense_e));
            win_base = start;
            win_end = start + 64 < n_sorted ? start + 64 : n_sorted;
        };
        // next live entry, or 0 when the list is used up; *next_head = list position after it
        auto take = [&](int &next_head) -> uint32_t {
            for (;;) {
                while (__builtin_expect(qi >= n_dense, 0)) {
                    if (win_end >= n_sorted) return 0u;
                    refill();
                }
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)dense_e, qi++);
                // An entry can die between the compaction of its window and its turn (a step resolved its
                // byte exactly, video.py:159-170): it would cost a full pipeline slot -- row, eight table
                // loads, a step that finds it dead.  On image-like input that is 0.6 slots per opcode
                // (tools/greedy_phases.py); one LDS word per entry taken avoids them.
                // (The LDS-shared form runs on input whose steps the nonces rarely decide -- iiv_encode.hip: shared_form_now --
                // and there hardly an entry dies in that interval (S-iid: none, S-coh / S-static: IIV_TAKE_CHECK_SHARED=1 measured);
                // a take is on every step's critical path and the round trip costs that form 4 %: it does without.)
                if constexpr ((W == 1 && IIV_TAKE_CHECK_PLAIN) || (W > 1 && IIV_TAKE_CHECK_SHARED)) {
                    const uint32_t loc = e & 0x1fffu;
                    const uint32_t w = IIV_SGPR(nz[loc >> 5]);
                    if (!((w >> (loc & 31)) & 1u)) continue;
                }
                next_head = win_base + (int)(e >> 24) + 1;
                return e | 0x80000000u;  // (bit 31 marks a real entry: page 0 / offset 0 / content 0 is a valid one)
            }
        };
        auto row_of = [&](uint32_t e) -> uint4 { return wd_rows[((e >> 8) & 31) * 64 + lane]; };

        // pipeline: entry A = next to be scored (its set is loaded or in flight), B = the one
        // after it (its row has arrived or is about to), C = the one after that (row in flight).
        // One half-iteration: issue B's eight table loads into the other set, take entry D and
        // request its row into the row buffer B's loads have just consumed, then score A.
        // Rows are requested two entries ahead of their use so that the wait for a row never
        // has to cover the stores of the step before it: vmcnt counts loads and stores in one
        // in-order queue, and between a row's load and its use there are always the eight table
        // loads of another entry, whatever the steps in between stored.
        // (Loads are issued unconditionally -- a missing entry reads page 0's row and content
        // 0's slices, which is harmless -- because a conditionally assigned register would be
        // merged with a copy, and the copy would wait for the load on the spot.)
        uint32_t eA, eB, eC;
        int hA = head, hB, hC;
        uint4 rowP, rowQ;
        Loaded set0, set1;
        // (the prelude issues its loads in the loop's own order -- row, eight table loads, row --
        // so that the wait counts the compiler derives at the loop header are the loop's)
        eA = take(hA);
        const uint4 rowA = row_of(eA);
        hB = hA;
        eB = eA ? take(hB) : 0u;
        rowP = row_of(eB);
        gather8(rowA, eA >> 16 & 0xffu, set0);
        hC = hB;
        eC = eB ? take(hC) : 0u;
        rowQ = row_of(eC);
        // `active` = false: the loop is about to end (the first half of this trip was the last
        // step); the loads are still issued -- every path from a row's load to the loop header
        // then crosses eight table loads and another row load, which is what keeps the
        // compiler's wait for the row from covering the stores of the latest step -- but
        // nothing is scored.
        auto half = [&](const Loaded &cur, Loaded &nxt, uint4 &row, bool active) -> bool {
#ifdef IIV_STAMPS
            asm volatile("" : "+v"(row.x), "+v"(row.y), "+v"(row.z), "+v"(row.w));   // (the row's arrival, timed on its own)
            IIV_PHASE(3);
#endif
            // A's bitmap words (the previous step's updates are in program order behind us; nothing up to step() changes them)
            const uint32_t nzA = nz[((eA >> 8) & 31) * 8 + wsel], pdA = pdone[((eA >> 8) & 31) * 8 + wsel];
            gather8(row, eB >> 16 & 0xffu, nxt);
            // every use of the old row is scheduled before the new one is requested, so that the
            // load can land in the same registers (otherwise: a copy, and a wait in front of it)
            __builtin_amdgcn_sched_barrier(0);
            int hD = hC;
            const uint32_t eD = (active && eC) ? take(hD) : 0u;
            row = row_of(eD);
            IIV_PHASE(0);   // wait for the row, issue eight table loads, take an entry, request its row
            if (__builtin_expect(!active, 0)) return false;
            twist_now();  // (the MT19937 block generation hides behind the loads)
            IIV_PHASE(1);   // MT19937 block generation
#ifdef IIV_STAMPS
            n_steps++;
#endif
            // A wave that has its table words scores and applies at raised priority: its step is a chain of dependent
            // reductions and scalar bookkeeping, and every cycle it waits for an issue slot behind waves that are only
            // issuing loads is a cycle its own next loads start later (+1 % DHGR, +1.2 % HGR; raising the load issue
            // instead costs HGR 5 %).
            __builtin_amdgcn_s_setprio(IIV_STEP_PRIO);
            (void)step(std::false_type{}, eA, cur, nzA, pdA);
            __builtin_amdgcn_s_setprio(0);
            IIV_PHASE(2);   // wait for the table words, score, apply
            head = hA;
            eA = eB;
            hA = hB;
            eB = eC;
            hB = hC;
            eC = eD;
            hC = hD;
            if (__builtin_expect(--guard < 0, 0)) err = kErrGuard;
            return eA && done < n_ops && !err;
        };
        if (eA && done < n_ops) {
            bool more;
            do {
                more = half(set0, set1, rowP, true);
                more = half(set1, set0, rowQ, more);
            } while (more);
        }
        if (!eA && !err) head = n_sorted;  // every entry of the list has been processed or was dead
    }

    // ---- phase B: the re-queued bag (video.py:124-131, 170-178), one entry at a time.
    // Entry i of pushed[] belongs to sub-bag i % 64, and lane l keeps the smallest key of sub-bag l (and where it
    // is) in a register: a pop is a wave minimum over that register; only the sub-bag it came from is read again
    // (by all lanes, its <= kPushedCap / 64 = 384 entries: up to kSub loads each, requested before the step and used after it);
    // a push is a comparison with one lane's minimum.  (It was a scan of the whole bag per pop: on input that
    // converges -- a static background -- every opcode comes from here, and the bag holds two entries per
    // opcode emitted so far.)
    if (done < n_ops && !err && !exhausted) {
        if constexpr (kMtRegs) (void)mt_service(0);   // (every queued key into pushed[]: it is read back from here on)
        uint32_t ck = INF, ci = 0;
        for (int i = lane; i < n_pushed; i += 64) {
            const uint32_t k = S.pushed[i];
            ci = k < ck ? (uint32_t)i : ci;
            ck = k < ck ? k : ck;
        }
        // (gfx9 VOP3 reads one SGPR only; v_writelane may take its lane select from m0 besides)
        auto writelane = [&](uint32_t &v, uint32_t val, int l) {
            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(IIV_SGPR(val)), "s"(IIV_SGPR(l)));
        };
        auto wave_min_u32 = [&](uint32_t v) -> uint32_t { return (uint32_t)wave_min_i32((int)(v ^ 0x80000000u)) ^ 0x80000000u; };
        auto cache_merge = [&](uint32_t key, uint32_t idx) {   // scalars
            const int sub = (int)(idx & 63u);
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)ck, sub);
            if (key < cur) {
                writelane(ck, key, sub);
                writelane(ci, idx, sub);
            }
        };
        while (done < n_ops && !err) {
            if (--guard < 0) {
                err = kErrGuard;
                break;
            }
            if (truncated) {  // more initial entries exist than were ordered: host budget bug
                err = kErrSortBudget;
                break;
            }
            const uint32_t bk = wave_min_u32(ck);
            if (bk == INF) {
                exhausted = 1;  // video.py:189
                break;
            }
            const int wl = (int)__builtin_ctzll(__ballot(ck == bk));   // (equal keys are equal heap tuples: any of them)
            const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)ci, wl);
            if (lane == 0) S.pushed[bi] = INF;
            // what is left of that sub-bag: entries wl + 64 j (the popped one excluded by its index, the store
            // above is for later visits)
            // (sized from the cap: with the fourth offset a generator can push 3 entries per step, and an entry at an
            // index >= 4 * 4096 that the rescan did not reach would be dropped from its sub-bag's minimum)
            constexpr int kSub = (kPushedCap + 4095) / 4096;
            static_assert(kPushedCap <= 64 * 64 * kSub, "the rescan of a sub-bag must reach every index below kPushedCap");
            uint32_t rk[kSub], ridx[kSub];
#pragma unroll
            for (int t = 0; t < kSub; t++) {
                ridx[t] = (uint32_t)wl + 64u * (uint32_t)(lane + 64 * t);
                rk[t] = INF;
                if (ridx[t] < (uint32_t)n_pushed && ridx[t] != bi) rk[t] = S.pushed[ridx[t]];
            }
            // pushed keys do not carry the content byte: it is the target byte of that offset
            const uint32_t loc = bk & 0x1fffu;
            const uint32_t c = (uint32_t)IIV_SGPR(tgt_frames[loc]);
            const uint32_t e = loc | (c << 16);
            const uint4 w = wd_rows[((e >> 8) & 31) * 64 + lane];
            Loaded L;
            gather8(w, c, L);
            twist_now();
            push_f1 = push_f2 = push_f3 = 0;
            (void)step(std::true_type{}, e, L, nz[((e >> 8) & 31) * 8 + wsel], pdone[((e >> 8) & 31) * 8 + wsel]);
            // the sub-bag's new minimum, then what this step pushed
            uint32_t mk = rk[0], mi = ridx[0];
#pragma unroll
            for (int t = 1; t < kSub; t++) {
                mi = rk[t] < mk ? ridx[t] : mi;
                mk = rk[t] < mk ? rk[t] : mk;
            }
            const uint32_t nk = wave_min_u32(mk);
            const uint32_t ni = (uint32_t)__builtin_amdgcn_readlane((int)mi, (int)__builtin_ctzll(__ballot(mk == nk)));
            writelane(ck, nk, wl);
            writelane(ci, ni, wl);
            // (pkey_v: the pushed keys in lanes 1, 2, 3 by winner -- kMtRegs: in lanes 0, 1, 2 by push)
            if (push_f1) cache_merge((uint32_t)__builtin_amdgcn_readlane((int)pkey_v, kMtRegs ? 0 : 1), (uint32_t)push_base);
            if (push_f2) cache_merge((uint32_t)__builtin_amdgcn_readlane((int)pkey_v, kMtRegs ? push_f1 : 2), (uint32_t)(push_base + push_f1));
            if (FOUR && push_f3) cache_merge((uint32_t)__builtin_amdgcn_readlane((int)pkey_v, kMtRegs ? push_f1 + push_f2 : 3), (uint32_t)(push_base + push_f1 + push_f2));
        }
    }

    flush_ops();
    if (exhausted && done < n_ops && !err) {
        for (int i = done + lane; i < n_ops; i += 64) {  // video.py:249-251
            uint8_t *q = out + (size_t)i * 6;
            q[0] = 32; q[1] = (uint8_t)pad_content; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0;
        }
        pad_ops += (unsigned long long)(n_ops - done);
        done = n_ops;
    }
    twist_now();
    if constexpr (kMtRegs) (void)mt_service(0);   // (the registers at the block mt_idx counts in, every queued key stored)
    if (W == 1) __syncthreads(); else wave_lds_sync();
    for (int i = lane; i < 256; i += 64) {
        S.nzbits[i] = nz[i];
        S.pdone[i] = pdone[i];
    }
    if constexpr (kMtRegs) mt_regs_store(mtr, S.mt_py, lane);
    else
        for (int i = lane; i < 624; i += 64) S.mt_py[i] = mt[i];
    if (lane == 0) {
        S.mt_py_idx = mt_idx;
        S.head = head;
        S.n_pushed = n_pushed;
        S.exhausted = exhausted;
        if (exhausted) S.out_of_work[is_aux] = 1;
        S.draws_py += (unsigned long long)draws;
        S.ops += (unsigned long long)done;
        S.pad_ops += pad_ops;
        if (err && S.error == 0) S.error = err;
        if (cost) cost[stream] = (uint32_t)(__builtin_amdgcn_s_memtime() - clk0);   // what this launch took this stream: the next launches' order (GreedyArgs::perm)
        if (count_stats) {   // (per stream, beside its other counters: 14336 x 3 atomics on one address per launch cost 7 % of it)
            S.stat_exact += (unsigned long long)n_exact;
            S.stat_ops += (unsigned long long)(done - (int)pad_ops);
            S.stat_runs += 1ull;
        }
#ifdef IIV_STAMPS
        for (int i = 0; i < 4; i++) S.stamps[16 + i] = ph[i];
        S.stamps[24] = wave_t0;
        S.stamps[25] = __builtin_amdgcn_s_memtime();
        S.stamps[26] = (unsigned long long)done;
        S.stamps[27] = (unsigned long long)n_steps;   // list entries that went through the pipeline (emitted or found dead)
        S.stamps[29] = wave_r0;   // start / end on the constant 100 MHz counter all XCDs share
        S.stamps[30] = __builtin_amdgcn_s_memrealtime();
        S.stamps[31] = (unsigned long long)n_ties | ((unsigned long long)n_ties_small << 20) | ((unsigned long long)n_tie_members << 40);
        S.stamps[20] = (unsigned long long)n_rel;
        S.stamps[21] = (unsigned long long)n_rel_single | ((unsigned long long)n_rel_double << 32);
        S.stamps[28] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 /* XCC_ID */) | (0 << 6) | (31 << 11)) << 32) |
                       (unsigned)__builtin_amdgcn_s_getreg((4 /* HW_ID */) | (0 << 6) | (31 << 11));
#endif
    }
    };   // run_stream

    if constexpr (W == 1) {
        run_stream(perm ? IIV_SGPR(perm[blockIdx.x]) : (int)blockIdx.x, lane0);
    } else {
        for (;;) {
            int next = 0;
            if (lane0 == 0) next = atomicAdd(queue, 1);
            next = IIV_SGPR(next);
            if (next >= n_streams) break;
            if (perm) next = IIV_SGPR(perm[next]);
            // (the lane index is laundered per stream: values derived from it would otherwise be hoisted out of this
            // loop and held in registers across it -- 96 VGPRs instead of 78)
            int lane_i = lane0;
            asm volatile("" : "+v"(lane_i));
            run_stream(next, lane_i);
            wave_lds_sync();   // (this wave's LDS is reused by its next stream)
        }
    }
}
#undef IIV_PHASE

// the LDS-shared form: persistent workgroups -- as many as are resident at once -- whose waves take streams off a queue
template <int MODE, bool FOUR> static int launch_shared(const GreedyArgs &a, hipStream_t st)
{
    using SC = SharedCfg<MODE>;
    // workgroups resident at once, per DEVICE (a process may drive several GPUs; the LDS attribute is per device too) and
    // per instantiation; encoders on different host threads may get here together
    constexpr int kMaxDev = 64;
    static int resident_of[kMaxDev] = {};
    static std::mutex mu;
    int dev = 0;
    if (hip_check(hipGetDevice(&dev), "hipGetDevice")) return IIV_ERR_HIP;
    if (dev < 0 || dev >= kMaxDev) return set_error(IIV_ERR_INVALID, "greedy_wave_kernel: device index %d", dev);
    int resident;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!resident_of[dev]) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            if (hip_check(hipFuncSetAttribute((const void *)greedy_wave_kernel<MODE, SC::kW, FOUR>, hipFuncAttributeMaxDynamicSharedMemorySize, SC::kLds),
                          "greedy_wave_kernel LDS attribute") ||
                hip_check(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)greedy_wave_kernel<MODE, SC::kW, FOUR>, 64 * SC::kW, (size_t)SC::kLds),
                          "greedy_wave_kernel occupancy") ||
                hip_check(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties"))
                return IIV_ERR_HIP;
            resident_of[dev] = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
        }
        resident = resident_of[dev];
    }
    const int wgs = (a.n_streams + SC::kW - 1) / SC::kW;
    hipLaunchKernelGGL((greedy_wave_kernel<MODE, SC::kW, FOUR>), dim3(wgs < resident ? wgs : resident), dim3(64 * SC::kW), (size_t)SC::kLds, st, a.states,
                       a.frames_main, a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.n_streams, a.uniform_bank, a.queue, a.count_stats ? 1 : 0, a.perm, a.cost);
    return IIV_OK;
}

int launch_greedy_wave(int mode, const GreedyArgs &a, hipStream_t st, int *form_out)
{
    // (the shared form needs every stream of the round on one bank -- always so in HGR -- and the host's stream counter)
    const bool shared = a.shared && a.uniform_bank >= 0 && a.queue && a.lds_pad == 0;
    if (form_out) *form_out = shared ? 1 : 0;
    int rc = IIV_OK;
    if (shared && a.fourth)
        rc = mode == kDHGR ? launch_shared<kDHGR, true>(a, st) : launch_shared<kHGR, true>(a, st);
    else if (a.fourth && mode == kDHGR)   // (f4: a real fourth offset per opcode -- the plain one-wave form only)
        hipLaunchKernelGGL((greedy_wave_kernel<kDHGR, 1, true>), dim3(a.n_streams), dim3(64), (size_t)a.lds_pad, st, a.states, a.frames_main,
                           a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.n_streams, a.uniform_bank, (int *)nullptr, a.count_stats ? 1 : 0, a.perm, a.cost);
    else if (a.fourth)
        hipLaunchKernelGGL((greedy_wave_kernel<kHGR, 1, true>), dim3(a.n_streams), dim3(64), (size_t)a.lds_pad, st, a.states, a.frames_main,
                           a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.n_streams, a.uniform_bank, (int *)nullptr, a.count_stats ? 1 : 0, a.perm, a.cost);
    else if (shared)
        rc = mode == kDHGR ? launch_shared<kDHGR, false>(a, st) : launch_shared<kHGR, false>(a, st);
    else if (mode == kDHGR)
        hipLaunchKernelGGL((greedy_wave_kernel<kDHGR, 1>), dim3(a.n_streams), dim3(64), (size_t)a.lds_pad, st, a.states, a.frames_main,
                           a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.n_streams, a.uniform_bank, (int *)nullptr, a.count_stats ? 1 : 0, a.perm, a.cost);
    else
        hipLaunchKernelGGL((greedy_wave_kernel<kHGR, 1>), dim3(a.n_streams), dim3(64), (size_t)a.lds_pad, st, a.states, a.frames_main,
                           a.frames_aux, a.n_frames, a.segs, a.seg_stride, a.nt, a.ops_out, a.ops_stride, a.n_streams, a.uniform_bank, (int *)nullptr, a.count_stats ? 1 : 0, a.perm, a.cost);
    if (rc) return rc;
    return hip_check(hipGetLastError(), "greedy_wave_kernel launch");
}

}  // namespace iiv
