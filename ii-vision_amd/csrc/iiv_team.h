// iiv_team.h -- the greedy selection loop (transcoder/video.py:121-187, 275-301;
// transcoder/screen.py:256-293) for FEW streams: one workgroup of W waves per stream.
//
// One stream is a chain of dependent steps, and one wave needs ~1.2 us for a step (row load,
// eight table loads, two wave reductions, the stores): a single clip runs at that latency
// whatever the rest of the GPU does.  But a step reads and writes only the row of ITS page
// (video.py:147-182: compute_delta_page, apply and the priorities it touches all lie on
// `page`), so the next entries of the sorted list are independent of each other as long as
// they lie on different pages -- except for three scalars: how many random.getrandbits(8)
// the earlier steps drew (one per candidate, video.py:290-293, plus one per re-queued byte,
// video.py:178), how many entries they pushed, and the position of their opcode in the
// output.  All three are prefix sums of numbers each step knows after scoring.
//
// A round: every wave takes one of the next <= W list entries (the run stops in front of the
// first entry whose page is already in the run), scores it on its own -- same arithmetic as
// greedy_wave_kernel, split store table included -- and publishes (candidates, winners,
// values, tie) in LDS; after a barrier every wave reads all results, forms the prefix sums,
// and commits its own entry: stores, bitmap updates, re-queued entries with the nonces at its
// own offset of the random stream, its opcode at its own position.  A step whose winners
// depend on the nonces (a tie, 2.7 % of the steps) ends the run: it is resolved with the
// nonces at its offset, and what was scored behind it is scored again next round.
// Seven waves score; the eighth keeps random's MT19937 stream ahead: the state lives in LDS as a
// ring of ten 624-word blocks (a round can draw 7 * 258 words), and while the others score, that
// wave twists up to two more blocks into the slots the stream has left behind.  (Two such waves, half of a block's chains
// each, were measured in round 5: no faster -- a ninth wave shares a SIMD with two others -- and removed; so was that wave
// skipping the run formation and picking the run's length up from LDS: the wait at barrier B fell from 590 to 340 clocks, the
// round from 6 447 to 6 406.)
// The table words of the entry a wave will most likely score next round (the one 7 positions
// further, if the whole run commits) are requested before this round's commit, so that they
// travel while the stores, the barrier and the bookkeeping run.
// Exact: same opcodes, same state, same RNG positions as the one-wave kernel and the reference.
//
// When the sorted list is used up, wave 0 goes on alone through the re-queued bag
// (video.py:124-131, 170-178), one entry at a time.
#pragma once
#include <type_traits>

#include "iiv_host.h"
#include "iiv_stream.h"
#include "iiv_wave.h"

namespace iiv {

constexpr unsigned long long kLiveEnd = 0xffull;   // byte 0 of a live-queue slot: "the launch ended here, short of its n_ops" (a page byte is 32..63)
// one slot of the live queue: a relaxed atomic store at SYSTEM scope -- global_store_dwordx2 ... sc0 sc1, written through to
// host memory at once (a plain or nontemporal store stays in the L2 until the kernel's end: measured -- the host then sees a
// launch's opcodes all together)
__device__ __forceinline__ void live_put(unsigned long long *slot, unsigned long long v)
{
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int kScoringWaves = 8;            // waves 0..6 score, wave 7 generates MT19937 blocks; further waves only
constexpr int kScorers = kScoringWaves - 1;  // follow the rounds
// a round reads < 623 + kScorers * 258 + 2 words past block 0's start (7 scorers: 2431 -> 4 blocks; 15: 4495 -> 8)
constexpr int kRingNeed = (623 + kScorers * 258 + 2 + 623) / 624;
constexpr int kRing = kRingNeed + 6;       // MT19937 blocks kept in LDS
static_assert(623 + kScorers * 258 + 2 <= 4991, "ring_word divides by 624 exactly only below 4991");

// TW = waves in the workgroup (greedy_team_kernel: 8; waves beyond the eighth would only follow the rounds)
// FOUR (f4, IIV_OPT_FOURTH_OFFSET; not the reference's behaviour): up to three extra offsets per opcode, as in
// greedy_wave_kernel<MODE, 1, true> (iiv_greedy.hip) and oracle/iiv_oracle.c: orc_video_set_fourth_offset.
template <int MODE, int TW, bool FOUR = false>
__device__ __forceinline__ void team_body(StreamState *__restrict__ states, const uint8_t *__restrict__ frames_main,
                                          const uint8_t *__restrict__ frames_aux, int n_frames, const LaunchSeg &g,
                                          const NarrowTables &nt,
                                          uint8_t *__restrict__ ops_out, size_t ops_stride,
                                          unsigned long long *__restrict__ live = nullptr, uint32_t live_tag = 0)
{
    constexpr int kTeamWaves = TW, kTeamThreads = 64 * TW;
    using T = SplitTraits<MODE>;
    constexpr uint32_t INF = 0xffffffffu;
    constexpr int W = kScorers;
    __shared__ uint32_t nz[256];             // update_priority != 0
    __shared__ uint32_t pdone[256];          // byte already emitted as a primary (its diff weight counts as 0)
    __shared__ uint32_t ring[kRing][624];    // random's MT19937: block b of the stream (b = 0: current) is ring[(rbase + b) % kRing]
    __shared__ uint32_t xw[kTeamWaves][64];  // per-wave compaction scratch (every wave keeps the same window)
    // the generator's wd[] (row indices + diff weights, immutable while it lives), staged once per launch: a
    // round then starts with an LDS read instead of a trip to HBM for its rows (one stream per workgroup has
    // the LDS to spare; the one-wave kernel hides that latency with its software pipeline instead)
    __shared__ __attribute__((aligned(16))) uint32_t wd_lds[8192];
    // per entry of the round: [0] = C | bad palette bit << 29 | tie << 30 | dead << 31, [1] = f1 | f2 << 1 (a byte is
    // re-queued).  Two copies used alternately: a wave may be writing round r + 1's result while a slower one
    // still reads round r's after the second barrier.
    __shared__ uint32_t res2[2][kTeamWaves][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Barrier of a round: only LDS traffic (results, bitmaps, MT blocks) has to be visible to the other
    // waves.  __syncthreads() would also drain vmcnt -- the stores of the commit and the table words
    // requested ahead -- and put a full memory round trip into every round.
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    StreamState &S = states[blockIdx.x];
    const int n_ops = IIV_SGPR(g.n_ops), is_aux = IIV_SGPR(g.is_aux), frame = IIV_SGPR(g.frame);
    if (n_ops <= 0) return;
    uint8_t *out = ops_out + (size_t)blockIdx.x * ops_stride + (size_t)IIV_SGPR(g.ops_base) * 6;
    // Live hand-over (iiv_encode_live, include/iivision.h; one-stream encoders): every opcode also goes, as ONE aligned
    // 8-byte store, into a queue in coherent host memory -- its six bytes and the launch's tag in the top two -- so that the
    // caller's generator (video.py: one next() per opcode) hands opcode i out while the kernel works on i + 1 ...: a slot is
    // valid when it carries the tag, no fence and no flag (an aligned 8-byte store arrives whole).  A launch that ends
    // short of its n_ops (one of the reference's asserts, an internal error) puts kLiveEnd behind its last opcode.
    unsigned long long *const lq = live ? live + IIV_SGPR(g.ops_base) : nullptr;
    const unsigned long long lq_tag = (unsigned long long)(live_tag & 0xffffu) << 48;

    if (!S.gen_active || S.error) {
        if (tid == 0 && !S.error) S.error = kErrNoGenerator;
        if (tid == 0 && lq) live_put(lq, lq_tag | kLiveEnd);
        return;
    }
    for (int i = tid; i < 256; i += kTeamThreads) {
        nz[i] = S.nzbits[i];
        pdone[i] = S.pdone[i];
    }
    for (int i = tid; i < 624; i += kTeamThreads) ring[0][i] = S.mt_py[i];
    for (int i = tid; i < 2048; i += kTeamThreads)
        reinterpret_cast<uint4 *>(wd_lds)[i] = reinterpret_cast<const uint4 *>(S.wd)[i];
    __syncthreads();
    int rbase = 0;
    for (int b = 1; b < kRing; b++) mt_twist<kTeamThreads>(ring[b - 1], ring[b], tid);   // (ends with a barrier)
    int valid = kRing;   // blocks 0 .. valid - 1 of the stream (from rbase) hold their words
    auto ring_word = [&](int q) -> uint32_t {   // word q of the stream counted from block 0's start, q < kRingNeed * 624
        const int b = (int)(((uint32_t)q * 6723u) >> 22);   // q / 624 for q < 4991
        int slot = rbase + b;
        slot = slot >= kRing ? slot - kRing : slot;
        return ring[slot][q - 624 * b];
    };
    auto ring_slot = [&](int b) -> int { return (rbase + b) % kRing; };
    // the stream has left block 0 behind: its slot becomes free (no work)
    auto ring_drop = [&]() {
        rbase = rbase + 1 == kRing ? 0 : rbase + 1;
        valid--;
    };
    // append one block (all threads; contains barriers)
    auto ring_grow_all = [&]() {
        mt_twist<kTeamThreads>(ring[ring_slot(valid - 1)], ring[ring_slot(valid)], tid);
        valid++;
    };
    int mt_idx = IIV_SGPR(S.mt_py_idx);
    while (mt_idx >= 624) {
        ring_drop();
        mt_idx -= 624;
    }
    while (valid < kRing) ring_grow_all();

    const int n_sorted = IIV_SGPR(S.n_sorted);
    const int truncated = IIV_SGPR(S.truncated);
    int head = IIV_SGPR(S.head), n_pushed = IIV_SGPR(S.n_pushed), exhausted = IIV_SGPR(S.exhausted);
    int done = 0, err = 0;
    uint32_t draws = 0;
    unsigned long long pad_ops = 0;
    const uint32_t pad_content = (uint32_t)IIV_SGPR(S.pad_content);

    const int o_e = byte_offset<MODE>(0, is_aux), o_d = byte_offset<MODE>(1, is_aux);
    // (narrow form of the split store table: see iiv_stream.h and greedy_wave_kernel)
    const uint32_t l1_e = (uint32_t)o_e << (T::kLeftCBits + T::kLeftRowBits + 1), l1_d = (uint32_t)o_d << (T::kLeftCBits + T::kLeftRowBits + 1);
    const uint32_t r1_e = nt.right_off + ((uint32_t)o_e << (T::kRightCBits + T::kRightRowBits + 1));
    const uint32_t r1_d = nt.right_off + ((uint32_t)o_d << (T::kRightCBits + T::kRightRowBits + 1));
    const uint8_t *tgt_frames = (MODE == kDHGR && is_aux ? frames_aux : frames_main) +
                                ((size_t)blockIdx.x * n_frames + frame) * 8192;
    const uint4 *wd_rows = reinterpret_cast<const uint4 *>(wd_lds);
    uint16_t *up = S.up16[is_aux];   // (the kernels' 16-bit copy of the priorities: a store value is <= 2047)
    uint8_t *mem = S.mem[is_aux];
    const int wsel = lane >> 3;          // this lane's word inside a page's 8 bitmap words
    const int sh0 = (4 * lane) & 31;     // its 4 bits inside that word
    const uint32_t y0 = 4u * (uint32_t)lane;

    // what one wave knows about its entry after scoring
    struct Scored {
        uint32_t nd01, nd23;            // store values of the lane's four bytes, packed in pairs
        int ke[4];                      // eligible keys delta << 20 | offset (>= 0: not eligible)
        unsigned long long cand[4];     // candidate masks (delta < 0, diff weight still counts), per byte column
        int C, y1, y2, y3;
        uint32_t nd1, nd2, nd3;
        bool tie, live;
        bool tie_soft;   // the nonces decide the winners, but not how many entries the step re-queues: it need not end the run
        int fcount;      // how many entries the step re-queues (the winners with a non-zero store value), if that is known
    };
    constexpr int kSlots = FOUR ? 3 : 2;   // extra offsets per opcode
    auto nd_of = [&](const Scored &sc, int y) -> uint32_t {
        const uint32_t pa = (uint32_t)__builtin_amdgcn_readlane((int)sc.nd01, y >> 2);
        const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)sc.nd23, y >> 2);
        return ((((y & 2) ? pb : pa) >> ((y & 1) * 16)) & 0xffffu) - kNarrowBias;   // (the sums are kept with RF's bias)
    };
    // the immutable inputs of a step: the entry's row of wd[] and its eight table words
    struct Loaded {
        uint32_t e;              // the entry they belong to (0 with bit 31 clear: nothing loaded)
        uint32_t kb[4], gl[4], gr[4];   // kb: -diff weight << 20 | offset, less the bias term of the key (see score)
    };
    // the allocation as a raw buffer: a slice base is one 32-bit scalar offset instead of a 64-bit pointer added per lane
    const __amdgpu_buffer_rsrc_t rsrc_nt = __builtin_amdgcn_make_buffer_rsrc((void *)nt.base, 0, 0x7fffffff, 0x00020000);
    // a key is value * kKeyMul + kb = delta << 20 | value << 8 | offset: the winner's key brings its store value along
    constexpr uint32_t kKeyMul = (1u << kWdDwShift) | (1u << 8);
    auto load = [&](uint32_t e, Loaded &L) {
        const int p = (e >> 8) & 31;
        const uint32_t c = (e >> 16) & 0xffu;
        const uint4 w = wd_rows[p * 64 + lane];
        const uint32_t sl_e = l1_e + (split_content_left<MODE>(c, 0) << (T::kLeftRowBits + 1));
        const uint32_t sl_d = l1_d + (split_content_left<MODE>(c, 1) << (T::kLeftRowBits + 1));
        const uint32_t sr_e = r1_e + (split_content_right<MODE>(c, 0) << (T::kRightRowBits + 1));
        const uint32_t sr_d = r1_d + (split_content_right<MODE>(c, 1) << (T::kRightRowBits + 1));
        const uint32_t wr[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            L.gl[r] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc_nt, (int)wd_off_left(wr[r]), (int)((r & 1) ? sl_d : sl_e), 0);
            L.gr[r] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc_nt, (int)wd_off_right(wr[r]), (int)((r & 1) ? sr_d : sr_e), 0);
            L.kb[r] = (y0 + (uint32_t)r) - (wd_dw(wr[r]) << kWdDwShift) - kNarrowBias * kKeyMul;
        }
        L.e = e | 0x80000000u;
    };
    // scoring of the loaded entry against the bitmaps as they are now; no side effects
    auto score = [&](const Loaded &L, Scored &sc) {
        const uint32_t e = L.e;
        const int p = (e >> 8) & 31, x = e & 255;
        uint32_t nzw = nz[p * 8 + wsel], pdw = pdone[p * 8 + wsel];
        const uint32_t xword = (uint32_t)__builtin_amdgcn_readlane((int)nzw, (x >> 5) * 8);
        sc.live = (xword >> (x & 31)) & 1u;     // video.py:130
        // x itself leaves both sets before the page is scored (video.py:140-141)
        const uint32_t xbit = wsel == (x >> 5) ? 1u << (x & 31) : 0u;
        nzw &= ~xbit;
        pdw |= xbit;
        uint32_t nd[4];
        sc.C = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            nd[r] = L.gl[r] + L.gr[r];   // L1 + RF = store value + kNarrowBias (iiv_stream.h: narrow form)
            const int d = (int)(__umul24(nd[r], kKeyMul) + L.kb[r]);   // delta << 20 | value << 8 | offset (screen.py:547)
            const int gone = __builtin_amdgcn_sbfe((int)pdw, sh0 + r, 1);
            const int live = __builtin_amdgcn_sbfe((int)nzw, sh0 + r, 1);
            sc.ke[r] = d & live;                                   // video.py:159
            sc.cand[r] = __ballot((d & ~gone) < 0);                // video.py:283
            sc.C += (int)__popcll(sc.cand[r]);                     // one nonce each (video.py:290-293)
        }
        sc.nd01 = nd[0] | (nd[1] << 16);
        sc.nd23 = nd[2] | (nd[3] << 16);
        const int a0 = sc.ke[0] < sc.ke[1] ? sc.ke[0] : sc.ke[1], b0 = sc.ke[0] < sc.ke[1] ? sc.ke[1] : sc.ke[0];
        const int a1 = sc.ke[2] < sc.ke[3] ? sc.ke[2] : sc.ke[3], b1 = sc.ke[2] < sc.ke[3] ? sc.ke[3] : sc.ke[2];
        const int k1 = a0 < a1 ? a0 : a1, hi01 = a0 < a1 ? a1 : a0, mb = b0 < b1 ? b0 : b1;
        const int k2 = hi01 < mb ? hi01 : mb;
        const int k3 = hi01 < mb ? mb : hi01;   // (FOUR: the lane's third smallest)
        int K1 = k1, K2 = k2;
        wave_top2_i32(K1, K2);   // the wave's two smallest eligible keys in one fused-DPP pass (iiv_wave.h)
        sc.y1 = sc.y2 = sc.y3 = -1;
        sc.nd1 = sc.nd2 = sc.nd3 = 0;
        sc.tie = false;
        sc.tie_soft = false;
        sc.fcount = 0;
        // the delta class of key K (eligible bytes with K's delta): its size; whether it holds zero / non-zero store values
        auto class_size = [&](int K) -> int {
            int n = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) n += (int)__popcll(__ballot(((uint32_t)(sc.ke[r] ^ K) >> kWdDwShift) == 0u));
            return n;
        };
        auto class_values = [&](int K, unsigned long long &zero, unsigned long long &nonzero) {
            zero = nonzero = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool member = ((uint32_t)(sc.ke[r] ^ K) >> kWdDwShift) == 0u;
                zero |= __ballot(member && nd[r] == kNarrowBias);
                nonzero |= __ballot(member && nd[r] != kNarrowBias);
            }
        };
        if (K1 < 0) {
            sc.y1 = K1 & 255;
            sc.nd1 = ((uint32_t)K1 >> 8) & 0x7ffu;
            int K3 = 0;
            if (K2 < 0) {
                sc.y2 = K2 & 255;
                sc.nd2 = ((uint32_t)K2 >> 8) & 0x7ffu;
                if constexpr (FOUR) {
                    K3 = wave_min_i32(k1 > K2 ? k1 : (k2 > K2 ? k2 : k3));   // every lane's smallest key above K2
                    if (K3 < 0) {
                        sc.y3 = K3 & 255;
                        sc.nd3 = ((uint32_t)K3 >> 8) & 0x7ffu;
                    }
                }
            }
            // the winners as found (no tie): what they re-queue
            sc.fcount = (sc.nd1 ? 1 : 0) + (sc.nd2 ? 1 : 0) + (sc.nd3 ? 1 : 0);
            // Do the nonces decide?  KL = the last winner found.  They do if two winners share a delta, or -- with every
            // slot filled -- a further eligible byte shares KL's.
            const int n_won = K2 >= 0 ? 1 : (FOUR && K3 < 0) ? 3 : 2;
            if (n_won >= 2) {
                const int KL = n_won == 3 ? K3 : K2;
                const int dL = KL >> kWdDwShift;
                const bool eq12 = (K1 >> kWdDwShift) == (K2 >> kWdDwShift);
                const bool eq23 = n_won == 3 && (K2 >> kWdDwShift) == dL;
                const bool full = n_won == kSlots;
                // winners with a smaller delta than the last one's: they win whatever the nonces say
                const int lt = ((K1 >> kWdDwShift) < dL ? 1 : 0) + ((n_won == 3 && (K2 >> kWdDwShift) < dL) ? 1 : 0);
                const int r = n_won - lt;                          // slots that go to the last winner's class
                // more members than slots: the nonces choose among them.  (With two slots and two winners sharing a delta
                // the class is counted only to tell a soft tie from a softer one: not worth it -- taken as shared.)
                const bool shared = full && ((!FOUR && eq12) || class_size(KL) > r);
                if (eq12 || eq23 || shared) {
                    // The steps behind a tie need only the NUMBER of random words it will draw (one per candidate, known, and
                    // one per re-queued byte: a winner whose store value is non-zero, video.py:170-178).  That number does not
                    // depend on the nonces if the class the nonces choose from goes in whole (as many members as slots), or
                    // if its members' store values are all zero or all non-zero: the run goes on, and the tie is resolved
                    // by its own wave at commit time, at its own offset of the random stream.  (On picture-like input 96 % of
                    // the steps tie: ending the run at each would leave one step per round.)
                    if (!shared) {
                        sc.tie_soft = true;        // (the winners are the ones found, in an order the nonces decide: fcount stands)
                    } else if (unsigned long long zero = 0, nonzero = 0; class_values(KL, zero, nonzero), !(zero && nonzero)) {
                        sc.tie_soft = true;
                        const int f_lt = ((K1 >> kWdDwShift) < dL && sc.nd1 ? 1 : 0) + ((n_won == 3 && (K2 >> kWdDwShift) < dL && sc.nd2) ? 1 : 0);
                        sc.fcount = f_lt + (nonzero ? r : 0);
                    } else {
                        sc.tie = true;
                    }
                }
            }
        }
    };
    // the reference's (delta, nonce, offset) order with every candidate's nonce materialised; the
    // entry's first nonce is word q0 of the ring (video.py:290-301)
    auto resolve_tie = [&](Scored &sc, int q0) {
        constexpr int kNone = 0x7fffffff;   // the keys below are < 2^28: signed minima order them
        int key[4];
        int run = q0;
#pragma unroll
        for (int q = 0; q < 4; q++) run += prefix_popc(sc.cand[q]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t nonce = mt_temper(ring_word(run)) >> 24;
            run += (int)((sc.cand[r] >> lane) & 1ull);
            const uint32_t k = ((uint32_t)((sc.ke[r] >> kWdDwShift) + 2048) << 16) | (nonce << 8) | (y0 + r);
            key[r] = sc.ke[r] < 0 ? (int)k : kNone;
        }
        // the two smallest: in the lane, then the one-pass fused-DPP top-2 (the keys end in the offset: unique)
        const int ta0 = key[0] < key[1] ? key[0] : key[1], tb0 = key[0] < key[1] ? key[1] : key[0];
        const int ta1 = key[2] < key[3] ? key[2] : key[3], tb1 = key[2] < key[3] ? key[3] : key[2];
        const int t1 = ta0 < ta1 ? ta0 : ta1;
        const int thi = ta0 < ta1 ? ta1 : ta0, tmb = tb0 < tb1 ? tb0 : tb1;
        const int t2 = thi < tmb ? thi : tmb;
        int T1 = t1, T2 = t2;
        wave_top2_i32(T1, T2);   // (kNone in many lanes: a value nobody asks about, see iiv_wave.h)
        sc.y1 = T1 != kNone ? (T1 & 255) : -1;
        sc.y2 = T2 != kNone ? (T2 & 255) : -1;
        sc.nd1 = sc.y1 >= 0 ? nd_of(sc, sc.y1) : 0u;
        sc.nd2 = sc.y2 >= 0 ? nd_of(sc, sc.y2) : 0u;
        if constexpr (FOUR) {
            const int t3 = thi < tmb ? tmb : thi;
            const int T3 = wave_min_i32(t1 > T2 ? t1 : (t2 > T2 ? t2 : t3));   // every lane's smallest key above T2
            sc.y3 = T3 != kNone ? (T3 & 255) : -1;
            sc.nd3 = sc.y3 >= 0 ? nd_of(sc, sc.y3) : 0u;
        }
        sc.fcount = (sc.nd1 ? 1 : 0) + (sc.nd2 ? 1 : 0) + (sc.nd3 ? 1 : 0);
    };
    // video.py:140-144, 170-187; screen.py:256-293: the stores of one step.  q_push = ring position of
    // the first re-queue nonce, push_at / op_at = where its pushed entries / its opcode go.
    auto commit = [&](uint32_t e, const Scored &sc, int q_push, int push_at, int op_at) {
        const int p = (e >> 8) & 31, x = e & 255;
        const uint32_t c = (e >> 16) & 0xffu;
        const uint32_t v1 = sc.y1 >= 0 ? sc.nd1 : 0u, v2 = sc.y2 >= 0 ? sc.nd2 : 0u, v3 = (FOUR && sc.y3 >= 0) ? sc.nd3 : 0u;
        const int y1e = sc.y1 >= 0 ? sc.y1 : x, y2e = sc.y2 >= 0 ? sc.y2 : x, y3e = (FOUR && sc.y3 >= 0) ? sc.y3 : x;   // video.py:185-186
        const int f1 = v1 ? 1 : 0, f2 = v2 ? 1 : 0;
        if (lane < 1 + kSlots) {
            const int off = lane == 0 ? x : lane == 1 ? y1e : lane == 2 ? y2e : y3e;
            const uint32_t val = lane == 0 ? 0u : lane == 1 ? v1 : lane == 2 ? v2 : v3;
            up[p * 256 + off] = (uint16_t)val;
            mem[p * 256 + off] = (uint8_t)c;
            if (val == 0) {
                atomicAnd(&nz[p * 8 + (off >> 5)], ~(1u << (off & 31)));
            } else {
                const int k = lane == 2 ? f1 : lane == 3 ? f1 + f2 : 0;
                const uint32_t nonce = mt_temper(ring_word(q_push + k)) >> 24;   // video.py:178
                S.pushed[push_at + k] = ((2047u - val) << 21) | (nonce << 13) | ((uint32_t)p << 8) | (uint32_t)off;
            }
            if (lane == 0) {
                atomicOr(&pdone[p * 8 + (x >> 5)], 1u << (x & 31));
                uint8_t *q = out + (size_t)op_at * 6;
                q[0] = (uint8_t)(p + 32);
                q[1] = (uint8_t)c;
                q[2] = (uint8_t)x;
                q[3] = (uint8_t)y1e;
                q[4] = (uint8_t)y2e;
                q[5] = (uint8_t)(FOUR ? y3e : x);
                if (lq)
                    live_put(lq + op_at, lq_tag | (unsigned long long)(uint32_t)(p + 32) | ((unsigned long long)c << 8) |
                                             ((unsigned long long)(uint32_t)x << 16) | ((unsigned long long)(uint32_t)y1e << 24) |
                                             ((unsigned long long)(uint32_t)y2e << 32) | ((unsigned long long)(uint32_t)(FOUR ? y3e : x) << 40));
            }
        }
    };

    // ---- phase A: the sorted initial list, up to W entries per round.  Every wave keeps the same
    // window of live entries (the compaction is done redundantly, each wave in its own scratch).
    int win_end = head, n_dense = 0, qi = 0, win_base = head;
    uint32_t dense_e = 0;
    bool list_done = head >= n_sorted || exhausted;
    int guard = n_ops + 8192 + 64;
    int parity = 0;
    const bool mt_wave = wave == kScorers;
    Loaded cur;
    cur.e = 0;
#ifdef IIV_STAMPS
    // diagnostic build: shader-clock time of wave 0 per phase of a round, rounds, entries, real time
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
    const unsigned long long t_start = ph_t, rt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long n_rounds = 0, n_entries = 0, n_end_avail = 0, n_end_room = 0, n_end_page = 0, n_dead = 0, n_end_tie = 0;
#define IIV_PHASE(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph[i] += now_ - ph_t;                                         \
        ph_t = now_;                                                  \
    } while (0)
#else
#define IIV_PHASE(i) do { } while (0)
#endif
    while (!list_done && done < n_ops && !err) {
        if (--guard < 0) {
            err = kErrGuard;
            break;
        }
        if (qi >= n_dense) {   // next window of 64 list positions (uniform across the workgroup)
            if (win_end >= n_sorted) {
                list_done = true;
                head = n_sorted;
                break;
            }
            const int start = win_end, idx = start + lane;
            const uint32_t ew = idx < n_sorted ? S.order[idx] : 0u;
            const uint32_t loc = ew & 0x1fffu;
            const bool v = idx < n_sorted && ((nz[loc >> 5] >> (loc & 31)) & 1u);
            const unsigned long long mask = __ballot(v);
            n_dense = (int)__popcll(mask);
            qi = 0;
            if (v) xw[wave][prefix_popc(mask)] = (ew & 0x00ffffffu) | ((uint32_t)lane << 24);
            wave_lds_sync();
            dense_e = xw[wave][lane];
            wave_lds_sync();
            win_base = start;
            win_end = start + 64 < n_sorted ? start + 64 : n_sorted;
            if (n_dense == 0) head = win_end;
            continue;
        }
        // the run: entries qi .. qi + B - 1 of the window, pairwise distinct pages, within the budget.
        // Lane v looks at entry qi + v: an exclusive OR-scan of the page bits over lanes 0..7 (three DPP
        // shifts inside a row of 16 lanes) tells each lane whether an earlier entry of the run is on its page.
        int B;
        uint32_t my_e;
        {
            const int avail = n_dense - qi < W ? n_dense - qi : W;
            const int room = n_ops - done < avail ? n_ops - done : avail;
            const uint32_t ev = (uint32_t)__builtin_amdgcn_ds_bpermute(((qi + lane) & 63) << 2, (int)dense_e);
            const uint32_t pbit = lane < room ? 1u << ((ev >> 8) & 31) : 0u;
            uint32_t acc = pbit;   // inclusive OR-scan over the first eight lanes
            acc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, false);   // row_shr:1
            acc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xf, 0xf, false);   // row_shr:2
            acc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xf, 0xf, false);   // row_shr:4
            if (W > 8) acc |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xf, 0xf, false);   // row_shr:8
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, false);  // exclusive
            const unsigned long long stop = __ballot(lane >= room || (before & pbit) != 0u);   // first lane that ends the run
            B = (int)__builtin_ctzll(stop | (1ull << W));
            my_e = (uint32_t)__builtin_amdgcn_readlane((int)ev, wave < W ? wave : 0) & 0x00ffffffu;
#ifdef IIV_STAMPS
            if (B < W) {   // why the run is shorter than W: the window, the launch's budget, or a page already in the run
                if (B >= avail) n_end_avail++;
                else if (B >= room) n_end_room++;
                else n_end_page++;
            }
#endif
        }
        IIV_PHASE(0);   // loop top: window, run formation
        // the MT19937 wave extends the stream meanwhile: one block while the others score, a second one
        // while they commit (the slots behind block `valid - 1` are free: this round reads only blocks
        // 0 .. kRingNeed - 1 <= valid - 1).  One block takes that wave about as long as a scoring.
        const int grow1 = kRing - valid >= 1 ? 1 : 0, grow2 = kRing - valid >= 2 ? 1 : 0;
#ifdef IIV_STAMPS
        const unsigned long long mt0 = __builtin_amdgcn_s_memtime();
#endif
        if (mt_wave && grow1) mt_twist_wave(ring[ring_slot(valid - 1)], ring[ring_slot(valid)], lane);
        uint32_t(*res)[2] = res2[parity];
        parity ^= 1;
        Scored sc;
        sc.C = 0;
        sc.live = false;
        sc.tie = false;
        sc.tie_soft = false;
        sc.y1 = sc.y2 = sc.y3 = -1;
        sc.nd1 = sc.nd2 = sc.nd3 = 0;
        sc.fcount = 0;
        if (wave < B) {
            if (cur.e != (my_e | 0x80000000u)) load(my_e, cur);   // (not the entry that was requested ahead)
            score(cur, sc);
            if (MODE == kDHGR && sc.live && ((my_e >> 16) & 0xffu) >= 0x80) sc.C |= 1 << 29;   // video.py:137
            if (lane == 0) {
                res[wave][0] = (uint32_t)sc.C | (sc.live ? 0u : 1u << 31) | (sc.tie ? 1u << 30 : 0u);
                res[wave][1] = (uint32_t)sc.fcount;   // entries the step re-queues (a tie the nonces decide: rewritten after its resolution)
            }
        }
        IIV_PHASE(1);   // load (if not requested ahead) + score
        lds_barrier();   // B: every result of the round is in LDS
        IIV_PHASE(2);   // waiting at barrier B
        // prefix sums (every wave computes all of them: they are the next round's state too).  Lane v
        // holds entry v's result; the draws / pushes / opcodes in front of every entry are one packed
        // DPP scan over the first row of lanes (draws < 2^12 in bits 0..11, pushes in 12..16, opcodes in 17..21).
        const uint2 rv = lane < B ? *reinterpret_cast<const uint2 *>(res[lane]) : make_uint2(0x80000000u, 0);
        const bool r_dead = rv.x >> 31, r_tie = (rv.x >> 30) & 1u, r_bad = (rv.x >> 29) & 1u;
        const unsigned long long stopper = __ballot(lane < B && !r_dead && (r_tie || r_bad));
        const int first_stop = stopper ? (int)__builtin_ctzll(stopper) : -1;
        const int bad_palette = first_stop >= 0 && ((uint32_t)__builtin_amdgcn_readlane((int)rv.x, first_stop) >> 29 & 1u);
        const int tie_at = first_stop >= 0 && !bad_palette ? first_stop : -1;
        // entries that take effect this round: up to and including a tie (it is resolved with the nonces
        // at its own offset; what was scored behind it is redone), up to but excluding a bad content byte
        int n_commit = first_stop < 0 ? B : bad_palette ? first_stop : first_stop + 1;
        const uint32_t f12 = rv.y;
        const bool counts = lane < n_commit && !r_dead && lane != tie_at;   // (the tie's own numbers are added after its resolution)
        const uint32_t item = counts ? ((rv.x & 0x1fffu) + f12) | (f12 << 12) | (1u << 17) : 0u;
        uint32_t scan = item;
        scan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)scan, 0x111, 0xf, 0xf, false);   // row_shr:1
        scan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)scan, 0x112, 0xf, 0xf, false);   // row_shr:2
        scan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)scan, 0x114, 0xf, 0xf, false);   // row_shr:4
        if (W > 8) scan += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)scan, 0x118, 0xf, 0xf, false);   // row_shr:8
        const uint32_t mine = (uint32_t)__builtin_amdgcn_readlane((int)(scan - item), wave < W ? wave : 0);
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)scan, W > 8 ? 15 : 7);
        const int q_mine = mt_idx + (int)(mine & 0xfffu), push_mine = n_pushed + (int)((mine >> 12) & 0x1fu),
                  op_mine = done + (int)(mine >> 17);
        int q = mt_idx + (int)(all & 0xfffu), pu = n_pushed + (int)((all >> 12) & 0x1fu), op = done + (int)(all >> 17);
        if (bad_palette) err = kErrPaletteBit;
        if (pu + kSlots > kPushedCap) {   // (+ the tie entry, whose pushes are not known yet)
            err = kErrPushedOverflow;
            n_commit = 0;
        }
        IIV_PHASE(3);   // prefix sums
        // request the table words of the entry this wave scores next round if the window has it
        Loaded nxt;
        nxt.e = 0;
        if (!mt_wave && !err && qi + n_commit + wave < n_dense) {
            const uint32_t en = (uint32_t)__builtin_amdgcn_readlane((int)dense_e, qi + n_commit + wave) & 0x00ffffffu;
            load(en, nxt);
        }
        if (mt_wave && grow2) mt_twist_wave(ring[ring_slot(valid)], ring[ring_slot(valid + 1)], lane);
#ifdef IIV_STAMPS
        if (mt_wave && lane == 0) {
            S.stamps[29] += __builtin_amdgcn_s_memtime() - mt0;
            S.stamps[30] += (unsigned long long)(grow1 + grow2);
        }
#endif
        if (wave < n_commit && sc.live) {
            if (sc.tie_soft) resolve_tie(sc, q_mine);   // (its re-queue count was published: nothing to tell the others)
            if (wave == tie_at) {
                resolve_tie(sc, q_mine);
                if (lane == 0) res[wave][1] = (uint32_t)sc.fcount;
            }
            commit(my_e, sc, q_mine + sc.C, push_mine, op_mine);
        }
        IIV_PHASE(4);   // request ahead + commit
        lds_barrier();   // C: bitmaps, the tie's outcome and the new MT blocks are visible (pushed[] is read only in phase B)
        IIV_PHASE(5);   // waiting at barrier C
        valid += grow1 + grow2;
        if (tie_at >= 0 && !err) {
            const uint32_t r0 = res[tie_at][0], r1 = res[tie_at][1];
            q += (int)(r0 & 0x1fffu) + (int)r1;
            pu += (int)r1;
            op += 1;
        }
        draws += (uint32_t)(q - mt_idx);
        mt_idx = q;
        n_pushed = pu;
        done = op;
        if (n_commit > 0) {
            const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)dense_e, qi + n_commit - 1);
            head = win_base + (int)(last >> 24) + 1;
            qi += n_commit;
        }
        while (mt_idx >= 624) {
            ring_drop();
            mt_idx -= 624;
        }
        while (valid < kRingNeed) ring_grow_all();   // (uniform: contains barriers; only if the MT wave fell behind)
        cur = nxt;
        IIV_PHASE(6);   // bookkeeping + taking over the requested words
#ifdef IIV_STAMPS
        n_rounds++;
        n_entries += (unsigned long long)n_commit;
        n_dead += (unsigned long long)__popcll(__ballot(lane < n_commit && r_dead));
        if (tie_at >= 0 && tie_at + 1 < B) n_end_tie++;
#endif
    }
#ifdef IIV_STAMPS
    if (tid == 0) {
        for (int i = 0; i < 8; i++) S.stamps[16 + i] += ph[i];
        S.stamps[24] += n_rounds;
        S.stamps[25] += n_entries;
        S.stamps[26] += __builtin_amdgcn_s_memtime() - t_start;
        S.stamps[27] += __builtin_amdgcn_s_memrealtime() - rt_start;
        S.stamps[28] += 1;
        S.stamps[13] += n_end_avail;
        S.stamps[14] += n_end_room;
        S.stamps[15] += n_end_page;
        S.stamps[31] += n_dead | (n_end_tie << 32);
    }
#endif
#undef IIV_PHASE

    // ---- phase B: the re-queued bag, wave 0 alone (the other waves are done)
    __syncthreads();
    if (wave != 0) return;
    while (done < n_ops && !err && !exhausted) {
        if (--guard < -2 * kPushedCap) {
            err = kErrGuard;
            break;
        }
        if (truncated) {  // more initial entries exist than were ordered: host budget bug
            err = kErrSortBudget;
            break;
        }
        if (!list_done) break;   // (the budget ran out inside the list)
        unsigned long long best = ~0ull;
        for (int i = lane; i < n_pushed; i += 64) {
            unsigned long long k = ((unsigned long long)S.pushed[i] << 32) | (unsigned)i;
            best = k < best ? k : best;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long other = __shfl_xor(best, d, 64);
            best = other < best ? other : best;
        }
        const uint32_t bk = (uint32_t)IIV_SGPR((uint32_t)(best >> 32));
        const uint32_t bi = (uint32_t)IIV_SGPR((uint32_t)best);
        if (bk == INF) {
            exhausted = 1;  // video.py:189
            break;
        }
        if (lane == 0) S.pushed[bi] = INF;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // that store precedes the next scan of pushed[]
        const uint32_t loc = bk & 0x1fffu;
        const uint32_t c = (uint32_t)IIV_SGPR(tgt_frames[loc]);
        const uint32_t e = loc | (c << 16);
        Scored sc;
        Loaded L;
        load(e, L);
        score(L, sc);
        if (!sc.live) continue;
        if (MODE == kDHGR && c >= 0x80) {
            err = kErrPaletteBit;
            break;
        }
        if (n_pushed + kSlots > kPushedCap) {
            err = kErrPushedOverflow;
            break;
        }
        if (sc.tie || sc.tie_soft) resolve_tie(sc, mt_idx);
        const int fq = sc.fcount;
        commit(e, sc, mt_idx + sc.C, n_pushed, done);
        wave_lds_sync();
        mt_idx += sc.C + fq;
        draws += (uint32_t)(sc.C + fq);
        n_pushed += fq;
        done++;
        while (mt_idx >= 624) {
            ring_drop();
            mt_idx -= 624;
        }
        while (valid < 2) {   // one wave: the twist without the workgroup barriers (a step reads < 624 + 258 words)
            mt_twist_wave(ring[ring_slot(valid - 1)], ring[ring_slot(valid)], lane);
            valid++;
        }
    }
    if (exhausted && done < n_ops && !err) {
        for (int i = done + lane; i < n_ops; i += 64) {  // video.py:249-251
            uint8_t *q = out + (size_t)i * 6;
            q[0] = 32; q[1] = (uint8_t)pad_content; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0;
            if (lq) live_put(lq + i, lq_tag | 32ull | ((unsigned long long)(pad_content & 0xffu) << 8));
        }
        pad_ops += (unsigned long long)(n_ops - done);
        done = n_ops;
    }
    wave_lds_sync();
    for (int i = lane; i < 256; i += 64) {
        S.nzbits[i] = nz[i];
        S.pdone[i] = pdone[i];
    }
    for (int i = lane; i < 624; i += 64) S.mt_py[i] = ring[rbase][i];
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
        if (lq && done < n_ops) live_put(lq + done, lq_tag | kLiveEnd);
    }
}

}  // namespace iiv
