// iiv_encode.hip -- video.Video.encode_frame on gfx950
// (reference: transcoder/video.py:72-301, transcoder/screen.py:383-547).
//
// One workgroup per independent video stream; all stream state lives in HBM
// (StreamState) between launches and in LDS inside one.
//
//   prologue_kernel (1024 threads / stream)           video.py:104-119, 254-271
//     diff_weights of current screen vs target from the precomputed
//     edit-distance table, hole masking, update_priority accumulation, nonce
//     draw from the numpy MT19937 stream, 64-bit key sort -> `order[]`.
//   greedy_kernel   (256 threads = one lane per page byte / stream)
//                                                     video.py:121-187, 275-301
//     pops `order[]` (then the pushed bag), scores every byte of the page
//     against the popped content via the store table, ranks candidates with a
//     ballot prefix + wave top-2 reduction, applies <=3 stores, emits one
//     opcode per step.
//
// How the reference's heap is restated (proved equivalent on the CPU by
// oracle/iiv_oracle.c:step_struct against the imported reference):
//   * initial heap entries (-priority, nonce, page, offset) all have negative
//     keys, re-queued entries have key 65536-p > 0 (np.uint16 negation wraps,
//     video.py:178) => initial entries pop first, in sorted order;
//   * a byte whose update_priority is 0 is never selected again inside one
//     generator (video.py:130,159 skip it), so lazy deletion is permanent and
//     validity only ever decays;
//   * _compute_error's heap (video.py:290-301) only matters up to the first two
//     entries whose priority is non-zero => top-2 of (delta, nonce, offset).
// RNG: both global MT19937 streams are advanced on the device exactly as
// random.getrandbits(8) (high byte) and np.random.randint(0,256,n) (low byte)
// advance them (video.py:178,265,291).
#include "iiv_host.h"
#include "iiv_edit.h"

#include <stdlib.h>
#include <vector>

namespace iiv {

constexpr int kPushedCap = 16384;  // >= 2 pushes x 7680 non-hole bytes

struct StreamState {
    uint8_t mem[2][8192];     // [is_aux] Video.memory_map / aux_memory_map
    int32_t up[2][8192];      // [is_aux] Video.update_priority / aux_update_priority
    uint32_t wd[8192];        // live generator, per byte of its bank: target window | diff_weight << 16
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
    unsigned long long stamps[32];  // diagnostic builds only (-DIIV_STAMPS): prologue s_memtime stamps [0,16), greedy phase cycles [16,24)
};

enum { kErrNone = 0, kErrHoles = 1, kErrNegative = 2, kErrPaletteBit = 3, kErrPushedOverflow = 4, kErrNoGenerator = 5, kErrGuard = 6, kErrSortBudget = 7 };

// LDS accesses of one wave execute in order: making one lane's LDS writes visible to the
// other lanes of the same wave needs no s_barrier, only that the compiler keeps the order.
__device__ static inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// next MT19937 block, one wave, fully unrolled (constant LDS offsets, no loop counters)
__device__ static inline void mt_twist_wave(const uint32_t *src, uint32_t *dst, int lane)
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

// ------------------------------------------------------------------------- prologue

constexpr int kProThreads = 1024;

#ifdef IIV_STAMPS
#define IIV_STAMP(i)                                                        \
    do {                                                                    \
        __syncthreads();                                                    \
        if (threadIdx.x == 0) S.stamps[i] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define IIV_STAMP(i) do { } while (0)
#endif

// Bitonic sort of 8*NT u64 keys, 8 consecutive elements per thread.  Exchange
// distances 1,2,4 stay inside a thread's registers, 8..256 are wave shuffles, and
// only distances >= 512 (10 of the 91 stages at 8192 keys) go through LDS.
__device__ static inline void cmpx(unsigned long long &lo, unsigned long long &hi, bool asc)
{
    bool sw = (lo > hi) == asc;
    unsigned long long t = lo;
    lo = sw ? hi : lo;
    hi = sw ? t : hi;
}

// Bitonic sort of KPT * NT u64 keys held KPT consecutive keys per thread by all NT
// threads of the workgroup.  Exchange distances below KPT stay in registers, up to
// 32 * KPT they are wave shuffles, and only larger ones go through LDS (xbuf).
template <int KPT, int NT>
__device__ static inline void bitonic_sort(unsigned long long (&v)[KPT], unsigned long long *xbuf, int tid)
{
    constexpr int N = KPT * NT;
#pragma unroll
    for (int k = 2; k <= KPT; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((r & j) == 0) cmpx(v[r], v[r | j], (((KPT * tid + r) & k) == 0));
        }
    }
    for (int k = 2 * KPT; k <= N; k <<= 1) {
        const bool asc = ((KPT * tid) & k) == 0;
        for (int j = k >> 1; j >= KPT; j >>= 1) {
            const int d = j / KPT;
            const bool take_min = ((tid & d) == 0) == asc;
            if (d < 64) {
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    unsigned long long p = __shfl_xor(v[r], d, 64);
                    v[r] = take_min ? (p < v[r] ? p : v[r]) : (p > v[r] ? p : v[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPT; r++) xbuf[r * NT + tid] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    unsigned long long p = xbuf[r * NT + (tid ^ d)];
                    v[r] = take_min ? (p < v[r] ? p : v[r]) : (p > v[r] ? p : v[r]);
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = KPT >> 1; j >= 1; j >>= 1) {
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((r & j) == 0) cmpx(v[r], v[r | j], asc);
        }
    }
}

// sorted keys -> order[] entries (page << 8 | offset | content << 16)
template <int KPT> __device__ static inline void write_order(uint32_t *order, const unsigned long long (&v)[KPT], int tid)
{
    uint32_t o[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) o[j] = ((uint32_t)(v[j] >> 8) & 0x1fffu) | (((uint32_t)v[j] & 0xffu) << 16);
    if (KPT == 2) {
        *reinterpret_cast<uint2 *>(order + 2 * tid) = make_uint2(o[0], o[1]);
    } else {
        uint4 *q = reinterpret_cast<uint4 *>(order + KPT * tid);
#pragma unroll
        for (int j = 0; j < KPT / 4; j++) q[j] = make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
    }
}

// exclusive prefix sum of one int per thread over the workgroup (row-major thread order)
template <int NT> __device__ static inline int block_scan_excl(int v, int tid, uint32_t *wsum, int &total)
{
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += o;
    }
    __syncthreads();  // wsum may still be read from a previous use
    if ((tid & 63) == 63) wsum[tid >> 6] = (uint32_t)incl;
    __syncthreads();
    int wbase = 0;
    total = 0;
    for (int w = 0; w < NT / 64; w++) {
        int x = (int)wsum[w];
        if (w < (tid >> 6)) wbase += x;
        total += x;
    }
    return wbase + incl - v;
}

constexpr int kSelNeedMax = 2048;  // partial sort is used when 3 * opcode budget <= this
constexpr int kBucketMax = 96;      // buckets larger than this fall back to the bitonic sort

// DP == false: diff weights are gathered from the precomputed table (one random
// HBM line per screen byte).  DP == true: they are recomputed by running the
// edit-distance recurrence on the two colour strings (an L2-resident 16 B LUT
// entry each) -- bit-identical by construction (same recurrence that built the
// table), and far cheaper than an HBM line fetch per byte.
template <int MODE, bool DP>
__global__ __launch_bounds__(kProThreads, 8) void prologue_kernel(StreamState *__restrict__ states,
                                                               const uint8_t *__restrict__ frames_main,
                                                               const uint8_t *__restrict__ frames_aux, int n_frames,
                                                               int frame, int is_aux,
                                                               const uint16_t *__restrict__ table,
                                                               const ulonglong2 *__restrict__ strings,
                                                               const uint16_t *__restrict__ sub, int need)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    constexpr int NB = ModeTraits<MODE>::kBanks;
    // One LDS block, carved by hand so that the small lookup tables sit at the lowest
    // addresses: their offsets then fold into the 16-bit offset field of ds_read and the
    // recurrence needs no address add per lookup.
    //      0  lut    16x16 substitute costs (u16)
    //    512  aux4k  DHGR colour-string LUTs | handed-over diff weights, then histogram, then bucket cursors
    //   4608  mtb    [0] = the stream's current MT19937 block; later the bucket starts
    //   9600  wsum, flags
    //   9728  smem   64 KiB: staged memory maps (cur | tgt) + generated MT blocks, then -- once
    //                every diff weight is in registers -- the sort's key buffer
    __shared__ __attribute__((aligned(16))) unsigned char lds[9728 + 65536];
    uint16_t *lut = reinterpret_cast<uint16_t *>(lds);
    uint32_t *aux4k = reinterpret_cast<uint32_t *>(lds + 512);
    uint32_t(*mtb)[624] = reinterpret_cast<uint32_t(*)[624]>(lds + 4608);
    uint32_t *wsum = reinterpret_cast<uint32_t *>(lds + 9600);
    int &flag_bad = *reinterpret_cast<int *>(lds + 9664);
    int &sel_bucket = *reinterpret_cast<int *>(lds + 9668);
    int &sel_count = *reinterpret_cast<int *>(lds + 9672);
    int &big_bucket = *reinterpret_cast<int *>(lds + 9676);
    unsigned char *smem = lds + 9728;
    uint8_t(*cur)[8192] = reinterpret_cast<uint8_t(*)[8192]>(smem);
    uint8_t(*tgt)[8192] = reinterpret_cast<uint8_t(*)[8192]>(smem + NB * 8192);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
    // np.random's MT19937 blocks 1..13 -- every block the <= 7680 draws of this call can reach --
    // are generated above cur | tgt by the last wave while the other waves score
    uint32_t *gen = reinterpret_cast<uint32_t *>(smem + 2 * NB * 8192);

    const int tid = threadIdx.x;
    StreamState &S = states[blockIdx.x];
    const size_t fbase = ((size_t)blockIdx.x * n_frames + frame) * 8192;

    IIV_STAMP(0);
    const int first = S.mt_np_idx;  // (read before the first barrier: thread 0 updates it later)
    if (tid == 0) flag_bad = 0;
    // stage current screen and target memory maps (16 B per lane per load)
    for (int i = tid; i < 512 * NB; i += kProThreads) {
        int b = i >> 9, k = i & 511;
        reinterpret_cast<uint4 *>(cur[b])[k] = reinterpret_cast<const uint4 *>(S.mem[b])[k];
        const uint8_t *src = (b == 0 ? frames_main : frames_aux) + fbase;
        reinterpret_cast<uint4 *>(tgt[b])[k] = reinterpret_cast<const uint4 *>(src)[k];
    }
    IIV_STAMP(10);
    for (int i = tid; i < 624; i += kProThreads) mtb[0][i] = S.mt_np[i];
    if (DP) load_cost_lut(lut, sub, tid);
    IIV_STAMP(11);
    // DHGR colour strings from three LDS lookups instead of ten rotates: pixels 0..3
    // depend on dots 0..6, pixels 4..6 on dots 4..9, pixels 7..9 on dots 7..12
    // (colours.py:100-134).  slut[odd][0..127 | 128..191 | 192..255], one pixel per byte.
    // The ten pixels of a string occupy byte slots 0..9 of three words: pixels 0..3 in word
    // 0, 4..6 in bytes 0..2 of word 1, 7 in byte 3 of word 1, 8..9 in word 2 -- so the
    // third LUT delivers a pair (lo: pixel 7 in byte 3; hi: pixels 8, 9).
    uint32_t *slut = aux4k;  // 2 parities x (128 + 64) words + 2 x 64 x 2 words = 2.5 KiB
    if (DP && MODE == kDHGR && tid < 512) {
        const int odd = tid >> 8, e = tid & 255;
        const int ph = phase_of(kDHGR, byte_offset<kDHGR>(odd, is_aux));
        const int k0 = e < 128 ? 0 : e < 192 ? 4 : 7, nk = e < 128 ? 4 : 3;
        const uint32_t v = e < 128 ? e : (e - 128) & 63;
        uint32_t px[4] = {0, 0, 0, 0};
        for (int k = 0; k < nk; k++) {
            const uint32_t win = (v >> k) & 0xf;
            px[k] = ((win | (win << 4)) >> (4 - ((ph + k0 + k) & 3))) & 0xf;
        }
        if (e < 192) {
            slut[odd * 320 + e] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        } else {
            slut[odd * 320 + 192 + 2 * (e - 192)] = px[0] << 24;
            slut[odd * 320 + 192 + 2 * (e - 192) + 1] = px[1] | (px[2] << 8);
        }
    }
    __syncthreads();
    IIV_STAMP(1);
    const int tgt_first = tgt[(MODE == kDHGR && is_aux) ? 1 : 0][0];
    const int own_b = (MODE == kDHGR && is_aux) ? 1 : 0;
    // 8 consecutive bytes of one page row per thread (row-major, as nonzero() walks them)
    const int i0 = tid * 8;
    // The last wave generates the MT19937 blocks instead of scoring its 512 bytes (pages 30, 31);
    // the first 512 threads score one of those bytes each on top of their own eight and hand the
    // diff weight over through LDS.
    const bool mt_wave = tid >= kProThreads - 64;
    uint16_t *dwx = reinterpret_cast<uint16_t *>(aux4k + 768);  // (the string LUTs end at word 640)

    int32_t upv[8];
    {
        const int4 *p = reinterpret_cast<const int4 *>(S.up[is_aux] + i0);
        int4 a = p[0], b = p[1];
        upv[0] = a.x; upv[1] = a.y; upv[2] = a.z; upv[3] = a.w;
        upv[4] = b.x; upv[5] = b.y; upv[6] = b.z; upv[7] = b.w;
    }
    if (mt_wave) {
        __builtin_amdgcn_s_setprio(3);  // the other waves only have to wait for this one
        const uint32_t *src = mtb[0];
        for (int k = 0; k < 13; k++) {
            mt_twist_wave(src, gen + k * 624, tid & 63);
            src = gen + k * 624;
        }
        __builtin_amdgcn_s_setprio(0);
    }
    IIV_STAMP(12);

    // windows of page byte (page, y) on the current screen and in the target (0, 0 for a hole)
    auto windows = [&](int page, int y, uint32_t &cm, uint32_t &tm) {
        const uint8_t *cur_own = cur[own_b] + page * 256, *cur_oth = cur[NB - 1 - own_b] + page * 256;
        const uint8_t *tgt_own = tgt[own_b] + page * 256, *tgt_oth = tgt[NB - 1 - own_b] + page * 256;
        uint32_t cp, cn, tp, tn;
        neighbours<MODE>(cur_own, cur_oth, y, is_aux, cp, cn);
        neighbours<MODE>(tgt_own, tgt_oth, y, is_aux, tp, tn);
        cm = masked_window<MODE>(cp, cur_own[y], cn, y & 1);
        tm = masked_window<MODE>(tp, tgt_own[y], tn, y & 1);
    };
    // diff weight of one byte = edit distance current window -> target window (screen.py:400-449)
    auto diff_weight = [&](uint32_t cm, uint32_t tm, int y) -> uint32_t {
        const int odd = y & 1;
        const int o = byte_offset<MODE>(y, is_aux);
        if (!DP) return table[((size_t)o << (2 * BITS)) + ((size_t)cm << BITS) + tm];  // screen.py:441-443
        if (cm == tm) return 0u;
        if (MODE == kDHGR) {
            // DHGR windows are already dot strings (screen.py:983-990)
            const uint32_t *sl = slut + 320 * odd;
            const uint2 ca = reinterpret_cast<const uint2 *>(sl + 192)[cm >> 7];
            const uint2 ct = reinterpret_cast<const uint2 *>(sl + 192)[tm >> 7];
            const uint32_t src[3] = {sl[cm & 127], sl[128 + ((cm >> 4) & 63)] | ca.x, ca.y};
            const uint32_t tgt[3] = {sl[tm & 127], sl[128 + ((tm >> 4) & 63)] | ct.x, ct.y};
            return edit_distance_bytes<ModeTraits<MODE>::kDots, 3>(src, tgt, lut);
        }
        const ulonglong2 *Sg = strings + ((size_t)o << BITS);
        const ulonglong2 a = Sg[cm], b = Sg[tm];
        return edit_distance<ModeTraits<MODE>::kDots>(a.x, (uint32_t)a.y, b.x, (uint32_t)b.y, lut);
    };

    uint32_t dwv[8], tmv[8];
    uint32_t cpk[2] = {0, 0};  // the 8 target bytes, packed (needed again when the keys are built)
    int bad = 0;
    {
        const int page = i0 >> 8;
        const uint8_t *cur_own = cur[own_b] + page * 256, *tgt_own = tgt[own_b] + page * 256;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int y = (i0 & 255) + j;
            tmv[j] = 0;
            dwv[j] = 0;
            cpk[j >> 2] |= (uint32_t)tgt_own[y] << (8 * (j & 3));
            if (is_hole(y)) {  // video.py:111
                if (cur_own[y] != 0) bad = kErrHoles;  // video.py:87
                continue;
            }
            uint32_t cm, tm;
            windows(page, y, cm, tm);
            tmv[j] = tm;
            if (!mt_wave) dwv[j] = diff_weight(cm, tm, y);
        }
    }
    if (tid < 512) {
        const int page = 30 + (tid >> 8), y = tid & 255;
        uint32_t d = 0;
        if (!is_hole(y)) {
            uint32_t cm, tm;
            windows(page, y, cm, tm);
            d = diff_weight(cm, tm, y);
        }
        dwx[tid] = (uint16_t)d;
    }
    __syncthreads();
    if (mt_wave) {
#pragma unroll
        for (int j = 0; j < 8; j++) dwv[j] = dwx[(tid - (kProThreads - 64)) * 8 + j];
    }
    uint32_t nzmask = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        int32_t u = upv[j];
        if (dwv[j] == 0) u = 0;  // video.py:115
        u += (int32_t)dwv[j];    // video.py:116
        if (u < 0) bad = kErrNegative;  // video.py:117
        upv[j] = u;
        if (u != 0) nzmask |= 1u << j;
    }
    {
        int4 *p = reinterpret_cast<int4 *>(S.up[is_aux] + i0);
        p[0] = make_int4(upv[0], upv[1], upv[2], upv[3]);
        p[1] = make_int4(upv[4], upv[5], upv[6], upv[7]);
        uint4 *q = reinterpret_cast<uint4 *>(S.wd + i0);
        q[0] = make_uint4(tmv[0] | (dwv[0] << 16), tmv[1] | (dwv[1] << 16), tmv[2] | (dwv[2] << 16),
                          tmv[3] | (dwv[3] << 16));
        q[1] = make_uint4(tmv[4] | (dwv[4] << 16), tmv[5] | (dwv[5] << 16), tmv[6] | (dwv[6] << 16),
                          tmv[7] | (dwv[7] << 16));
        reinterpret_cast<uint8_t *>(S.nzbits)[tid] = (uint8_t)nzmask;  // bit j of byte tid = byte 8*tid+j
        if (tid < 256) S.pdone[tid] = 0;
    }
    if (bad) flag_bad = bad;

    IIV_STAMP(2);
    // row-major rank of each non-zero entry (its index into the nonce draw, video.py:259-265)
    int n;
    const int rank0 = block_scan_excl<kProThreads>(__popc(nzmask), tid, wsum, n);

    // ---- bucket the priorities: 1024 buckets over [0, max].  This is a counting sort
    // (bucket starts = cumulative counts from the top bucket down) whose tiny buckets
    // (~7 entries) are finished below by ranking each entry inside its own bucket; it
    // replaces ~65 dependent bitonic stages by four barriers.
    // It also yields the optional prefix selection: a generator that will be asked for
    // at most B opcodes consumes at most 3B list entries (one primary and at most two
    // secondaries resolved to zero per opcode), so when the host knows B (`need` = 3B)
    // only the buckets holding the `need` highest priorities are ordered at all (every
    // entry of the boundary bucket is kept: a superset of the true top-`need`).
    uint32_t *hist = aux4k;  // the string LUTs are dead by now
    int sh;
    {
        int mx = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) mx = upv[j] > mx ? upv[j] : mx;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_xor(mx, d, 64);
            mx = o > mx ? o : mx;
        }
        hist[tid] = 0;
        __syncthreads();  // (also orders the previous wsum reads before this write)
        if ((tid & 63) == 0) wsum[tid >> 6] = (uint32_t)mx;
        __syncthreads();
        for (int w = 0; w < kProThreads / 64; w++) mx = (int)wsum[w] > mx ? (int)wsum[w] : mx;
        sh = mx < 1024 ? 0 : (32 - __clz(mx)) - 10;
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (nzmask & (1u << j)) atomicAdd(&hist[upv[j] >> sh], 1u);
    __syncthreads();
    // thread t owns bucket 1023 - t; bstart = number of entries in higher buckets
    const int bcount = (int)hist[1023 - tid];
    int total_unused;
    const int bstart = block_scan_excl<kProThreads>(bcount, tid, wsum, total_unused);
    if (tid == 0) {
        sel_bucket = 0;
        sel_count = n;
        big_bucket = 0;
    }
    __syncthreads();
    const bool want_prefix = need > 0 && need <= kSelNeedMax && n > need;
    if (want_prefix && bstart < need && bstart + bcount >= need) {
        sel_bucket = 1023 - tid;
        sel_count = bstart + bcount;
    }
    __syncthreads();
    const int n_sel = sel_count;
    const int first_bucket = sel_bucket;  // buckets >= this are ordered
    if (bcount > kBucketMax && (1023 - tid) >= first_bucket) big_bucket = 1;
    uint32_t selmask = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if ((nzmask & (1u << j)) && (upv[j] >> sh) >= first_bucket) selmask |= 1u << j;
    __syncthreads();  // hist is dead from here on; big_bucket is final
    const bool by_buckets = big_bucket == 0;

    IIV_STAMP(3);
    // n draws of np.random.randint(0, 256): low byte of the next n MT outputs (video.py:265).
    // Draw r is output first + r of the block sequence (block 0 = mtb[0], block k = gen[k - 1]);
    // the stream is left in the block holding the last draw.
    {
        const int blk = first + n > 0 ? (first + n - 1) / 624 : 0;
        const uint32_t *fin = blk ? gen + (blk - 1) * 624 : mtb[0];
        for (int w = tid; w < 624; w += kProThreads) S.mt_np[w] = fin[w];
        if (tid == 0) {
            S.mt_np_idx = first + n - blk * 624;
            S.draws_np += (unsigned long long)n;
        }
    }

    IIV_STAMP(4);
    // keys (-priority, nonce, page, offset) -> ascending u64 (video.py:259-268); bytes
    // whose priority is zero (or that were not selected) get the all-ones key and sink
    unsigned long long kv[8];
    {
        int bk = (first + rank0) / 624, bw = first + rank0 - bk * 624;  // block / word of this thread's next draw
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kv[j] = ~0ull;
            if (nzmask & (1u << j)) {
                // the content byte rides in the low bits (offsets are unique, so it never
                // takes part in the ordering)
                if (selmask & (1u << j)) {
                    const uint32_t nonce = mt_temper((bk ? gen + (bk - 1) * 624 : mtb[0])[bw]) & 0xffu;
                    kv[j] = ((unsigned long long)(0x7fffffffu - (uint32_t)upv[j]) << 29) |
                            ((unsigned long long)nonce << 21) | ((unsigned long long)(i0 + j) << 8) |
                            (unsigned long long)((cpk[j >> 2] >> (8 * (j & 3))) & 0xffu);
                }
                if (++bw == 624) {
                    bw = 0;
                    bk++;
                }
            }
        }
    }
    IIV_STAMP(5);
    if (by_buckets) {
        // counting sort: scatter every selected key into its bucket's slot range, then
        // rank it among the (few) keys of the same bucket
        uint32_t *cursor = aux4k;                                 // histogram is dead
        uint32_t *start = reinterpret_cast<uint32_t *>(mtb);      // MT state is back in HBM
        __syncthreads();
        cursor[1023 - tid] = (uint32_t)bstart;
        start[1023 - tid] = (uint32_t)bstart;
        __syncthreads();
        IIV_STAMP(8);
        // (the bucket is recovered from the key, so the priorities need not stay in registers)
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (kv[j] != ~0ull) {
                const uint32_t bk = (0x7fffffffu - (uint32_t)(kv[j] >> 29)) >> sh;
                keys[atomicAdd(&cursor[bk], 1u)] = kv[j];
            }
        __syncthreads();
        IIV_STAMP(9);
        // rank: the scattered keys lie densely in keys[0, n_sel) grouped by bucket; thread t takes
        // slot t (not "its own" entries, which in prefix mode occupy ~1 lane in 9) and counts
        // the smaller keys of that key's bucket, four independent LDS reads per trip
        for (int t = tid; t < n_sel; t += kProThreads) {
            const unsigned long long key = keys[t];
            const int bk = (int)((0x7fffffffu - (uint32_t)(key >> 29)) >> sh);
            const int s0 = (int)start[bk];
            const int e0 = bk == 0 ? n : (int)start[bk - 1];
            const int last = e0 - 1;
            int below = 0;
            for (int i = s0; i < e0; i += 4) {
                const unsigned long long k0 = keys[i];
                const unsigned long long k1 = keys[i + 1 < e0 ? i + 1 : last];
                const unsigned long long k2 = keys[i + 2 < e0 ? i + 2 : last];
                const unsigned long long k3 = keys[i + 3 < e0 ? i + 3 : last];
                below += (k0 < key ? 1 : 0) + ((i + 1 < e0 && k1 < key) ? 1 : 0) +
                         ((i + 2 < e0 && k2 < key) ? 1 : 0) + ((i + 3 < e0 && k3 < key) ? 1 : 0);
            }
            S.order[s0 + below] = ((uint32_t)(key >> 8) & 0x1fffu) | (((uint32_t)key & 0xffu) << 16);
        }
    } else if (n_sel <= 4 * kProThreads && n_sel < n) {
        // degenerate buckets, prefix selection still small: compact + bitonic
        __syncthreads();  // every key is built: the MT blocks under keys[] are dead
        int tot;
        int nsel_mine = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) nsel_mine += kv[j] != ~0ull ? 1 : 0;
        int pos = block_scan_excl<kProThreads>(nsel_mine, tid, wsum, tot);
        const bool small = tot <= 2 * kProThreads;
        const int n_pad = small ? 2 * kProThreads : 4 * kProThreads;
        for (int i = tot + tid; i < n_pad; i += kProThreads) keys[i] = ~0ull;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (kv[j] != ~0ull) keys[pos++] = kv[j];
        __syncthreads();
        if (small) {
            unsigned long long v2[2] = {keys[2 * tid], keys[2 * tid + 1]};
            __syncthreads();
            bitonic_sort<2, kProThreads>(v2, keys, tid);
            write_order<2>(S.order, v2, tid);
        } else {
            unsigned long long v4[4] = {keys[4 * tid], keys[4 * tid + 1], keys[4 * tid + 2], keys[4 * tid + 3]};
            __syncthreads();
            bitonic_sort<4, kProThreads>(v4, keys, tid);
            write_order<4>(S.order, v4, tid);
        }
    } else {
        // order everything with the bitonic network (keys not selected sink to the end)
        __syncthreads();  // every key is built: the MT blocks under keys[] are dead
        bitonic_sort<8, kProThreads>(kv, keys, tid);
        write_order<8>(S.order, kv, tid);  // entries >= n_sel are never read
    }
    IIV_STAMP(6);
    IIV_STAMP(7);
    if (tid == 0) {
        S.n_sorted = n_sel;
        S.truncated = n_sel < n ? 1 : 0;
        S.head = 0;
        S.n_pushed = 0;
        S.exhausted = 0;
        S.gen_active = 1;
        S.gen_is_aux = is_aux;
        S.gen_frame = frame;
        S.pad_content = tgt_first;
        if (flag_bad && S.error == 0) S.error = flag_bad;
    }
}

// ------------------------------------------------------------------------- greedy

constexpr int kChunk = 8;  // initial-list entries whose store-table rows are gathered together

template <int MODE>
__global__ __launch_bounds__(256) void greedy_kernel(StreamState *__restrict__ states,
                                                     const uint8_t *__restrict__ frames_main,
                                                     const uint8_t *__restrict__ frames_aux, int n_frames, int frame,
                                                     int is_aux, int n_ops, const uint16_t *__restrict__ store,
                                                     uint8_t *__restrict__ ops_out, size_t ops_stride,
                                                     size_t ops_base)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    constexpr int CB = ModeTraits<MODE>::kContentBits;
    constexpr int NB = ModeTraits<MODE>::kBanks;
    constexpr uint32_t INF = 0xffffffffu;
    __shared__ __attribute__((aligned(16))) uint8_t tgt[NB][8192];  // [0] = bank being encoded, [1] = the other one
    __shared__ __attribute__((aligned(16))) uint16_t dwf[8192];     // diff_weight | (priority != 0) << 15
    __shared__ uint32_t mt[2][624];
    __shared__ uint32_t xw_cnt[4];
    __shared__ uint32_t xw_key[8];
    __shared__ unsigned long long xw_pop[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    StreamState &S = states[blockIdx.x];
    const size_t fbase = ((size_t)blockIdx.x * n_frames + frame) * 8192;
    uint8_t *out = ops_out + (size_t)blockIdx.x * ops_stride + ops_base;

    if (!S.gen_active || S.error) {
        if (tid == 0 && !S.error) S.error = kErrNoGenerator;
        return;
    }

    // ---- stage target bytes, diff weights + validity flags, sorted order, RNG block
    for (int i = tid; i < 512 * NB; i += 256) {
        int b = i >> 9, k = i & 511;
        const uint8_t *src;
        if (MODE == kDHGR)
            src = ((b == 0) == (is_aux != 0) ? frames_aux : frames_main) + fbase;
        else
            src = frames_main + fbase;
        reinterpret_cast<uint4 *>(tgt[b])[k] = reinterpret_cast<const uint4 *>(src)[k];
    }
    for (int i = tid; i < 8192; i += 256) {
        uint32_t bit = (S.nzbits[i >> 5] >> (i & 31)) & 1u;
        uint32_t dn = (S.pdone[i >> 5] >> (i & 31)) & 1u;
        dwf[i] = (uint16_t)((dn ? 0u : (S.wd[i] >> 16)) | (bit << 15));
    }
    for (int i = tid; i < 624; i += 256) mt[0][i] = S.mt_py[i];
    __syncthreads();
    int cb = 0;  // mt[cb] = current block, mt[cb^1] = the block after it
    mt_twist<256>(mt[0], mt[1], tid);
    int mt_idx = S.mt_py_idx;
    if (mt_idx >= 624) {
        mt_twist<256>(mt[1], mt[0], tid);
        cb = 1;
        mt_idx -= 624;
    }

    const int n_sorted = S.n_sorted;
    int head = S.head, n_pushed = S.n_pushed, exhausted = S.exhausted;
    int done = 0, err = 0;
    unsigned long long draws = 0, pad_ops = 0;
    const int y = tid;
    const int odd = y & 1;
    const int o = byte_offset<MODE>(y, is_aux);
    const uint16_t *store_o = store + ((size_t)o << (CB + BITS));

    // every iteration either emits an opcode, skips >= 1 list entry or pops a pushed
    // entry, so this bound is never reached; it turns a logic error into an error
    // code instead of a hung GPU.
    int guard = n_ops + 8192 + 2 * kPushedCap + 64;
    while (done < n_ops && !err) {
        if (--guard < 0) {
            err = kErrGuard;
            break;
        }
        if (exhausted) {
            // video.py:249-251: pad forever with (32, target[0,0], [0,0,0,0])
            uint32_t c0 = tgt[0][0];
            for (int i = done + tid; i < n_ops; i += 256) {
                uint8_t *q = out + (size_t)i * 6;
                q[0] = 32; q[1] = (uint8_t)c0; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0;
            }
            pad_ops += (unsigned long long)(n_ops - done);
            done = n_ops;
            break;
        }

        // ---- form a chunk of entries (uniform across the workgroup)
        uint32_t ent[kChunk];
        int pos[kChunk];
        int cnt = 0, chunk_end = head;
        bool from_pushed = false;
        __syncthreads();  // validity flags written by their owner lanes -> visible to the scan
        if (head < n_sorted) {
            int idx = head + lane;
            uint32_t e = idx < n_sorted ? (S.order[idx] & 0x1fffu) : 0u;
            bool v = idx < n_sorted && (dwf[e] & 0x8000u);
            unsigned long long mask = __ballot(v);
            int window_end = head + 64 < n_sorted ? head + 64 : n_sorted;
            if (mask == 0) {
                head = window_end;
                continue;
            }
#pragma unroll
            for (int m = 0; m < kChunk; m++) {
                ent[m] = 0;
                pos[m] = 0;
                if (mask) {
                    int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    ent[m] = __builtin_amdgcn_readlane(e, l);
                    pos[m] = head + l;
                    cnt = m + 1;
                }
            }
            chunk_end = mask ? pos[kChunk - 1] + 1 : window_end;
        } else {
            if (S.truncated) {  // more initial entries exist than were ordered: host budget bug
                err = kErrSortBudget;
                break;
            }
            // pop-min over the pushed bag
            from_pushed = true;
            unsigned long long best = ~0ull;
            for (int i = tid; i < n_pushed; i += 256) {
                unsigned long long k = ((unsigned long long)S.pushed[i] << 32) | (unsigned)i;
                best = k < best ? k : best;
            }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                unsigned long long other = __shfl_xor(best, d, 64);
                best = other < best ? other : best;
            }
            if (lane == 0) xw_pop[wave] = best;
            __syncthreads();
            best = xw_pop[0];
            for (int w = 1; w < 4; w++) best = xw_pop[w] < best ? xw_pop[w] : best;
            uint32_t bk = (uint32_t)(best >> 32);
            if (bk == INF) {
                exhausted = 1;  // video.py:189
                continue;
            }
            if (tid == 0) S.pushed[(uint32_t)best] = INF;
#pragma unroll
            for (int m = 0; m < kChunk; m++) {
                ent[m] = 0;
                pos[m] = 0;
            }
            ent[0] = bk & 0x1fff;
            cnt = 1;
            if (!(dwf[ent[0]] & 0x8000u)) continue;  // video.py:130
        }

        // ---- gather the store-table row of every chunk entry (all in flight together)
        uint32_t ndv[kChunk];
#pragma unroll
        for (int m = 0; m < kChunk; m++) {
            {   // branch-free: see greedy_wave_kernel
                int p = ent[m] >> 8;
                uint32_t c = tgt[0][ent[m]];
                const uint8_t *own_row = tgt[0] + p * 256;
                const uint8_t *oth_row = tgt[NB - 1] + p * 256;
                uint32_t pv, nx;
                neighbours<MODE>(own_row, oth_row, y, is_aux, pv, nx);
                uint32_t win = masked_window<MODE>(pv, own_row[y], nx, odd);
                ndv[m] = store_o[((size_t)(c & ((1u << CB) - 1)) << BITS) + win];
            }
        }

        __builtin_amdgcn_sched_barrier(0);  // retire the gathers here: see greedy_wave_kernel
#pragma unroll
        for (int m = 0; m < kChunk; m++) asm volatile("" : "+v"(ndv[m]));
        __builtin_amdgcn_sched_barrier(0);

        // ---- process the chunk sequentially
        uint32_t dead = 0;
#pragma unroll
        for (int m = 0; m < kChunk; m++) {
            if (m >= cnt || done >= n_ops || err) break;
            if (dead & (1u << m)) {
                head = pos[m] + 1;
                continue;
            }
            const int p = ent[m] >> 8, x = ent[m] & 255;
            const uint32_t c = tgt[0][ent[m]];  // video.py:134
            if (MODE == kDHGR && c >= 0x80) {   // video.py:137
                err = kErrPaletteBit;
                break;
            }
            const uint32_t nd = ndv[m];
            const uint32_t w = dwf[p * 256 + y];
            const uint32_t dwy = (y == x) ? 0u : (w & 0x7fffu);       // video.py:141
            const bool nzy = (w & 0x8000u) && (y != x);               // video.py:140
            const int d = (int)nd - (int)dwy;                         // screen.py:547
            const bool cand = d < 0;                                  // video.py:283
            const unsigned long long bal = __ballot(cand);
            if (lane == 0) xw_cnt[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            const uint32_t c0 = xw_cnt[0], c1 = xw_cnt[1], c2 = xw_cnt[2], c3 = xw_cnt[3];
            const int C = (int)(c0 + c1 + c2 + c3);
            const int wbase = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
            uint32_t key = INF;
            if (cand) {
                // one random.getrandbits(8) per candidate, ascending offset (video.py:290-293)
                int j = mt_idx + wbase + prefix_popc(bal);
                uint32_t word = j < 624 ? mt[cb][j] : mt[cb ^ 1][j - 624];
                uint32_t nonce = mt_temper(word) >> 24;
                if (nzy)  // video.py:159
                    key = ((uint32_t)(d + 2048) << 17) | (nonce << 9) | ((uint32_t)y << 1) | (nd != 0 ? 1u : 0u);
            }
            uint32_t k1 = key, k2 = INF;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                uint32_t o1 = __shfl_xor(k1, s, 64), o2 = __shfl_xor(k2, s, 64);
                uint32_t lo = k1 < o1 ? k1 : o1, hi = k1 < o1 ? o1 : k1;
                uint32_t m2 = k2 < o2 ? k2 : o2;
                k1 = lo;
                k2 = hi < m2 ? hi : m2;
            }
            if (lane == 0) {
                xw_key[2 * wave] = k1;
                xw_key[2 * wave + 1] = k2;
            }
            __syncthreads();
            uint32_t K1 = INF, K2 = INF;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                uint32_t k = xw_key[q];
                if (k < K1) {
                    K2 = K1;
                    K1 = k;
                } else if (k < K2) {
                    K2 = k;
                }
            }
            const int y1 = K1 != INF ? (int)((K1 >> 1) & 255) : -1;
            const int f1 = K1 != INF ? (int)(K1 & 1) : 0;
            const int y2 = K2 != INF ? (int)((K2 >> 1) & 255) : -1;
            const int f2 = K2 != INF ? (int)(K2 & 1) : 0;
            if (n_pushed + f1 + f2 > kPushedCap) {
                err = kErrPushedOverflow;
                break;
            }

            // ---- apply (video.py:140-144, 170-178; screen.py:256-293)
            if (y == x) {
                dwf[p * 256 + x] = 0;
                S.up[is_aux][p * 256 + x] = 0;
                S.mem[is_aux][p * 256 + x] = (uint8_t)c;
            }
            if (y == y1 || y == y2) {
                const int second = (y == y2) ? 1 : 0;
                S.up[is_aux][p * 256 + y] = (int32_t)nd;  // byte_pair_difference == nd[y] (screen.py:383-398)
                S.mem[is_aux][p * 256 + y] = (uint8_t)c;
                dwf[p * 256 + y] = (uint16_t)((w & 0x7fffu) | (nd ? 0x8000u : 0u));
                if (nd) {
                    int j = mt_idx + C + (second ? f1 : 0);
                    uint32_t word = j < 624 ? mt[cb][j] : mt[cb ^ 1][j - 624];
                    uint32_t nonce = mt_temper(word) >> 24;  // video.py:178
                    S.pushed[n_pushed + (second ? f1 : 0)] =
                        ((2047u - nd) << 21) | (nonce << 13) | ((uint32_t)p << 8) | (uint32_t)y;
                }
            }
            if (tid == 0) {
                uint8_t *q = out + (size_t)done * 6;
                q[0] = (uint8_t)(p + 32);
                q[1] = (uint8_t)c;
                q[2] = (uint8_t)x;
                q[3] = (uint8_t)(y1 >= 0 ? y1 : x);  // video.py:185-186
                q[4] = (uint8_t)(y2 >= 0 ? y2 : x);
                q[5] = (uint8_t)x;
            }
            // later chunk entries that this step resolved exactly are now dead
#pragma unroll
            for (int m2 = 0; m2 < kChunk; m2++)
                if (m2 > m && m2 < cnt) {
                    if (y1 >= 0 && !f1 && ent[m2] == (uint32_t)((p << 8) | y1)) dead |= 1u << m2;
                    if (y2 >= 0 && !f2 && ent[m2] == (uint32_t)((p << 8) | y2)) dead |= 1u << m2;
                }
            mt_idx += C + f1 + f2;
            draws += (unsigned long long)(C + f1 + f2);
            n_pushed += f1 + f2;
            done++;
            if (!from_pushed) head = pos[m] + 1;
            if (mt_idx >= 624) {
                __syncthreads();  // every lane is done with block cb
                mt_twist<256>(mt[cb ^ 1], mt[cb], tid);
                cb ^= 1;
                mt_idx -= 624;
            }
        }
        if (!from_pushed && !err && done < n_ops) head = chunk_end > head ? chunk_end : head;
    }

    // ---- write the generator back (flags as bitmaps; diff weights themselves are immutable)
    __syncthreads();
    for (int wi = tid; wi < 256; wi += 256) {
        uint32_t nzw = 0, pdw = S.pdone[wi];
        for (int b = 0; b < 32; b++) {
            uint32_t v = dwf[wi * 32 + b];
            nzw |= ((v >> 15) & 1u) << b;
            // a byte whose diff weight was non-zero at the prologue and is zero now was a primary
            if ((v & 0x7fffu) == 0 && (S.wd[wi * 32 + b] >> 16) != 0) pdw |= 1u << b;
        }
        S.nzbits[wi] = nzw;
        S.pdone[wi] = pdw;
    }
    for (int i = tid; i < 624; i += 256) S.mt_py[i] = mt[cb][i];
    if (tid == 0) {
        S.mt_py_idx = mt_idx;
        S.head = head;
        S.n_pushed = n_pushed;
        S.exhausted = exhausted;
        if (exhausted) S.out_of_work[is_aux] = 1;
        S.draws_py += draws;
        S.ops += (unsigned long long)done;
        S.pad_ops += pad_ops;
        if (err && S.error == 0) S.error = err;
    }
}

// ------------------------------------------------------------------------- greedy, one wave per stream
//
// Same algorithm, one 64-lane wave per stream: lane l owns page bytes 4l..4l+3.
// Everything that is uniform per opcode (pop, validity, candidate counts, the two
// winners, RNG cursor, opcode emission) lives in SGPRs -- values loaded from the
// stream state are passed through v_readfirstlane so that the compiler keeps the
// bookkeeping on the scalar unit and the control flow in scalar branches.  There is
// no workgroup barrier and no LDS exchange in the loop.
//
// Scoring, fast form.  The reference orders candidates by (delta, nonce, offset)
// (video.py:290-301) and draws one nonce per candidate, eligible or not.  The two
// winners depend on the nonces only if two eligible candidates share the smallest or
// the second-smallest delta; otherwise the random stream just advances by the number
// of candidates.  So a step computes signed keys delta << 16 | offset, takes the wave
// minimum twice (fused-DPP min), and checks that no third eligible key shares the
// second delta.  Only if a tie is found (2.7 % of the opcodes of the bench workload)
// the entry is re-scored the long way, materialising the nonces exactly as the
// reference draws them.  LDS accesses of one wave execute in order, which is all the
// cross-lane ordering the loop needs.

#define IIV_SGPR(x) __builtin_amdgcn_readfirstlane((int)(x))
#ifndef IIV_WAVE_OCC
#define IIV_WAVE_OCC 6     // waves per SIMD the register allocation is held to (<= 80 VGPRs)
#endif
#ifndef IIV_WAVE_CHUNK
#define IIV_WAVE_CHUNK 2   // list entries whose rows and store-table values are fetched together
#endif

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

// The store table as the wave kernel reads it when every value fits 10 bits (DHGR):
// three values per 32-bit word, [offset][content][kP10Words], value t of a slice in
// word t / 3 at bit 10 * (t % 3).  A slice is 86 cache lines instead of 128, so the
// half of the table one launch uses (two of the four byte offsets) is 2.7 MiB and
// stays in the 4 MiB L2 of an XCD next to the streams' own state; 64 random lookups
// also fall into fewer distinct lines.  Built once per encoder from the u16 table.
constexpr int kP10Words = 2731;  // ceil(8192 / 3)

__global__ void pack10_kernel(const uint16_t *__restrict__ store, uint32_t *__restrict__ out, int n_slices,
                              int slice_len, uint32_t *__restrict__ d_max)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x, slice = blockIdx.y;
    if (q >= kP10Words || slice >= n_slices) return;
    const uint16_t *src = store + (size_t)slice * slice_len;
    uint32_t v[3];
#pragma unroll
    for (int r = 0; r < 3; r++) v[r] = 3 * q + r < slice_len ? src[3 * q + r] : 0u;
    out[(size_t)slice * kP10Words + q] = (v[0] & 1023u) | ((v[1] & 1023u) << 10) | ((v[2] & 1023u) << 20);
    const uint32_t mx = max(v[0], max(v[1], v[2]));
    if (mx >= 1024u) atomicMax(d_max, mx);
}

template <int MODE, bool P10>
__global__ __launch_bounds__(64, IIV_WAVE_OCC) void greedy_wave_kernel(StreamState *__restrict__ states,
                                                         const uint8_t *__restrict__ frames_main,
                                                         const uint8_t *__restrict__ frames_aux, int n_frames,
                                                         int frame, int is_aux, int n_ops,
                                                         const void *__restrict__ store_any,
                                                         uint8_t *__restrict__ ops_out, size_t ops_stride,
                                                         size_t ops_base)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    constexpr int CB = ModeTraits<MODE>::kContentBits;
    constexpr uint32_t INF = 0xffffffffu;
    constexpr uint32_t HI = 0xffff0000u;
    constexpr int M = IIV_WAVE_CHUNK;
    typedef uint32_t __attribute__((aligned(2))) u32_a2;
    // LDS per stream: two 1 KiB bitmaps + 1.4 MT19937 blocks (5.7 KiB): 24 streams per CU,
    // the limit the registers set.  The per-byte rows a step needs (target window |
    // diff weight, written once by the prologue and immutable while the generator
    // lives) are fetched from L2 for a whole chunk at a time.
    __shared__ uint32_t nz[256];     // update_priority != 0
    __shared__ uint32_t pdone[256];  // byte already emitted as a primary (its diff weight counts as 0)
    __shared__ uint32_t mt[624 + 256];  // random's current MT19937 block + the first 256 words of the next one
    __shared__ uint32_t xw[64];         // compaction of a list window

    const int lane = threadIdx.x;
    StreamState &S = states[blockIdx.x];
    uint8_t *out = ops_out + (size_t)blockIdx.x * ops_stride + ops_base;

    if (!S.gen_active || S.error) {
        if (lane == 0 && !S.error) S.error = kErrNoGenerator;
        return;
    }
    for (int i = lane; i < 256; i += 64) {
        nz[i] = S.nzbits[i];
        pdone[i] = S.pdone[i];
    }
    for (int i = lane; i < 624; i += 64) mt[i] = S.mt_py[i];
    __syncthreads();
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
    gen_ahead();
    int mt_idx = IIV_SGPR(S.mt_py_idx);
    if (mt_idx >= 624) {
        move_head();
        gen_rest();
        gen_ahead();
        mt_idx -= 624;
    }

    const int n_sorted = IIV_SGPR(S.n_sorted);
    const int truncated = IIV_SGPR(S.truncated);
    int head = IIV_SGPR(S.head), n_pushed = IIV_SGPR(S.n_pushed), exhausted = IIV_SGPR(S.exhausted);
    int done = 0, err = 0;
    unsigned long long draws = 0, pad_ops = 0;
    // slices of the even / odd page bytes of this bank, in 16-bit units
    constexpr size_t SLICE = P10 ? 2 * (size_t)kP10Words : (size_t)1 << BITS;
    const uint16_t *store_e = (const uint16_t *)store_any + ((size_t)byte_offset<MODE>(0, is_aux) << CB) * SLICE;
    const uint16_t *store_d = (const uint16_t *)store_any + ((size_t)byte_offset<MODE>(1, is_aux) << CB) * SLICE;
    const uint8_t *tgt_frames = (MODE == kDHGR && is_aux ? frames_aux : frames_main) +
                                ((size_t)blockIdx.x * n_frames + frame) * 8192;
    const uint4 *wd_rows = reinterpret_cast<const uint4 *>(S.wd);
    int32_t *up = S.up[is_aux];
    uint8_t *mem = S.mem[is_aux];
    const uint32_t pad_content = (uint32_t)IIV_SGPR(S.pad_content);
    const int wsel = lane >> 3;          // this lane's word inside a page's 8 bitmap words
    const int sh0 = (4 * lane) & 31;     // its 4 bits inside that word
    const int y0 = 4 * lane;

    // issue the four store-table loads of one entry (content c, row w); P10: the low
    // half of each row word (the window, not needed any more) is replaced by the bit
    // position of the value inside the loaded word
    auto gather4 = [&](uint4 &w, uint32_t c, uint32_t (&nd)[4]) {
        const size_t cbase = (size_t)(c & ((1u << CB) - 1)) * SLICE;
        const uint16_t *se = store_e + cbase, *sd = store_d + cbase;
        if (P10) {
            uint32_t *wr[4] = {&w.x, &w.y, &w.z, &w.w};
            // both lookups of a slice back to back: the second finds part of its lines in L1
            constexpr int ORD[4] = {0, 2, 1, 3};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = ORD[k];
                const uint32_t t = *wr[r] & 0xffffu;
                const uint32_t q = (t * 43691u) >> 17;  // t / 3 for t < 2^16
                nd[r] = reinterpret_cast<const uint32_t *>(r & 1 ? sd : se)[q];
                *wr[r] = (*wr[r] & HI) | ((t - 3u * q) * 10u);
            }
        } else {
            nd[0] = se[w.x & 0xffffu];
            nd[1] = sd[w.y & 0xffffu];
            nd[2] = se[w.z & 0xffffu];
            nd[3] = sd[w.w & 0xffffu];
        }
    };
    auto finish4 = [&](const uint4 &w, uint32_t (&nd)[4]) {
        if (P10) {
            nd[0] = __builtin_amdgcn_ubfe(nd[0], w.x, 10);  // v_bfe_u32 takes the offset from bits 4:0
            nd[1] = __builtin_amdgcn_ubfe(nd[1], w.y, 10);
            nd[2] = __builtin_amdgcn_ubfe(nd[2], w.z, 10);
            nd[3] = __builtin_amdgcn_ubfe(nd[3], w.w, 10);
        }
    };

    // per-byte keys of entry (p, x) from its wd row and store-table values:
    //   kt[r] = delta << 16 | y   with the reference's diff weight (0 for primaries, video.py:141)
    //   ke[r] = the same for bytes that may still be chosen (update_priority != 0, video.py:159),
    //           >= 0 for every other byte
    // C = number of candidates (delta < 0), i.e. nonces the reference draws (video.py:290-293);
    // *below = candidates in lower lanes (only the slow form needs it)
    auto score = [&](const uint4 &w, const uint32_t (&nd)[4], int p, int x, int (&kt)[4], int (&ke)[4], int &C,
                     int *below) -> bool {
        uint32_t nzw = nz[p * 8 + wsel], pdw = pdone[p * 8 + wsel];
        // video.py:130 -- skip a byte whose priority was cleared since the chunk was formed
        const uint32_t xw = (uint32_t)__builtin_amdgcn_readlane((int)nzw, (x >> 5) * 8);
        if (!((xw >> (x & 31)) & 1u)) return false;
        // x itself leaves both sets before the page is scored (video.py:140-141)
        const uint32_t xbit = wsel == (x >> 5) ? 1u << (x & 31) : 0u;
        nzw &= ~xbit;
        pdw |= xbit;
        const uint32_t wdr[4] = {w.x, w.y, w.z, w.w};
        C = 0;
        if (below) *below = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int sh = 31 - (sh0 + r);
            const uint32_t gone = (uint32_t)((int)(pdw << sh) >> 31);
            const uint32_t live = (uint32_t)((int)(nzw << sh) >> 31);
            const int n16 = (int)((nd[r] << 16) | (uint32_t)(y0 + r));
            kt[r] = n16 - (int)(wdr[r] & ~gone & HI);   // screen.py:547
            ke[r] = n16 - (int)(wdr[r] & live & HI);
            const unsigned long long bal = __ballot(kt[r] < 0);  // video.py:283
            C += (int)__popcll(bal);
            if (below) *below += prefix_popc(bal);
        }
        return true;
    };

    // store-table value of page byte y of the scored entry, as a scalar
    auto nd_of = [&](const uint32_t (&nd)[4], int y) -> uint32_t {
        const int r = y & 3;
        const uint32_t v = r == 0 ? nd[0] : r == 1 ? nd[1] : r == 2 ? nd[2] : nd[3];
        return (uint32_t)__builtin_amdgcn_readlane((int)v, y >> 2);
    };

    // opcodes emitted but not yet written: opcode ob_base + l sits in lane l of (ob0, ob1)
    uint32_t ob0 = 0, ob1 = 0;
    int ob_base = 0;
    auto flush_ops = [&]() {
        if (lane < done - ob_base) {
            uint8_t *q = out + (size_t)(ob_base + lane) * 6;
            *reinterpret_cast<u32_a2 *>(q) = ob0;
            *reinterpret_cast<uint16_t *>(q + 4) = (uint16_t)ob1;
        }
        ob_base = done;
    };

#ifdef IIV_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#define IIV_PHASE(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph[i] += now_ - ph_t;                                         \
        ph_t = now_;                                                  \
    } while (0)
#else
#define IIV_PHASE(i) do { } while (0)
#endif
    // after a block switch only words 0..255 of the current block are in place until twist_now()
    bool twist_pending = false;
    auto twist_now = [&]() {
        if (twist_pending) {
            gen_rest();
            gen_ahead();
            twist_pending = false;
        }
    };

    // video.py:140-144, 170-187; screen.py:256-293.  Lanes 0..2 carry (x, y1, y2); a
    // missing secondary repeats the primary's stores.  Returns false on overflow.
    auto apply = [&](int p, int x, uint32_t c, int y1, uint32_t nd1, int y2, uint32_t nd2, int C) -> bool {
        const uint32_t v1 = y1 >= 0 ? nd1 : 0u, v2 = y2 >= 0 ? nd2 : 0u;
        const int y1e = y1 >= 0 ? y1 : x, y2e = y2 >= 0 ? y2 : x;   // video.py:185-186
        const int f1 = v1 ? 1 : 0, f2 = v2 ? 1 : 0;
        if (n_pushed + f1 + f2 > kPushedCap) return false;
        if (mt_idx + C + 2 >= 256) twist_now();
        if (lane < 3) {
            const int off = lane == 0 ? x : lane == 1 ? y1e : y2e;
            const uint32_t val = lane == 0 ? 0u : lane == 1 ? v1 : v2;
            up[p * 256 + off] = (int32_t)val;  // byte_pair_difference == store-table value (screen.py:383-398)
            mem[p * 256 + off] = (uint8_t)c;
            if (val == 0) {
                atomicAnd(&nz[p * 8 + (off >> 5)], ~(1u << (off & 31)));
            } else {
                const int k = lane == 2 ? f1 : 0;
                const uint32_t nonce = mt_temper(mt[mt_idx + C + k]) >> 24;  // video.py:178
                S.pushed[n_pushed + k] = ((2047u - val) << 21) | (nonce << 13) | ((uint32_t)p << 8) | (uint32_t)off;
            }
            if (lane == 0) atomicOr(&pdone[p * 8 + (x >> 5)], 1u << (x & 31));
        }
        // the opcode (page + 32, content, x, y1, y2, x) goes into lane (done - ob_base) of a
        // register pair; 64 of them leave in two coalesced stores
        const bool mine = lane == done - ob_base;
        ob0 = mine ? (uint32_t)(p + 32) | (c << 8) | ((uint32_t)x << 16) | ((uint32_t)y1e << 24) : ob0;
        ob1 = mine ? (uint32_t)y2e | ((uint32_t)x << 8) : ob1;
        mt_idx += C + f1 + f2;
        draws += (unsigned long long)(C + f1 + f2);
        n_pushed += f1 + f2;
        done++;
        if (done - ob_base == 64) flush_ops();
        if (mt_idx >= 624) {
            // the next block becomes the current one: its head moves down now, the rest of it
            // and the new head are generated later, while store-table loads are in flight
            // (twist_now), at the latest before a step reads past word 255
            move_head();
            mt_idx -= 624;
            twist_pending = true;
        }
        return true;
    };

    // one greedy step on list entry e (row w, store-table values nd), fast form.  Returns
    // 0 = opcode emitted, 1 = entry skipped (its priority is gone), 2 = tie: nothing was
    // done and the caller re-scores the entry with slow_step, 3 = error (err is set).
    auto fast_step = [&](uint32_t e, const uint4 &w, const uint32_t (&nd)[4]) -> int {
        const int p = (e >> 8) & 31, x = e & 255;
        const uint32_t c = (e >> 16) & 0xffu;  // video.py:134
        int kt[4], ke[4], C;
        if (!score(w, nd, p, x, kt, ke, C, nullptr)) return 1;
        if (MODE == kDHGR && c >= 0x80) {  // video.py:137
            err = kErrPaletteBit;
            return 3;
        }
        // two smallest eligible keys: in the lane, then across the wave
        const int a0 = ke[0] < ke[1] ? ke[0] : ke[1], b0 = ke[0] < ke[1] ? ke[1] : ke[0];
        const int a1 = ke[2] < ke[3] ? ke[2] : ke[3], b1 = ke[2] < ke[3] ? ke[3] : ke[2];
        const int k1 = a0 < a1 ? a0 : a1, hi01 = a0 < a1 ? a1 : a0, mb = b0 < b1 ? b0 : b1;
        const int k2 = hi01 < mb ? hi01 : mb;
        const int K1 = wave_min_i32(k1);
        int y1 = -1, y2 = -1;
        uint32_t nd1 = 0, nd2 = 0;
        if (K1 < 0) {
            y1 = K1 & 255;
            const int K2 = wave_min_i32(k1 == K1 ? k2 : k1);
            if (K2 < 0) {
                y2 = K2 & 255;
                if ((K1 >> 16) == (K2 >> 16)) return 2;
                // does a third eligible byte share the second delta?
                const uint32_t D2 = (uint32_t)K2 >> 16;
                int n2 = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) n2 += (int)__popcll(__ballot(((uint32_t)ke[r] >> 16) == D2));
                if (n2 > 1) return 2;  // the nonces decide
                nd2 = nd_of(nd, y2);
            }
            nd1 = nd_of(nd, y1);
        }
        if (!apply(p, x, c, y1, nd1, y2, nd2, C)) {
            err = kErrPushedOverflow;
            return 3;
        }
        return 0;
    };

    // the same step in the reference's (delta, nonce, offset) heap order: every candidate's
    // nonce is materialised.  0 = emitted, 3 = error.
    auto slow_step = [&](uint32_t e, const uint4 &w, const uint32_t (&nd)[4]) -> int {
        const int p = (e >> 8) & 31, x = e & 255;
        const uint32_t c = (e >> 16) & 0xffu;
        int kt[4], ke[4], C, below;
        twist_now();
        score(w, nd, p, x, kt, ke, C, &below);  // the entry was live a moment ago: still is
        // one random.getrandbits(8) per candidate in ascending offset (video.py:290-293)
        uint32_t key[4];
        int run = mt_idx + below;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fj = run;
            run += kt[r] < 0 ? 1 : 0;
            const uint32_t nonce = mt_temper(mt[fj]) >> 24;
            const uint32_t k = ((uint32_t)((ke[r] >> 16) + 2048) << 16) | (nonce << 8) | (uint32_t)(y0 + r);
            key[r] = ke[r] < 0 ? k : INF;  // video.py:159
        }
        // two smallest (delta, nonce, offset): lane, row of 16 (DPP), wave (readlane)
        uint32_t a0 = key[0] < key[1] ? key[0] : key[1], b0 = key[0] < key[1] ? key[1] : key[0];
        uint32_t a1 = key[2] < key[3] ? key[2] : key[3], b1 = key[2] < key[3] ? key[3] : key[2];
        uint32_t k1 = a0 < a1 ? a0 : a1;
        uint32_t hi01 = a0 < a1 ? a1 : a0, mb = b0 < b1 ? b0 : b1;
        uint32_t k2 = hi01 < mb ? hi01 : mb;
        top2_step<0xB1>(k1, k2);   // quad_perm [1,0,3,2]
        top2_step<0x4E>(k1, k2);   // quad_perm [2,3,0,1]
        top2_step<0x141>(k1, k2);  // row_half_mirror
        top2_step<0x140>(k1, k2);  // row_mirror
        uint32_t K1 = INF, K2 = INF;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t r1 = __builtin_amdgcn_readlane(k1, 16 * q), r2 = __builtin_amdgcn_readlane(k2, 16 * q);
            uint32_t lo = K1 < r1 ? K1 : r1, hi = K1 < r1 ? r1 : K1;
            uint32_t m2 = K2 < r2 ? K2 : r2;
            K1 = lo;
            K2 = hi < m2 ? hi : m2;
        }
        const int y1 = K1 != INF ? (int)(K1 & 255) : -1;
        const int y2 = K2 != INF ? (int)(K2 & 255) : -1;
        const uint32_t nd1 = y1 >= 0 ? nd_of(nd, y1) : 0u, nd2 = y2 >= 0 ? nd_of(nd, y2) : 0u;
        if (!apply(p, x, c, y1, nd1, y2, nd2, C)) {
            err = kErrPushedOverflow;
            return 3;
        }
        return 0;
    };

    // ---- the sorted list is read 64 entries at a time; the entries of that window whose
    // priority is still non-zero are compacted into one register (lane k = k-th live entry,
    // through a 256 B LDS scatter), so that handing out the next chunk is two v_readlane
    // with a scalar index.  An entry that dies between compaction and use is skipped by the
    // step itself (video.py:130).
    // entries: page << 8 | offset | content << 16 | (list position - win_base) << 24
    int win_base = 0, win_end = 0;  // list positions [win_base, win_end) are in the window
    int n_dense = 0, qi = 0;        // live entries of the window, next one to hand out
    uint32_t dense_e = 0;
    auto refill = [&](int start) {
        win_base = start;
        win_end = start + 64 < n_sorted ? start + 64 : n_sorted;
        const int idx = start + lane;
        const uint32_t e = idx < n_sorted ? S.order[idx] : 0u;
        const uint32_t loc = e & 0x1fffu;
        const bool v = idx < n_sorted && ((nz[loc >> 5] >> (loc & 31)) & 1u);
        const unsigned long long mask = __ballot(v);
        n_dense = (int)__popcll(mask);
        qi = 0;
        if (v) xw[prefix_popc(mask)] = e | ((uint32_t)lane << 24);
        wave_lds_sync();
        dense_e = xw[lane];
        wave_lds_sync();
    };
    auto take = [&](int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)dense_e, k); };

    // current chunk (its rows are in flight or here) and the one after it (rows prefetched
    // while the current one is scored)
    uint32_t c_ent[M];
    uint4 c_rows[M];
    int c_cnt = 0, c_base = 0, c_end = 0;  // c_end: where `head` moves once the chunk is done (0: nowhere)
    bool have_cur = false;

    int guard = n_ops + 8192 + 2 * kPushedCap + 64;
    while (done < n_ops && !err) {
        if (--guard < 0) {
            err = kErrGuard;
            break;
        }
        if (exhausted) {
            flush_ops();
            for (int i = done + lane; i < n_ops; i += 64) {  // video.py:249-251
                uint8_t *q = out + (size_t)i * 6;
                q[0] = 32; q[1] = (uint8_t)pad_content; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0;
            }
            pad_ops += (unsigned long long)(n_ops - done);
            done = n_ops;
            ob_base = done;  // nothing buffered
            break;
        }

        if (head >= n_sorted) {
            // ---- the initial list is used up: pop the re-queued bag (video.py:124-131, 170-178)
            if (truncated) {  // more initial entries exist than were ordered: host budget bug
                err = kErrSortBudget;
                break;
            }
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
                continue;
            }
            if (lane == 0) S.pushed[bi] = INF;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // that store precedes the next scan of pushed[]
            // pushed keys do not carry the content byte: it is the target byte of that offset
            const uint32_t loc = bk & 0x1fffu;
            const uint32_t c = (uint32_t)IIV_SGPR(tgt_frames[loc]);
            const uint32_t e = loc | (c << 16);
            uint4 w = wd_rows[((e >> 8) & 31) * 64 + lane];
            uint32_t nd[4];
            gather4(w, c, nd);
            finish4(w, nd);
            int rc = fast_step(e, w, nd);
            if (rc == 2) rc = slow_step(e, w, nd);
            if (rc == 3) break;
            continue;
        }

        IIV_PHASE(0);  // loop overhead, pushed path
        if (!have_cur) {
            if (qi >= n_dense) {
                refill(head);
                if (n_dense == 0) {
                    head = win_end;
                    continue;
                }
            }
            c_cnt = n_dense - qi < M ? n_dense - qi : M;
#pragma unroll
            for (int m = 0; m < M; m++) c_ent[m] = take(qi + (m < c_cnt ? m : 0));  // spare slots repeat entry 0
            qi += c_cnt;
            c_base = win_base;
            c_end = qi >= n_dense ? win_end : 0;
#pragma unroll
            for (int m = 0; m < M; m++) c_rows[m] = wd_rows[((c_ent[m] >> 8) & 31) * 64 + lane];
        }
        IIV_PHASE(1);  // forming the first chunk + its row loads
        // ---- this chunk's store-table values (branch-free on purpose: a branch makes the
        // compiler retire each slot's loads before issuing the next slot's)
        uint32_t ndv[M][4];
#pragma unroll
        for (int m = 0; m < M; m++) gather4(c_rows[m], c_ent[m] >> 16, ndv[m]);
        IIV_PHASE(2);  // waiting for the rows + issuing the store-table loads
        // ---- meanwhile: start fetching the rows of the next chunk of this window
        uint32_t n_ent[M];
        uint4 n_rows[M];
        const int n_cnt = n_dense - qi < M ? n_dense - qi : M;
        const int n_base = win_base;
        int n_end = 0;
        if (n_cnt > 0) {
#pragma unroll
            for (int m = 0; m < M; m++) n_ent[m] = take(qi + (m < n_cnt ? m : 0));
            qi += n_cnt;
            n_end = qi >= n_dense ? win_end : 0;
#pragma unroll
            for (int m = 0; m < M; m++) n_rows[m] = wd_rows[((n_ent[m] >> 8) & 31) * 64 + lane];
        }
        IIV_PHASE(3);  // next chunk
        twist_now();  // (the MT19937 block generation hides behind the loads)
        IIV_PHASE(4);  // twist
        // vmcnt is one in-order counter for loads AND stores: retire the gathers once, here,
        // before the steps below start issuing stores.
        // (hipcc does not track a builtin s_waitcnt in its scoreboard, so the loaded
        // registers are passed through empty asm statements: the compiler then places its
        // own, accurate waits here and treats the values as plain registers afterwards)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < M; m++) {
            asm volatile("" : "+v"(c_rows[m].x), "+v"(c_rows[m].y), "+v"(c_rows[m].z), "+v"(c_rows[m].w));
            asm volatile("" : "+v"(ndv[m][0]), "+v"(ndv[m][1]), "+v"(ndv[m][2]), "+v"(ndv[m][3]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < M; m++) finish4(c_rows[m], ndv[m]);
        IIV_PHASE(5);  // waiting for the store-table values

        // ---- process the chunk
#pragma unroll
        for (int m = 0; m < M; m++) {
            if (m >= c_cnt || done >= n_ops || err) break;
            int rc = fast_step(c_ent[m], c_rows[m], ndv[m]);
            if (rc == 2) rc = slow_step(c_ent[m], c_rows[m], ndv[m]);
            if (rc == 3) break;
            head = c_base + (int)(c_ent[m] >> 24) + 1;
        }
        IIV_PHASE(6);  // scoring + applying the chunk
        if (err || done >= n_ops) break;
        head = c_end > head ? c_end : head;
        // the prefetched chunk becomes the current one
        have_cur = n_cnt > 0;
        if (have_cur) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                c_ent[m] = n_ent[m];
                c_rows[m] = n_rows[m];
            }
            c_cnt = n_cnt;
            c_base = n_base;
            c_end = n_end;
        }
    }

    flush_ops();
    twist_now();
    __syncthreads();
    for (int i = lane; i < 256; i += 64) {
        S.nzbits[i] = nz[i];
        S.pdone[i] = pdone[i];
    }
    for (int i = lane; i < 624; i += 64) S.mt_py[i] = mt[i];
    if (lane == 0) {
        S.mt_py_idx = mt_idx;
        S.head = head;
        S.n_pushed = n_pushed;
        S.exhausted = exhausted;
        if (exhausted) S.out_of_work[is_aux] = 1;
        S.draws_py += draws;
        S.ops += (unsigned long long)done;
        S.pad_ops += pad_ops;
        if (err && S.error == 0) S.error = err;
#ifdef IIV_STAMPS
        for (int i = 0; i < 8; i++) S.stamps[16 + i] = ph[i];
#endif
    }
}
#undef IIV_PHASE

// stand-alone packed view of the current screen for IIV_STATE_PACKED: reuse iiv_bitmap's pack

// ------------------------------------------------------------------------- host object

struct Encoder {
    int mode;
    int n_streams;
    const uint16_t *d_table;
    const uint16_t *d_store;
    uint32_t *d_store10;    // DHGR: d_store repacked 3 x 10 bit (see pack10_kernel); null if a value needs more
    ulonglong2 *d_strings;  // colour string of every masked value (recurrence mode)
    uint16_t *d_sub;        // 16x16 substitute costs
    int dw_mode;            // IIV_DW_TABLE / IIV_DW_RECURRENCE
    int greedy_mode;        // IIV_GREEDY_WAVE / IIV_GREEDY_WORKGROUP
    int partial_sort;       // allow the prologue's prefix sort when the budget is known
    int packed_store;       // let the wave kernel use d_store10
    StreamState *d_states;
    StreamState *d_snapshot;  // iiv_encoder_snapshot copy (lazily allocated)
    // generator bookkeeping shared by all streams (same schedule)
    int gen_active, gen_is_aux, gen_frame;
    int snap_gen_active, snap_gen_is_aux, snap_gen_frame;
    // profiling
    int profiling;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;  // class of interval i = events [2i, 2i+1]
    double ms[2];
    int64_t launches[2];
};

static void seed_by_array(uint32_t mt[624], const uint32_t *key, int n)
{
    mt[0] = 19650218u;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int i = 1, j = 0;
    for (int k = 624 > n ? 624 : n; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= n) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

static void seed_genrand(uint32_t mt[624], uint32_t s)
{
    mt[0] = s;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

int encoder_create(int mode, const uint16_t *d_table, const uint16_t *d_store, const int32_t *dm, int n_streams,
                   Encoder **out)
{
    if ((mode != kHGR && mode != kDHGR) || (!d_table && !dm) || !d_store || n_streams <= 0 || !out)
        return set_error(IIV_ERR_INVALID, "iiv_encoder_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(IIV_ERR_NO_DEVICE, "iiv_encoder_create: no HIP device");
    Encoder *e = new Encoder();
    e->mode = mode;
    e->n_streams = n_streams;
    e->d_table = d_table;
    e->d_store = d_store;
    e->d_store10 = nullptr;
    e->d_states = nullptr;
    e->d_snapshot = nullptr;
    e->d_strings = nullptr;
    e->d_sub = nullptr;
    e->dw_mode = dm ? IIV_DW_RECURRENCE : IIV_DW_TABLE;
    e->greedy_mode = IIV_GREEDY_AUTO;
    e->partial_sort = 1;
    e->packed_store = 1;
    e->gen_active = 0;
    e->gen_is_aux = 0;
    e->gen_frame = 0;
    e->profiling = 0;
    e->ms[0] = e->ms[1] = 0;
    e->launches[0] = e->launches[1] = 0;
    hipError_t he = hipMalloc(&e->d_states, sizeof(StreamState) * (size_t)n_streams);
    if (he != hipSuccess) {
        delete e;
        return hip_check(he, "hipMalloc(stream states)");
    }
    if (dm) {
        int rc = build_strings(mode, dm, &e->d_strings, &e->d_sub, 0);
        if (rc) {
            (void)hipFree(e->d_states);
            delete e;
            return rc;
        }
    }
    if (mode == kDHGR) {
        // 10-bit repack of the store table for the wave kernel; kept only if lossless
        const int n_slices = 4 << ModeTraits<kDHGR>::kContentBits, slice_len = 1 << ModeTraits<kDHGR>::kBits;
        uint32_t *d_max = nullptr, h_max = 0;
        bool ok = hipDeviceSynchronize() == hipSuccess &&  // d_store may have been built on any stream
                  hipMalloc(&e->d_store10, (size_t)n_slices * kP10Words * 4) == hipSuccess &&
                  hipMalloc(&d_max, 4) == hipSuccess && hipMemset(d_max, 0, 4) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(pack10_kernel, dim3((kP10Words + 255) / 256, n_slices), dim3(256), 0, 0, d_store,
                               e->d_store10, n_slices, slice_len, d_max);
            ok = hipMemcpy(&h_max, d_max, 4, hipMemcpyDeviceToHost) == hipSuccess && h_max < 1024u;
        }
        if (d_max) (void)hipFree(d_max);
        if (!ok) {
            (void)hipGetLastError();
            if (e->d_store10) (void)hipFree(e->d_store10);
            e->d_store10 = nullptr;
        }
    }
    // Video.__init__ (video.py:21-62): blank screen, zero priorities.  RNG
    // streams default to random.seed(0) / np.random.seed(0).
    StreamState *h = (StreamState *)calloc(1, sizeof(StreamState));
    uint32_t key0 = 0;
    seed_by_array(h->mt_py, &key0, 1);
    h->mt_py_idx = 624;
    seed_genrand(h->mt_np, 0);
    h->mt_np_idx = 624;
    int rc = IIV_OK;
    for (int s = 0; s < n_streams && !rc; s++)
        rc = hip_check(hipMemcpy(e->d_states + s, h, sizeof(StreamState), hipMemcpyHostToDevice), "init state");
    free(h);
    if (rc) {
        (void)hipFree(e->d_states);
        delete e;
        return rc;
    }
    *out = e;
    return IIV_OK;
}

void encoder_destroy(Encoder *e)
{
    if (!e) return;
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->d_states) (void)hipFree(e->d_states);
    if (e->d_snapshot) (void)hipFree(e->d_snapshot);
    if (e->d_strings) (void)hipFree(e->d_strings);
    if (e->d_sub) (void)hipFree(e->d_sub);
    if (e->d_store10) (void)hipFree(e->d_store10);
    delete e;
}

int encoder_snapshot(Encoder *e, hipStream_t st)
{
    if (!e) return set_error(IIV_ERR_INVALID, "snapshot: null encoder");
    const size_t bytes = sizeof(StreamState) * (size_t)e->n_streams;
    if (!e->d_snapshot) IIV_HIP(hipMalloc(&e->d_snapshot, bytes));
    IIV_HIP(hipMemcpyAsync(e->d_snapshot, e->d_states, bytes, hipMemcpyDeviceToDevice, st));
    e->snap_gen_active = e->gen_active;
    e->snap_gen_is_aux = e->gen_is_aux;
    e->snap_gen_frame = e->gen_frame;
    return IIV_OK;
}

int encoder_rollback(Encoder *e, hipStream_t st)
{
    if (!e || !e->d_snapshot) return set_error(IIV_ERR_INVALID, "rollback: no snapshot");
    const size_t bytes = sizeof(StreamState) * (size_t)e->n_streams;
    IIV_HIP(hipMemcpyAsync(e->d_states, e->d_snapshot, bytes, hipMemcpyDeviceToDevice, st));
    e->gen_active = e->snap_gen_active;
    e->gen_is_aux = e->snap_gen_is_aux;
    e->gen_frame = e->snap_gen_frame;
    return IIV_OK;
}

int encoder_set_option(Encoder *e, int option, int value)
{
    if (!e) return set_error(IIV_ERR_INVALID, "set_option: null encoder");
    if (option == IIV_OPT_DIFF_WEIGHTS) {
        if (value == IIV_DW_TABLE && !e->d_table) return set_error(IIV_ERR_INVALID, "no table was given at creation");
        if (value == IIV_DW_RECURRENCE && !e->d_strings)
            return set_error(IIV_ERR_INVALID, "no diff matrix was given at creation");
        if (value != IIV_DW_TABLE && value != IIV_DW_RECURRENCE) return set_error(IIV_ERR_INVALID, "bad value");
        e->dw_mode = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_PREFIX_SORT) {
        e->partial_sort = value ? 1 : 0;
        return IIV_OK;
    }
    if (option == IIV_OPT_PACKED_STORE) {
        e->packed_store = value ? 1 : 0;
        return IIV_OK;
    }
    if (option == IIV_OPT_GREEDY_KERNEL) {
        if (value != IIV_GREEDY_WAVE && value != IIV_GREEDY_WORKGROUP && value != IIV_GREEDY_AUTO)
            return set_error(IIV_ERR_INVALID, "bad value");
        e->greedy_mode = value;
        return IIV_OK;
    }
    return set_error(IIV_ERR_INVALID, "unknown option %d", option);
}

static int state_item(int mode, int what, size_t &off, size_t &bytes, bool &writable)
{
    writable = true;
    switch (what) {
    case IIV_STATE_MEM_MAIN: off = offsetof(StreamState, mem[0]); bytes = 8192; return 0;
    case IIV_STATE_MEM_AUX:
        if (mode != kDHGR) return -1;
        off = offsetof(StreamState, mem[1]); bytes = 8192; return 0;
    case IIV_STATE_UP_MAIN: off = offsetof(StreamState, up[0]); bytes = 8192 * 4; return 0;
    case IIV_STATE_UP_AUX:
        if (mode != kDHGR) return -1;
        off = offsetof(StreamState, up[1]); bytes = 8192 * 4; return 0;
    case IIV_STATE_OUT_OF_WORK: off = offsetof(StreamState, out_of_work); bytes = 8; return 0;
    case IIV_STATE_COUNTERS: off = offsetof(StreamState, draws_py); bytes = 32; writable = false; return 0;
    case 100: off = offsetof(StreamState, stamps); bytes = 256; writable = false; return 0;  // diagnostic
    default: return -1;
    }
}

int encoder_get_state(Encoder *e, int s, int what, void *buf, size_t bytes)
{
    if (!e || !buf || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "get_state: bad argument");
    IIV_HIP(hipDeviceSynchronize());
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s);
    if (what == IIV_STATE_RNG_PY || what == IIV_STATE_RNG_NP) {
        if (bytes != 625 * 4) return set_error(IIV_ERR_INVALID, "get_state: RNG state is 625 u32");
        size_t o_mt = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py) : offsetof(StreamState, mt_np);
        size_t o_ix = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py_idx) : offsetof(StreamState, mt_np_idx);
        IIV_HIP(hipMemcpy(buf, base + o_mt, 624 * 4, hipMemcpyDeviceToHost));
        IIV_HIP(hipMemcpy((uint8_t *)buf + 624 * 4, base + o_ix, 4, hipMemcpyDeviceToHost));
        return IIV_OK;
    }
    if (what == IIV_STATE_PACKED) {
        if (bytes != 4096 * 8) return set_error(IIV_ERR_INVALID, "get_state: packed is 32x128 u64");
        uint64_t *d_p = nullptr;
        IIV_HIP(hipMalloc(&d_p, 4096 * 8));
        int rc = pack(e->mode, 1, base + offsetof(StreamState, mem[0]), base + offsetof(StreamState, mem[1]), d_p, 0);
        if (!rc) rc = hip_check(hipMemcpy(buf, d_p, 4096 * 8, hipMemcpyDeviceToHost), "copy packed");
        (void)hipFree(d_p);
        return rc;
    }
    size_t off, want;
    bool writable;
    if (state_item(e->mode, what, off, want, writable)) return set_error(IIV_ERR_INVALID, "get_state: unknown item %d", what);
    if (bytes != want) return set_error(IIV_ERR_INVALID, "get_state: item %d is %zu bytes, got %zu", what, want, bytes);
    IIV_HIP(hipMemcpy(buf, base + off, bytes, hipMemcpyDeviceToHost));
    return IIV_OK;
}

int encoder_set_state(Encoder *e, int s, int what, const void *buf, size_t bytes)
{
    if (!e || !buf || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "set_state: bad argument");
    IIV_HIP(hipDeviceSynchronize());
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s);
    if (what == IIV_STATE_RNG_PY || what == IIV_STATE_RNG_NP) {
        if (bytes != 625 * 4) return set_error(IIV_ERR_INVALID, "set_state: RNG state is 625 u32");
        uint32_t idx = ((const uint32_t *)buf)[624];
        if (idx > 624) return set_error(IIV_ERR_INVALID, "set_state: RNG index %u > 624", idx);
        size_t o_mt = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py) : offsetof(StreamState, mt_np);
        size_t o_ix = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py_idx) : offsetof(StreamState, mt_np_idx);
        IIV_HIP(hipMemcpy(base + o_mt, buf, 624 * 4, hipMemcpyHostToDevice));
        IIV_HIP(hipMemcpy(base + o_ix, (const uint8_t *)buf + 624 * 4, 4, hipMemcpyHostToDevice));
        return IIV_OK;
    }
    size_t off, want;
    bool writable;
    if (state_item(e->mode, what, off, want, writable) || !writable)
        return set_error(IIV_ERR_INVALID, "set_state: item %d is not settable", what);
    if (bytes != want) return set_error(IIV_ERR_INVALID, "set_state: item %d is %zu bytes, got %zu", what, want, bytes);
    IIV_HIP(hipMemcpy(base + off, buf, bytes, hipMemcpyHostToDevice));
    return IIV_OK;
}

static int prof_begin(Encoder *e, int cls, hipStream_t st, size_t &slot)
{
    slot = e->ev_class.size();
    if (e->ev_pool.size() < 2 * (slot + 1)) {
        hipEvent_t a, b;
        IIV_HIP(hipEventCreate(&a));
        IIV_HIP(hipEventCreate(&b));
        e->ev_pool.push_back(a);
        e->ev_pool.push_back(b);
    }
    e->ev_class.push_back(cls);
    IIV_HIP(hipEventRecord(e->ev_pool[2 * slot], st));
    return IIV_OK;
}

static int prof_end(Encoder *e, size_t slot, hipStream_t st)
{
    IIV_HIP(hipEventRecord(e->ev_pool[2 * slot + 1], st));
    return IIV_OK;
}

static int prof_flush(Encoder *e)
{
    for (size_t i = 0; i < e->ev_class.size(); i++) {
        IIV_HIP(hipEventSynchronize(e->ev_pool[2 * i + 1]));
        float ms = 0;
        IIV_HIP(hipEventElapsedTime(&ms, e->ev_pool[2 * i], e->ev_pool[2 * i + 1]));
        e->ms[e->ev_class[i]] += ms;
        e->launches[e->ev_class[i]] += 1;
    }
    e->ev_class.clear();
    return IIV_OK;
}

int encoder_profile(Encoder *e, int enable)
{
    if (!e) return set_error(IIV_ERR_INVALID, "profile: null encoder");
    e->profiling = enable ? 1 : 0;
    if (enable) {
        e->ev_class.clear();
        e->ms[0] = e->ms[1] = 0;
        e->launches[0] = e->launches[1] = 0;
    }
    return IIV_OK;
}

int encoder_profile_read(Encoder *e, double ms[2], int64_t launches[2])
{
    if (!e) return set_error(IIV_ERR_INVALID, "profile_read: null encoder");
    int rc = prof_flush(e);
    if (rc) return rc;
    ms[0] = e->ms[0];
    ms[1] = e->ms[1];
    launches[0] = e->launches[0];
    launches[1] = e->launches[1];
    return IIV_OK;
}

int encode(Encoder *e, const uint8_t *d_main, const uint8_t *d_aux, int n_frames, const iiv_segment *segs, int n_segs,
           uint8_t *d_ops, hipStream_t st)
{
    if (!e || !d_main || !segs || n_segs < 0 || n_frames <= 0 || (!d_ops && n_segs > 0))
        return set_error(IIV_ERR_INVALID, "iiv_encode: bad argument");
    if (e->mode == kDHGR && !d_aux) return set_error(IIV_ERR_INVALID, "iiv_encode: DHGR needs aux frames");
    size_t total = 0;
    for (int i = 0; i < n_segs; i++) {
        const iiv_segment &g = segs[i];
        if (g.n_ops < 0 || g.frame < 0 || g.frame >= n_frames || (g.is_aux != 0 && g.is_aux != 1) ||
            (e->mode == kHGR && g.is_aux))
            return set_error(IIV_ERR_INVALID, "iiv_encode: bad segment %d", i);
        total += (size_t)g.n_ops;
    }
    const size_t stride = total * 6;
    size_t done = 0;
    for (int i = 0; i < n_segs; i++) {
        const iiv_segment &g = segs[i];
        if (g.n_ops == 0) {
            // encode_frame() only creates a lazy generator (video.py:72-93): remember
            // it, run nothing.  A later restart == 0 segment will start it.
            if (g.restart) {
                e->gen_active = 2;  // created, prologue pending
                e->gen_is_aux = g.is_aux;
                e->gen_frame = g.frame;
            }
            continue;
        }
        bool need_prologue = g.restart != 0;
        if (!g.restart) {
            if (!e->gen_active) return set_error(IIV_ERR_INVALID, "iiv_encode: segment %d continues no generator", i);
            if (e->gen_is_aux != g.is_aux || e->gen_frame != g.frame)
                return set_error(IIV_ERR_INVALID, "iiv_encode: segment %d continues a different target/bank", i);
            if (e->gen_active == 2) need_prologue = true;
        }
        size_t slot = 0;
        int need = 0;
        if (need_prologue && e->partial_sort) {
            // opcodes this generator can be asked for = n_ops of this segment and of the
            // restart == 0 segments that follow it -- known only if another restart comes
            // later in this call (otherwise a later call might continue the generator)
            long budget = g.n_ops;
            bool closed = false;
            for (int k = i + 1; k < n_segs; k++) {
                if (segs[k].restart) {
                    closed = true;
                    break;
                }
                budget += segs[k].n_ops;
            }
            if (closed && 3 * budget <= kSelNeedMax) need = (int)(3 * budget);
        }
        if (need_prologue) {
            if (e->profiling) { int prc = prof_begin(e, 0, st, slot); if (prc) return prc; }
            const bool dp = e->dw_mode == IIV_DW_RECURRENCE;
#define IIV_PRO(M, D)                                                                                              \
    hipLaunchKernelGGL((prologue_kernel<M, D>), dim3(e->n_streams), dim3(kProThreads), 0, st, e->d_states, d_main, \
                       d_aux, n_frames, g.frame, g.is_aux, e->d_table, e->d_strings, e->d_sub, need)
            if (e->mode == kDHGR) {
                if (dp) IIV_PRO(kDHGR, true); else IIV_PRO(kDHGR, false);
            } else {
                if (dp) IIV_PRO(kHGR, true); else IIV_PRO(kHGR, false);
            }
#undef IIV_PRO
            IIV_HIP(hipGetLastError());
            if (e->profiling) { int prc = prof_end(e, slot, st); if (prc) return prc; }
        }
        e->gen_active = 1;
        e->gen_is_aux = g.is_aux;
        e->gen_frame = g.frame;
        if (e->profiling) { int prc = prof_begin(e, 1, st, slot); if (prc) return prc; }
#define IIV_GREEDY(K, T, STORE)                                                                                 \
    hipLaunchKernelGGL(K, dim3(e->n_streams), dim3(T), 0, st, e->d_states, d_main, d_aux, n_frames, g.frame,       \
                       g.is_aux, g.n_ops, STORE, d_ops, stride, done * 6)
        // few streams: the 4-wave workgroup has the shorter dependent chain per opcode;
        // many streams: the one-wave kernel keeps 16 streams resident per CU instead of 4
        const bool use_wave = e->greedy_mode == IIV_GREEDY_WAVE ||
                              (e->greedy_mode == IIV_GREEDY_AUTO && e->n_streams >= 1536);
        if (use_wave) {
            if (e->mode == kDHGR && e->d_store10 && e->packed_store) IIV_GREEDY((greedy_wave_kernel<kDHGR, true>), 64, (const void *)e->d_store10);
            else if (e->mode == kDHGR) IIV_GREEDY((greedy_wave_kernel<kDHGR, false>), 64, (const void *)e->d_store);
            else IIV_GREEDY((greedy_wave_kernel<kHGR, false>), 64, (const void *)e->d_store);
        } else {
            if (e->mode == kDHGR) IIV_GREEDY(greedy_kernel<kDHGR>, 256, e->d_store); else IIV_GREEDY(greedy_kernel<kHGR>, 256, e->d_store);
        }
#undef IIV_GREEDY
        IIV_HIP(hipGetLastError());
        if (e->profiling) { int prc = prof_end(e, slot, st); if (prc) return prc; }
        done += (size_t)g.n_ops;
    }
    return IIV_OK;
}

__global__ void error_scan_kernel(const StreamState *states, int n, int *result)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && states[i].error) {
        atomicMin(&result[0], i);
    }
}

int encoder_check(Encoder *e, int *bad_stream, hipStream_t st)
{
    if (!e) return set_error(IIV_ERR_INVALID, "check: null encoder");
    int *d_res = nullptr;
    IIV_HIP(hipMalloc(&d_res, sizeof(int)));
    int init = 0x7fffffff;
    int rc = hip_check(hipMemcpyAsync(d_res, &init, sizeof(int), hipMemcpyHostToDevice, st), "check init");
    if (!rc) {
        hipLaunchKernelGGL(error_scan_kernel, dim3((e->n_streams + 255) / 256), dim3(256), 0, st, e->d_states,
                           e->n_streams, d_res);
        rc = hip_check(hipGetLastError(), "error_scan launch");
    }
    int first = 0x7fffffff;
    if (!rc) rc = hip_check(hipMemcpyAsync(&first, d_res, sizeof(int), hipMemcpyDeviceToHost, st), "check read");
    if (!rc) rc = hip_check(hipStreamSynchronize(st), "check sync");
    (void)hipFree(d_res);
    if (rc) return rc;
    if (first == 0x7fffffff) return IIV_OK;
    if (bad_stream) *bad_stream = first;
    int32_t code = 0;
    IIV_HIP(hipMemcpy(&code, reinterpret_cast<uint8_t *>(e->d_states + first) + offsetof(StreamState, error), 4,
                      hipMemcpyDeviceToHost));
    static const char *names[] = {"",
                                  "memory map has non-zero screen-hole bytes (video.py:87)",
                                  "negative update_priority (video.py:117)",
                                  "DHGR content byte has the palette bit set (video.py:137)",
                                  "pushed-entry capacity exceeded",
                                  "next() on a stream with no generator",
                                  "internal: greedy loop guard tripped",
                                  "internal: prefix sort exhausted before the opcode budget"};
    int is_overflow = code == kErrPushedOverflow;
    return set_error(is_overflow ? IIV_ERR_OVERFLOW : IIV_ERR_ASSERT, "stream %d: %s", first,
                     (code > 0 && code < 8) ? names[code] : "unknown error");
}

}  // namespace iiv

// ------------------------------------------------------------------------- C ABI

struct iiv_encoder {
    iiv::Encoder *impl;
};

extern "C" {

int iiv_encoder_create(int mode, const uint16_t *d_table, const uint16_t *d_store_table, const int32_t dm[256],
                       int n_streams, iiv_encoder **out)
{
    if (!out) return iiv::set_error(IIV_ERR_INVALID, "iiv_encoder_create: out is NULL");
    iiv::Encoder *impl = nullptr;
    int rc = iiv::encoder_create(mode, d_table, d_store_table, dm, n_streams, &impl);
    if (rc) return rc;
    *out = new iiv_encoder{impl};
    return IIV_OK;
}

void iiv_encoder_destroy(iiv_encoder *enc)
{
    if (!enc) return;
    iiv::encoder_destroy(enc->impl);
    delete enc;
}

int iiv_encoder_snapshot(iiv_encoder *enc, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_snapshot(enc->impl, (hipStream_t)stream);
}

int iiv_encoder_rollback(iiv_encoder *enc, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_rollback(enc->impl, (hipStream_t)stream);
}

int iiv_encoder_set_option(iiv_encoder *enc, int option, int value)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_option(enc->impl, option, value);
}

int iiv_encoder_get_state(iiv_encoder *enc, int stream_index, int what, void *host_buf, size_t bytes)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_get_state(enc->impl, stream_index, what, host_buf, bytes);
}

int iiv_encoder_set_state(iiv_encoder *enc, int stream_index, int what, const void *host_buf, size_t bytes)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_state(enc->impl, stream_index, what, host_buf, bytes);
}

int iiv_encode(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux, int n_frames,
               const iiv_segment *segments, int n_segments, uint8_t *d_ops_out, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encode(enc->impl, d_frames_main, d_frames_aux, n_frames, segments, n_segments, d_ops_out,
                       (hipStream_t)stream);
}

int iiv_encoder_check(iiv_encoder *enc, int *bad_stream, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_check(enc->impl, bad_stream, (hipStream_t)stream);
}

int iiv_encoder_profile(iiv_encoder *enc, int enable)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_profile(enc->impl, enable);
}

int iiv_encoder_profile_read(iiv_encoder *enc, double ms[2], int64_t launches[2])
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_profile_read(enc->impl, ms, launches);
}

}  // extern "C"
