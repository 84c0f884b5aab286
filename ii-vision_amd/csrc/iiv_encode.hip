// iiv_encode.hip -- video.Video.encode_frame on gfx950: the encoder object, its launch planning and the C ABI
// (reference: transcoder/video.py:72-301, transcoder/screen.py:383-547, transcoder/movie.py:56-111).
//
// One workgroup (or wave) per independent video stream; all stream state lives in HBM (StreamState) between
// launches and in LDS / registers inside one.  A launch round = for every stream, optionally a new generator
// (iiv_prologue.hip: prologue_kernel) and then its next n_ops opcodes (iiv_greedy.hip / iiv_team.hip /
// iiv_workgroup.hip).
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
#include "iiv_stream.h"
#include "iiv_wave.h"

#include <stdlib.h>
#include <vector>

namespace iiv {

constexpr int kSharedHgrMinStreams = 4096;   // 16 streams per workgroup x 256 CUs
constexpr int kSharedDhgrMinStreams = 2048;  // (at 1024 clips the LDS-shared form is 3 % behind the plain one)
// The LDS-shared form of the one-wave kernel keeps 16 waves per CU busy, the plain form 28.  On input whose steps are
// rarely decided by the nonces the shared form wins (DHGR +1-7 %, HGR +3-12 %); on picture-like input, where nearly every
// step takes the exact-nonce path with its extra wave reductions and LDS reads, 16 waves hide that latency worse than 28
// and it loses 5 %.  Both forms produce the same bytes, so the encoder picks by what its own kernels saw: they count the
// steps the nonces decided and the opcodes emitted, the counters come back with an asynchronous copy (never waited for:
// a call uses what an earlier call's copy has delivered), and a batch above this share runs the plain form.
// Round 6, re-measured per content (tools/content_kernel_split.py, greedy ms per 14336-stream launch, shared / plain):
//   DHGR  S-iid (share 0.03) 1.01 / 1.22   error-diffusion frames (0.14) 1.02 / 1.24   ordered-dither frames (0.36) 1.06 / 1.25
//         S-img (0.90) 1.30 / 1.30   -- two rounds of diets later the shared form is the better one up to ~85 %;
//   HGR   S-iid (0.02) 2.83 / 3.45   error-diffusion frames (0.11) 2.82 / 3.25   ordered-dither frames (0.36) 3.67 / 3.33
//         S-img (0.85) 6.26 / 3.37   -- its shared form keeps MT19937 in registers: the exact-nonce path is dear there.
constexpr unsigned kTieHeavyPercentDHGR = 85, kTieHeavyPercentHGR = 30;
// ... and so does a batch whose streams have little left to do per launch (converging content that is mostly out of work:
// a frame shown four times runs 5.7 M frames/s plain, 4.6 M shared -- the workgroups' table copy and stream queue are
// overhead that a launch of a few dozen real opcodes per stream does not repay)
constexpr unsigned kSharedMinOpsPerLaunch = 96;
constexpr int kTeamMaxStreams = 768;   // IIV_GREEDY_AUTO: at most this many streams run the team kernel
// Longest first (iiv_stream.h: GreedyArgs::perm): batches of at least this many streams run their one-wave launches in the
// order of what the streams' latest launches cost, re-sorted every kOrderEvery greedy launches (a stream's cost follows its
// content, which changes slowly; the sort is one 1024-thread workgroup, ~10 us)
constexpr int kOrderMinStreams = 2048;
constexpr int kOrderEvery = 4;


// ------------------------------------------------------------------------- host object

struct GenState {
    int active;   // 0 = no generator, 1 = live on the device, 2 = created (lazily), prologue pending
    int is_aux, frame;
};

constexpr int kSmallSlots = 4;          // iiv_encoder_set_state_async's staging ring
constexpr size_t kSmallBytes = 2560;    // >= 625 words

struct Encoder {
    int mode;
    int n_streams;
    const uint16_t *d_table;
    const uint16_t *d_store;
    uint32_t *d_left, *d_right;  // split store table (iiv_stream.h), built at creation when dm is given
    NarrowTables nt;             // the narrow form of the split store table the greedy kernels read (iiv_stream.h)
    uint32_t *d_dwl, *d_dwr;     // split diff-weight table, likewise
    uint32_t *d_left_t, *d_right_t;  // split store table, content innermost (IIV_CONTENT_JOINT_SPLIT; built on first use)
    uint32_t *d_joint_l, *d_joint_r; // narrow form, content innermost, two byte values per word (IIV_CONTENT_JOINT; built on first use)
    void *d_brief;                   // iiv_encoder_get_video_brief's staging struct
    ulonglong2 *d_strings;  // colour string of every masked value (recurrence mode)
    uint32_t *d_hgr_dots;   // HGR: the window -> dots lookups the prologue copies into LDS (iiv_edit.h: hgr_dot_slot_lo)
    uint32_t *d_dw_pieces;  // the diff weights' pair-term table the prologue copies into LDS (iiv_tables.hip: dw_piece_kernel)
    uint16_t *d_sub;        // 16x16 substitute costs
    int dw_mode;            // IIV_DW_TABLE / IIV_DW_RECURRENCE
    int greedy_mode;        // IIV_GREEDY_WAVE / IIV_GREEDY_WORKGROUP / IIV_GREEDY_AUTO
    int partial_sort;       // allow the prologue's prefix sort when the budget is known
    int greedy_lds_pad;     // IIV_OPT_GREEDY_LDS_PAD
    int content_choice;     // IIV_OPT_CONTENT_CHOICE
    int fourth_offset;      // IIV_OPT_FOURTH_OFFSET
    StreamState *d_states;
    StreamState *d_snapshot[2];  // iiv_encoder_snapshot copies (lazily allocated; slot 1: iiv_encoder_snapshot_slot)
    // iiv_encoder_set_state_async: a small ring of pinned staging slots (one allocation), an event behind each slot's copies
    uint8_t *h_small[1];
    hipEvent_t small_ev[4];
    int small_slot;
    // live hand-over (iiv_encode_live): two opcode queues in coherent host memory (lazily allocated), and the one / the tag
    // the launches of the call in progress write to (NULL outside such a call)
    unsigned long long *h_live[2], *d_live[2], *live_now;   // (d_live: the same memory as the device addresses it)
    uint32_t live_tag;
    // generator bookkeeping: one entry while every stream has run the same schedule, else one per stream
    std::vector<GenState> gens, snap_gens[2];
    // launch descriptors: pinned staging ring -> device buffer, both grown on demand
    LaunchSeg *h_segs[2];
    hipEvent_t seg_ev[2];
    size_t h_cap[2];
    int seg_slot;
    LaunchSeg *d_segs;
    int *d_queue;           // one stream counter per launch round of a call (persistent greedy workgroups), zeroed per call
    size_t queue_cap;
    // what the one-wave kernels saw (kTieHeavyPercentDHGR / HGR): device counters, their pinned host copy, the event behind the copy
    unsigned long long *d_tie_stats, *h_tie_stats, tie_seen[3];
    hipEvent_t tie_ev;
    bool tie_copy_pending;
    int tie_heavy;          // -1: not known yet, 0 / 1
    double tie_rate;        // of the latest interval looked at
    double ops_per_launch;  // real (not padding) opcodes per stream and launch, likewise
    size_t d_cap;
    // scratch for encoder_check / IIV_STATE_PACKED
    int *d_result;
    uint64_t *d_packed;
    // profiling
    int profiling;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_class;  // class of interval i = events [2i, 2i+1]
    double ms[2];
    int64_t launches[2];
    int64_t form_launches[4];   // greedy launches since profiling was switched on: one-wave plain / LDS-shared, team, workgroup
    // longest-first launch order of the one-wave kernel: what every stream's latest launch cost, the permutation in use
    uint32_t *d_cost;
    int *d_perm;
    int order_countdown;        // greedy launches until the next re-sort
    int order_streams;          // IIV_OPT_STREAM_ORDER: 1 (default) / 0
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

// Video.__init__ (video.py:21-62) for every stream: blank screen, zero priorities, both RNG
// streams at their seed-0 state (rng0 = mt_py | mt_np, 1248 words)
__global__ __launch_bounds__(256) void init_states_kernel(StreamState *__restrict__ states, const uint32_t *__restrict__ rng0)
{
    StreamState &S = states[blockIdx.x];
    uint2 *q = reinterpret_cast<uint2 *>(&S);
    const uint2 z = make_uint2(0, 0);
    for (size_t i = threadIdx.x; i < sizeof(StreamState) / 8; i += 256) q[i] = z;
    __syncthreads();
    for (int i = threadIdx.x; i < 624; i += 256) {
        S.mt_py[i] = rng0[i];
        S.mt_np[i] = rng0[624 + i];
    }
    if (threadIdx.x == 0) {
        S.mt_py_idx = 624;
        S.mt_np_idx = 624;
    }
}
static_assert(sizeof(StreamState) % 8 == 0, "StreamState is cleared 8 bytes at a time");

// The priorities as the host sees them (int32, StreamState::up) from the kernels' 16-bit copy, and back (iiv_stream.h):
// one workgroup per stream, before the host reads / after it has written them.
__global__ __launch_bounds__(256) void materialise_up_kernel(StreamState *__restrict__ states)
{
    StreamState &S = states[blockIdx.x];
    for (int i = threadIdx.x; i < 2 * 8192; i += 256) {
        const uint32_t v = S.up16[i >> 13][i & 8191];
        if (v != kUpBig) S.up[i >> 13][i & 8191] = (int32_t)v;
    }
}
__global__ __launch_bounds__(256) void compact_up_kernel(StreamState *__restrict__ states)
{
    StreamState &S = states[blockIdx.x];
    for (int i = threadIdx.x; i < 2 * 8192; i += 256) {
        const uint32_t u = (uint32_t)S.up[i >> 13][i & 8191];
        S.up16[i >> 13][i & 8191] = (uint16_t)(u < kUpBig ? u : kUpBig);
    }
}
static int materialise_up(Encoder *e, int s0, int n)
{
    hipLaunchKernelGGL(materialise_up_kernel, dim3(n), dim3(256), 0, 0, e->d_states + s0);
    int rc = hip_check(hipGetLastError(), "materialise_up_kernel launch");
    return rc ? rc : hip_check(hipDeviceSynchronize(), "materialise_up sync");
}

// perm = the streams by descending cost (a counting sort over 1024 cost classes between the batch's cheapest and dearest
// stream; the order inside a class is whatever the atomics give -- any permutation is correct, a better one is faster).
// identity != 0: perm[i] = i.
__global__ __launch_bounds__(1024) void order_streams_kernel(const uint32_t *__restrict__ cost, int n, int *__restrict__ perm, int identity)
{
    __shared__ uint32_t lo_s, hi_s;
    __shared__ uint32_t count[1024], start[1024];
    const int tid = threadIdx.x;
    if (identity) {
        for (int i = tid; i < n; i += 1024) perm[i] = i;
        return;
    }
    if (tid == 0) lo_s = 0xffffffffu, hi_s = 0u;
    count[tid] = 0;
    __syncthreads();
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = tid; i < n; i += 1024) {
        const uint32_t c = cost[i];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
    atomicMin(&lo_s, lo);
    atomicMax(&hi_s, hi);
    __syncthreads();
    lo = lo_s, hi = hi_s;
    const unsigned long long span = (unsigned long long)(hi - lo) + 1ull;
    auto cls = [&](uint32_t c) -> int { return (int)(((unsigned long long)(hi - c) * 1024ull) / span); };   // 0 = the dearest
    for (int i = tid; i < n; i += 1024) atomicAdd(&count[cls(cost[i])], 1u);
    __syncthreads();
    // exclusive prefix sum over the 1024 classes (one per thread: a plain two-level scan)
    __shared__ uint32_t wave_tot[16];
    uint32_t v = count[tid], incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += wave_tot[w];
    start[tid] = base + incl - v;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) perm[atomicAdd(&start[cls(cost[i])], 1u)] = i;
}

__global__ __launch_bounds__(256) void max_u16_kernel(const uint16_t *__restrict__ v, size_t n, uint32_t *__restrict__ result)
{
    uint32_t mx = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) mx = max(mx, (uint32_t)v[i]);
    for (int d = 1; d < 64; d <<= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(result, mx);
}

void encoder_destroy(Encoder *e)
{
    if (!e) return;
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (int k = 0; k < 2; k++) {
        if (e->h_segs[k]) (void)hipHostFree(e->h_segs[k]);
        if (e->seg_ev[k]) (void)hipEventDestroy(e->seg_ev[k]);
    }
    if (e->d_segs) (void)hipFree(e->d_segs);
    if (e->d_queue) (void)hipFree(e->d_queue);
    if (e->d_tie_stats) (void)hipFree(e->d_tie_stats);
    if (e->h_tie_stats) (void)hipHostFree(e->h_tie_stats);
    if (e->tie_ev) (void)hipEventDestroy(e->tie_ev);
    if (e->d_result) (void)hipFree(e->d_result);
    if (e->d_cost) (void)hipFree(e->d_cost);
    if (e->d_perm) (void)hipFree(e->d_perm);
    if (e->d_packed) (void)hipFree(e->d_packed);
    if (e->d_states) (void)hipFree(e->d_states);
    for (int k = 0; k < 2; k++) {
        if (e->d_snapshot[k]) (void)hipFree(e->d_snapshot[k]);
        if (e->h_live[k]) (void)hipHostFree(e->h_live[k]);
    }
    if (e->h_small[0]) (void)hipHostFree(e->h_small[0]);
    for (int k = 0; k < kSmallSlots; k++)
        if (e->small_ev[k]) (void)hipEventDestroy(e->small_ev[k]);
    if (e->d_strings) (void)hipFree(e->d_strings);
    if (e->d_hgr_dots) (void)hipFree(e->d_hgr_dots);
    if (e->d_dw_pieces) (void)hipFree(e->d_dw_pieces);
    if (e->d_sub) (void)hipFree(e->d_sub);
    if (e->d_left) (void)hipFree(e->d_left);
    if (e->d_right) (void)hipFree(e->d_right);
    free_narrow_tables(&e->nt);
    if (e->d_dwl) (void)hipFree(e->d_dwl);
    if (e->d_left_t) (void)hipFree(e->d_left_t);
    if (e->d_joint_l) (void)hipFree(e->d_joint_l);
    if (e->d_joint_r) (void)hipFree(e->d_joint_r);
    if (e->d_brief) (void)hipFree(e->d_brief);
    if (e->d_right_t) (void)hipFree(e->d_right_t);
    if (e->d_dwr) (void)hipFree(e->d_dwr);
    delete e;
}

int encoder_create(int mode, const uint16_t *d_table, const uint16_t *d_store, const int32_t *dm, int n_streams,
                   Encoder **out)
{
    if ((mode != kHGR && mode != kDHGR) || (!d_table && !dm) || !d_store || n_streams <= 0 || !out)
        return set_error(IIV_ERR_INVALID, "iiv_encoder_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(IIV_ERR_NO_DEVICE, "iiv_encoder_create: no HIP device");
    if (dm) {
        // every diff weight is a sum of <= MASKED_DOTS substitute costs; the kernels pack diff
        // weights and store values into 11-bit fields (iiv_stream.h)
        int mx = 0;
        for (int i = 0; i < 256; i++) {
            if (dm[i] < 0) return set_error(IIV_ERR_INVALID, "iiv_encoder_create: dm[%d] < 0", i);
            mx = dm[i] > mx ? dm[i] : mx;
        }
        if (mx * masked_dots(mode) > kMaxValue)
            return set_error(IIV_ERR_INVALID, "iiv_encoder_create: max(dm) * MASKED_DOTS = %d exceeds %d",
                             mx * masked_dots(mode), kMaxValue);
    }
    Encoder *e = new Encoder();
    e->mode = mode;
    e->n_streams = n_streams;
    e->d_table = d_table;
    e->d_store = d_store;
    e->d_left = e->d_right = nullptr;
    e->nt.base = nullptr;
    e->nt.exact = 0;
    e->d_dwl = e->d_dwr = nullptr;
    e->d_left_t = e->d_right_t = nullptr;
    e->d_joint_l = e->d_joint_r = nullptr;
    e->d_brief = nullptr;
    e->d_states = e->d_snapshot[0] = e->d_snapshot[1] = nullptr;
    e->h_live[0] = e->h_live[1] = e->d_live[0] = e->d_live[1] = e->live_now = nullptr;
    e->h_small[0] = nullptr;
    for (int k = 0; k < kSmallSlots; k++) e->small_ev[k] = nullptr;
    e->small_slot = 0;
    e->live_tag = 0;
    e->d_strings = nullptr;
    e->d_hgr_dots = nullptr;
    e->d_dw_pieces = nullptr;
    e->d_sub = nullptr;
    e->dw_mode = dm ? IIV_DW_RECURRENCE : IIV_DW_TABLE;
    e->greedy_mode = IIV_GREEDY_AUTO;
    e->partial_sort = 1;
    e->greedy_lds_pad = 0;
    e->content_choice = IIV_CONTENT_TARGET;
    e->fourth_offset = 0;
    e->gens.assign(1, GenState{0, 0, 0});
    e->h_segs[0] = e->h_segs[1] = nullptr;
    e->seg_ev[0] = e->seg_ev[1] = nullptr;
    e->h_cap[0] = e->h_cap[1] = 0;
    e->seg_slot = 0;
    e->d_segs = nullptr;
    e->d_queue = nullptr;
    e->queue_cap = 0;
    e->d_tie_stats = e->h_tie_stats = nullptr;
    e->tie_seen[0] = e->tie_seen[1] = e->tie_seen[2] = 0;
    e->ops_per_launch = 0.0;
    e->tie_ev = nullptr;
    e->tie_copy_pending = false;
    e->tie_heavy = -1;
    e->tie_rate = 0.0;
    e->d_cap = 0;
    e->d_result = nullptr;
    e->d_packed = nullptr;
    e->profiling = 0;
    e->ms[0] = e->ms[1] = 0;
    e->launches[0] = e->launches[1] = 0;
    e->form_launches[0] = e->form_launches[1] = e->form_launches[2] = e->form_launches[3] = 0;
    e->d_cost = nullptr;
    e->d_perm = nullptr;
    e->order_countdown = 2;     // (the first launches have no history: two of them, then the first sort)
    e->order_streams = 1;
    uint32_t *d_rng0 = nullptr;
    int rc = IIV_OK;
    do {
        if ((rc = hip_check(hipMalloc(&e->d_states, sizeof(StreamState) * (size_t)n_streams), "hipMalloc(stream states)"))) break;
        if ((rc = hip_check(hipMalloc(&e->d_result, 2 * sizeof(int)), "hipMalloc(result)"))) break;
        if (n_streams >= kOrderMinStreams) {
            if ((rc = hip_check(hipMalloc(&e->d_cost, sizeof(uint32_t) * (size_t)n_streams), "hipMalloc(stream costs)"))) break;
            if ((rc = hip_check(hipMemset(e->d_cost, 0, sizeof(uint32_t) * (size_t)n_streams), "memset"))) break;
            if ((rc = hip_check(hipMalloc(&e->d_perm, sizeof(int) * (size_t)n_streams), "hipMalloc(stream order)"))) break;
        }
        if ((rc = hip_check(hipMalloc(&e->d_packed, 4096 * 8), "hipMalloc(packed)"))) break;
        // (the brief's staging struct: here, not on first use -- an allocation inside the asynchronous entry point would synchronise)
        if ((rc = hip_check(hipMalloc(&e->d_brief, sizeof(iiv_video_brief)), "hipMalloc(brief)"))) break;
        if ((rc = hip_check(hipEventCreateWithFlags(&e->seg_ev[0], hipEventDisableTiming), "event"))) break;
        if ((rc = hip_check(hipEventCreateWithFlags(&e->seg_ev[1], hipEventDisableTiming), "event"))) break;
        if ((rc = hip_check(hipMalloc(&e->d_tie_stats, 3 * sizeof(unsigned long long)), "hipMalloc(tie statistics)"))) break;
        if ((rc = hip_check(hipMemset(e->d_tie_stats, 0, 3 * sizeof(unsigned long long)), "memset"))) break;
        if ((rc = hip_check(hipHostMalloc(&e->h_tie_stats, 3 * sizeof(unsigned long long)), "hipHostMalloc(tie statistics)"))) break;
        e->h_tie_stats[0] = e->h_tie_stats[1] = e->h_tie_stats[2] = 0;
        if ((rc = hip_check(hipEventCreateWithFlags(&e->tie_ev, hipEventDisableTiming), "event"))) break;
        // the table values must fit the 11-bit fields too (a caller-made table may not come from dm)
        uint32_t h_max = 0;
        if ((rc = hip_check(hipMemset(e->d_result, 0, 8), "memset"))) break;
        const size_t n_store = (size_t)num_offsets(mode) << (content_bits(mode) + masked_bits(mode));
        hipLaunchKernelGGL(max_u16_kernel, dim3(1024), dim3(256), 0, 0, d_store, n_store, (uint32_t *)e->d_result);
        if (!dm && d_table)
            hipLaunchKernelGGL(max_u16_kernel, dim3(4096), dim3(256), 0, 0, d_table,
                               (size_t)num_offsets(mode) << (2 * masked_bits(mode)), (uint32_t *)e->d_result);
        if ((rc = hip_check(hipMemcpy(&h_max, e->d_result, 4, hipMemcpyDeviceToHost), "read max"))) break;
        if (h_max > (uint32_t)kMaxValue) {
            rc = set_error(IIV_ERR_INVALID, "iiv_encoder_create: table value %u exceeds %d", h_max, kMaxValue);
            break;
        }
        if (dm) {
            if ((rc = build_strings(mode, dm, &e->d_strings, &e->d_sub, 0))) break;
            if (mode == kHGR && (rc = build_hgr_dot_lut(&e->d_hgr_dots, 0))) break;
            if ((rc = build_dw_piece_table(mode, e->d_sub, &e->d_dw_pieces, 0))) break;
            if ((rc = hip_check(hipMalloc(&e->d_left, split_entries(mode, 0) * 4), "hipMalloc(split left)"))) break;
            if ((rc = hip_check(hipMalloc(&e->d_right, split_entries(mode, 1) * 4), "hipMalloc(split right)"))) break;
            if ((rc = build_split_tables(mode, e->d_strings, e->d_sub, e->d_left, e->d_right, 0))) break;
            // (the folded narrow form is compared with the caller's store table entry by entry: e->nt.exact)
            if ((rc = build_narrow_tables(mode, e->d_strings, e->d_sub, e->d_left, d_store, &e->nt, 0))) break;
        }
        // Video.__init__ (video.py:21-62); RNG streams default to random.seed(0) / np.random.seed(0)
        uint32_t rng0[1248], key0 = 0;
        seed_by_array(rng0, &key0, 1);
        seed_genrand(rng0 + 624, 0);
        if ((rc = hip_check(hipMalloc(&d_rng0, sizeof(rng0)), "hipMalloc(rng0)"))) break;
        if ((rc = hip_check(hipMemcpy(d_rng0, rng0, sizeof(rng0), hipMemcpyHostToDevice), "copy rng0"))) break;
        if (e->d_perm) hipLaunchKernelGGL(order_streams_kernel, dim3(1), dim3(1024), 0, 0, e->d_cost, n_streams, e->d_perm, 1);
        hipLaunchKernelGGL(init_states_kernel, dim3(n_streams), dim3(256), 0, 0, e->d_states, d_rng0);
        if ((rc = hip_check(hipGetLastError(), "init_states launch"))) break;
        rc = hip_check(hipDeviceSynchronize(), "init sync");
    } while (0);
    if (d_rng0) (void)hipFree(d_rng0);
    if (rc) {
        encoder_destroy(e);
        return rc;
    }
    *out = e;
    return IIV_OK;
}

int encoder_snapshot(Encoder *e, int slot, hipStream_t st)
{
    if (!e || slot < 0 || slot > 1) return set_error(IIV_ERR_INVALID, "snapshot: null encoder or slot not 0 / 1");
    const size_t bytes = sizeof(StreamState) * (size_t)e->n_streams;
    if (!e->d_snapshot[slot]) IIV_HIP(hipMalloc(&e->d_snapshot[slot], bytes));
    IIV_HIP(hipMemcpyAsync(e->d_snapshot[slot], e->d_states, bytes, hipMemcpyDeviceToDevice, st));
    e->snap_gens[slot] = e->gens;
    return IIV_OK;
}

int encoder_rollback(Encoder *e, int slot, hipStream_t st)
{
    if (!e || slot < 0 || slot > 1 || !e->d_snapshot[slot]) return set_error(IIV_ERR_INVALID, "rollback: no snapshot in that slot");
    const size_t bytes = sizeof(StreamState) * (size_t)e->n_streams;
    IIV_HIP(hipMemcpyAsync(e->d_states, e->d_snapshot[slot], bytes, hipMemcpyDeviceToDevice, st));
    e->gens = e->snap_gens[slot];
    return IIV_OK;
}

int encoder_set_option(Encoder *e, int option, int value)
{
    if (!e) return set_error(IIV_ERR_INVALID, "set_option: null encoder");
    if (option == IIV_OPT_DIFF_WEIGHTS) {
        if (value == IIV_DW_TABLE && !e->d_table) return set_error(IIV_ERR_INVALID, "no table was given at creation");
        if ((value == IIV_DW_RECURRENCE || value == IIV_DW_SPLIT) && !e->d_strings)
            return set_error(IIV_ERR_INVALID, "no diff matrix was given at creation");
        if (value != IIV_DW_TABLE && value != IIV_DW_RECURRENCE && value != IIV_DW_SPLIT)
            return set_error(IIV_ERR_INVALID, "bad value");
        if (value == IIV_DW_SPLIT && !e->d_dwl) {  // built on first use: 4-5 MiB, a few hundred microseconds
            IIV_HIP(hipMalloc(&e->d_dwl, split_dw_entries(e->mode, 0) * 4));
            IIV_HIP(hipMalloc(&e->d_dwr, split_dw_entries(e->mode, 1) * 4));
            int rc = build_split_dw_tables(e->mode, e->d_strings, e->d_sub, e->d_dwl, e->d_dwr, 0);
            if (rc) return rc;
            IIV_HIP(hipDeviceSynchronize());
        }
        e->dw_mode = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_PREFIX_SORT) {
        e->partial_sort = value ? 1 : 0;
        return IIV_OK;
    }
    if (option == IIV_OPT_GREEDY_KERNEL) {
        if (value != IIV_GREEDY_WAVE && value != IIV_GREEDY_WORKGROUP && value != IIV_GREEDY_AUTO && value != IIV_GREEDY_TEAM &&
            value != IIV_GREEDY_WAVE_SHARED && value != IIV_GREEDY_WAVE_PLAIN)
            return set_error(IIV_ERR_INVALID, "bad value");
        if ((value == IIV_GREEDY_WAVE || value == IIV_GREEDY_TEAM || value == IIV_GREEDY_WAVE_SHARED || value == IIV_GREEDY_WAVE_PLAIN) && !e->d_left)
            return set_error(IIV_ERR_INVALID, "the one-wave kernel reads the split store table, which is built from dm "
                                               "(none was given at creation)");
        if (value != IIV_GREEDY_WORKGROUP && value != IIV_GREEDY_AUTO && !e->nt.exact)
            return set_error(IIV_ERR_INVALID, "the split store table built from dm does not reproduce the store table given at creation: "
                                               "this encoder runs the dense-table workgroup kernel only");
        e->greedy_mode = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_GREEDY_LDS_PAD) {
        if (value < 0 || value > 32768) return set_error(IIV_ERR_INVALID, "bad value");
        e->greedy_lds_pad = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_CONTENT_CHOICE) {
        if (value != IIV_CONTENT_TARGET && value != IIV_CONTENT_JOINT && value != IIV_CONTENT_JOINT_SPLIT)
            return set_error(IIV_ERR_INVALID, "bad value");
        if (value != IIV_CONTENT_TARGET && !e->d_left)
            return set_error(IIV_ERR_INVALID, "the joint content choice reads the split store table, which is built "
                                               "from dm (none was given at creation)");
        // (a store table that the narrow form does not reproduce -- not the one dm yields -- leaves only the split-table form)
        if (value == IIV_CONTENT_JOINT && !e->nt.exact) value = IIV_CONTENT_JOINT_SPLIT;
        if (value == IIV_CONTENT_JOINT && !e->d_joint_l) {
            int rc = build_joint_tables(e->mode, e->nt, &e->d_joint_l, &e->d_joint_r, 0);
            if (rc) return rc;
            IIV_HIP(hipDeviceSynchronize());
        }
        if (value == IIV_CONTENT_JOINT_SPLIT && !e->d_left_t) {
            IIV_HIP(hipMalloc(&e->d_left_t, split_entries(e->mode, 0) * 4));
            IIV_HIP(hipMalloc(&e->d_right_t, split_entries(e->mode, 1) * 4));
            int rc = transpose_split_tables(e->mode, e->d_left, e->d_right, e->d_left_t, e->d_right_t, 0);
            if (rc) return rc;
            IIV_HIP(hipDeviceSynchronize());
        }
        e->content_choice = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_FOURTH_OFFSET) {
        if (value != 0 && value != 1) return set_error(IIV_ERR_INVALID, "bad value");
        if (value && (!e->d_left || !e->nt.exact))
            return set_error(IIV_ERR_INVALID, "the fourth offset runs in the one-wave kernel, which reads the split store table "
                                               "built from dm (none was given at creation, or it does not reproduce the store table given)");
        e->fourth_offset = value;
        return IIV_OK;
    }
    if (option == IIV_OPT_STREAM_ORDER) {
        if (value != 0 && value != 1) return set_error(IIV_ERR_INVALID, "bad value");
        e->order_streams = value;
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
        int rc = pack(e->mode, 1, base + offsetof(StreamState, mem[0]), base + offsetof(StreamState, mem[1]), e->d_packed, 0);
        if (!rc) rc = hip_check(hipMemcpy(buf, e->d_packed, 4096 * 8, hipMemcpyDeviceToHost), "copy packed");
        return rc;
    }
    size_t off, want;
    bool writable;
    if (state_item(e->mode, what, off, want, writable)) return set_error(IIV_ERR_INVALID, "get_state: unknown item %d", what);
    if (bytes != want) return set_error(IIV_ERR_INVALID, "get_state: item %d is %zu bytes, got %zu", what, want, bytes);
    if (what == IIV_STATE_UP_MAIN || what == IIV_STATE_UP_AUX)
        if (int rc = materialise_up(e, s, 1)) return rc;
    IIV_HIP(hipMemcpy(buf, base + off, bytes, hipMemcpyDeviceToHost));
    return IIV_OK;
}

// n consecutive streams starting at s0; buf = n items back to back
int encoder_set_state(Encoder *e, int s0, int n, int what, const void *buf, size_t bytes)
{
    if (!e || !buf || n <= 0 || s0 < 0 || s0 + n > e->n_streams) return set_error(IIV_ERR_INVALID, "set_state: bad argument");
    IIV_HIP(hipDeviceSynchronize());
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s0);
    const size_t pitch = sizeof(StreamState);
    if (what == IIV_STATE_RNG_PY || what == IIV_STATE_RNG_NP) {
        if (bytes != 625 * 4) return set_error(IIV_ERR_INVALID, "set_state: RNG state is 625 u32");
        for (int i = 0; i < n; i++) {
            uint32_t idx = ((const uint32_t *)buf)[(size_t)i * 625 + 624];
            if (idx > 624) return set_error(IIV_ERR_INVALID, "set_state: RNG index %u > 624 (stream %d)", idx, s0 + i);
        }
        size_t o_mt = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py) : offsetof(StreamState, mt_np);
        size_t o_ix = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py_idx) : offsetof(StreamState, mt_np_idx);
        IIV_HIP(hipMemcpy2D(base + o_mt, pitch, buf, 625 * 4, 624 * 4, (size_t)n, hipMemcpyHostToDevice));
        IIV_HIP(hipMemcpy2D(base + o_ix, pitch, (const uint8_t *)buf + 624 * 4, 625 * 4, 4, (size_t)n, hipMemcpyHostToDevice));
        return IIV_OK;
    }
    size_t off, want;
    bool writable;
    if (state_item(e->mode, what, off, want, writable) || !writable)
        return set_error(IIV_ERR_INVALID, "set_state: item %d is not settable", what);
    if (bytes != want) return set_error(IIV_ERR_INVALID, "set_state: item %d is %zu bytes, got %zu", what, want, bytes);
    if (what == IIV_STATE_UP_MAIN || what == IIV_STATE_UP_AUX)   // (compact_up_kernel rewrites BOTH banks' 16-bit copies from up[])
        if (int rc = materialise_up(e, s0, n)) return rc;
    IIV_HIP(hipMemcpy2D(base + off, pitch, buf, bytes, bytes, (size_t)n, hipMemcpyHostToDevice));
    if (what == IIV_STATE_UP_MAIN || what == IIV_STATE_UP_AUX) {
        hipLaunchKernelGGL(compact_up_kernel, dim3(n), dim3(256), 0, 0, e->d_states + s0);
        if (int rc = hip_check(hipGetLastError(), "compact_up_kernel launch")) return rc;
        return hip_check(hipDeviceSynchronize(), "compact_up sync");
    }
    return IIV_OK;
}

// The small items of ONE stream, enqueued on `st` behind the launches already there -- no device-wide synchronisation, no
// blocking copy: the bytes are taken into a pinned slot before the call returns.
int encoder_set_state_async(Encoder *e, int s, int what, const void *buf, size_t bytes, hipStream_t st)
{
    if (!e || !buf || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "set_state_async: bad argument");
    if (what != IIV_STATE_OUT_OF_WORK && what != IIV_STATE_RNG_PY && what != IIV_STATE_RNG_NP)
        return set_error(IIV_ERR_INVALID, "set_state_async: item %d (only OUT_OF_WORK, RNG_PY, RNG_NP)", what);
    const size_t want = what == IIV_STATE_OUT_OF_WORK ? 8 : 625 * 4;
    if (bytes != want) return set_error(IIV_ERR_INVALID, "set_state_async: item %d is %zu bytes, got %zu", what, want, bytes);
    if (what != IIV_STATE_OUT_OF_WORK && ((const uint32_t *)buf)[624] > 624)
        return set_error(IIV_ERR_INVALID, "set_state_async: RNG index %u > 624", ((const uint32_t *)buf)[624]);
    if (!e->h_small[0]) {
        void *p = nullptr;
        IIV_HIP(hipHostMalloc(&p, kSmallSlots * kSmallBytes, hipHostMallocDefault));
        e->h_small[0] = static_cast<uint8_t *>(p);
        for (int k = 0; k < kSmallSlots; k++) IIV_HIP(hipEventCreateWithFlags(&e->small_ev[k], hipEventDisableTiming));
        e->small_slot = -kSmallSlots;   // (the first round of the ring has nothing to wait for)
    }
    const int k = e->small_slot < 0 ? e->small_slot + kSmallSlots : e->small_slot;
    if (e->small_slot >= 0) IIV_HIP(hipEventSynchronize(e->small_ev[k]));   // the copies that last used this slot are done
    e->small_slot = e->small_slot < 0 ? e->small_slot + 1 : (e->small_slot + 1) % kSmallSlots;
    uint8_t *slot = e->h_small[0] + (size_t)k * kSmallBytes;
    memcpy(slot, buf, bytes);
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s);
    if (what == IIV_STATE_OUT_OF_WORK) {
        IIV_HIP(hipMemcpyAsync(base + offsetof(StreamState, out_of_work), slot, 8, hipMemcpyHostToDevice, st));
    } else {
        const size_t o_mt = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py) : offsetof(StreamState, mt_np);
        const size_t o_ix = what == IIV_STATE_RNG_PY ? offsetof(StreamState, mt_py_idx) : offsetof(StreamState, mt_np_idx);
        IIV_HIP(hipMemcpyAsync(base + o_mt, slot, 624 * 4, hipMemcpyHostToDevice, st));
        IIV_HIP(hipMemcpyAsync(base + o_ix, slot + 624 * 4, 4, hipMemcpyHostToDevice, st));
    }
    IIV_HIP(hipEventRecord(e->small_ev[k], st));
    return IIV_OK;
}

static_assert(offsetof(StreamState, up) == 2 * 8192 && offsetof(StreamState, mt_np) == offsetof(StreamState, mt_py) + 2496 &&
                  offsetof(StreamState, mt_py_idx) == offsetof(StreamState, mt_np) + 2496 &&
                  offsetof(StreamState, mt_np_idx) == offsetof(StreamState, mt_py_idx) + 4,
              "iiv_video_state is copied in blocks that follow StreamState's layout");

int encoder_get_video_state(Encoder *e, int s, iiv_video_state *out)
{
    if (!e || !out || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "get_video_state: bad argument");
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s);
    IIV_HIP(hipDeviceSynchronize());
    int rc = pack(e->mode, 1, base + offsetof(StreamState, mem[0]), base + offsetof(StreamState, mem[1]), e->d_packed, 0);
    if (!rc) rc = materialise_up(e, s, 1);
    if (rc) return rc;
    IIV_HIP(hipMemcpy(out->mem_main, base, 2 * 8192 + 2 * 8192 * 4, hipMemcpyDeviceToHost));   // mem[2] | up[2]
    uint32_t rng[1250];
    IIV_HIP(hipMemcpy(rng, base + offsetof(StreamState, mt_py), sizeof(rng), hipMemcpyDeviceToHost));
    memcpy(out->rng_py, rng, 2496);
    memcpy(out->rng_np, rng + 624, 2496);
    out->rng_py[624] = rng[1248];
    out->rng_np[624] = rng[1249];
    IIV_HIP(hipMemcpy(out->out_of_work, base + offsetof(StreamState, out_of_work), 8, hipMemcpyDeviceToHost));
    IIV_HIP(hipMemcpy(out->packed, e->d_packed, 4096 * 8, hipMemcpyDeviceToHost));
    return IIV_OK;
}

// iiv_video_brief of one stream, assembled on the device so that it travels in one copy: priority sums, hole bytes,
// out_of_work and both MT19937 states in the struct's own layout
__global__ __launch_bounds__(256) void brief_kernel(const StreamState *__restrict__ S, iiv_video_brief *__restrict__ out)
{
    __shared__ long long sums[2][4];
    __shared__ int holes[2][4];
    const int tid = threadIdx.x;
    long long a[2] = {0, 0};
    int h[2] = {0, 0};
    for (int b = 0; b < 2; b++)
        for (int i = tid; i < 8192; i += 256) {
            a[b] += up_value(*S, b, i);
            if (is_hole(i & 255) && S->mem[b][i] != 0) h[b]++;
        }
    for (int i = tid; i < 624; i += 256) {
        out->rng_py[i] = S->mt_py[i];
        out->rng_np[i] = S->mt_np[i];
    }
    for (int b = 0; b < 2; b++) {
        for (int d = 1; d < 64; d <<= 1) {
            a[b] += __shfl_xor(a[b], d, 64);
            h[b] += __shfl_xor(h[b], d, 64);
        }
        if ((tid & 63) == 0) {
            sums[b][tid >> 6] = a[b];
            holes[b][tid >> 6] = h[b];
        }
    }
    __syncthreads();
    if (tid < 2) {
        out->priority_sum[tid] = sums[tid][0] + sums[tid][1] + sums[tid][2] + sums[tid][3];
        out->hole_bytes[tid] = holes[tid][0] + holes[tid][1] + holes[tid][2] + holes[tid][3];
        out->out_of_work[tid] = S->out_of_work[tid];
        (tid ? out->rng_np : out->rng_py)[624] = tid ? S->mt_np_idx : S->mt_py_idx;
    }
}

int encoder_get_video_brief(Encoder *e, int s, iiv_video_brief *out)
{
    if (!e || !out || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "get_video_brief: bad argument");
    IIV_HIP(hipDeviceSynchronize());
    if (!e->d_brief) IIV_HIP(hipMalloc(&e->d_brief, sizeof(iiv_video_brief)));
    hipLaunchKernelGGL(brief_kernel, dim3(1), dim3(256), 0, 0, e->d_states + s, (iiv_video_brief *)e->d_brief);
    IIV_HIP(hipGetLastError());
    IIV_HIP(hipMemcpy(out, e->d_brief, sizeof(iiv_video_brief), hipMemcpyDeviceToHost));
    return IIV_OK;
}

// The same, enqueued on `st` behind whatever was launched there: nothing waits.  host_out holds the brief once the
// stream has been synchronised (iiv_encoder_check does); pinned memory makes the copy truly asynchronous.
int encoder_get_video_brief_async(Encoder *e, int s, iiv_video_brief *out, hipStream_t st)
{
    if (!e || !out || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "get_video_brief_async: bad argument");
    if (!e->d_brief) IIV_HIP(hipMalloc(&e->d_brief, sizeof(iiv_video_brief)));
    hipLaunchKernelGGL(brief_kernel, dim3(1), dim3(256), 0, st, e->d_states + s, (iiv_video_brief *)e->d_brief);
    IIV_HIP(hipGetLastError());
    IIV_HIP(hipMemcpyAsync(out, e->d_brief, sizeof(iiv_video_brief), hipMemcpyDeviceToHost, st));
    return IIV_OK;
}

int encoder_set_video_state(Encoder *e, int s, const iiv_video_state *in)
{
    if (!e || !in || s < 0 || s >= e->n_streams) return set_error(IIV_ERR_INVALID, "set_video_state: bad argument");
    if (in->rng_py[624] > 624 || in->rng_np[624] > 624) return set_error(IIV_ERR_INVALID, "set_video_state: RNG index > 624");
    uint8_t *base = reinterpret_cast<uint8_t *>(e->d_states + s);
    IIV_HIP(hipDeviceSynchronize());
    IIV_HIP(hipMemcpy(base, in->mem_main, 2 * 8192 + 2 * 8192 * 4, hipMemcpyHostToDevice));
    uint32_t rng[1250];
    memcpy(rng, in->rng_py, 2496);
    memcpy(rng + 624, in->rng_np, 2496);
    rng[1248] = in->rng_py[624];
    rng[1249] = in->rng_np[624];
    IIV_HIP(hipMemcpy(base + offsetof(StreamState, mt_py), rng, sizeof(rng), hipMemcpyHostToDevice));
    IIV_HIP(hipMemcpy(base + offsetof(StreamState, out_of_work), in->out_of_work, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(compact_up_kernel, dim3(1), dim3(256), 0, 0, e->d_states + s);   // (both banks' priorities were just written)
    if (int rc = hip_check(hipGetLastError(), "compact_up_kernel launch")) return rc;
    return hip_check(hipDeviceSynchronize(), "compact_up sync");
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
        e->form_launches[0] = e->form_launches[1] = e->form_launches[2] = e->form_launches[3] = 0;
    }
    return IIV_OK;
}

int encoder_launch_forms(Encoder *e, int64_t counts[4])
{
    if (!e || !counts) return set_error(IIV_ERR_INVALID, "launch_forms: bad argument");
    for (int k = 0; k < 4; k++) counts[k] = e->form_launches[k];
    return IIV_OK;
}

// do the kernels' tie statistics decide anything for this encoder?  (only a batch that fills the GPU with the one-wave
// kernel in its automatic form choice: a single-clip Video, a forced form, the team and the workgroup kernel ignore them,
// and their launches should not pay for counters, a reduction launch and a copy per call)
static bool tie_stats_wanted(const Encoder *e)
{
    return e->d_left && e->nt.exact && (e->greedy_mode == IIV_GREEDY_AUTO || e->greedy_mode == IIV_GREEDY_WAVE) &&
           e->content_choice == IIV_CONTENT_TARGET &&
           e->n_streams >= (e->mode == kHGR ? kSharedHgrMinStreams : kSharedDhgrMinStreams);
}

// the form of the one-wave kernel the next launch of a full batch runs (launch_round)
static bool shared_form_now(const Encoder *e)
{
    return e->greedy_mode == IIV_GREEDY_WAVE_SHARED ||
           (e->greedy_mode != IIV_GREEDY_WAVE_PLAIN &&
            e->n_streams >= (e->mode == kHGR ? kSharedHgrMinStreams : kSharedDhgrMinStreams) &&
            (e->tie_heavy < 0 ? e->mode == kHGR : e->tie_heavy == 0));
}

int encoder_input_stats(Encoder *e, double stats[2], int *form)
{
    if (!e) return set_error(IIV_ERR_INVALID, "input_stats: null encoder");
    if (stats) {
        stats[0] = e->tie_rate;
        stats[1] = e->ops_per_launch;
    }
    if (form) *form = shared_form_now(e) ? IIV_GREEDY_WAVE_SHARED : IIV_GREEDY_WAVE_PLAIN;
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

// One stream's segment list -> one LaunchSeg per round (a segment that emits nothing takes no
// round).  `rounds` receives the descriptors; gs is the stream's generator bookkeeping.
static int plan_stream(Encoder *e, int stream, const iiv_segment *segs, int n_segs, int n_frames, GenState &gs,
                       std::vector<LaunchSeg> &rounds, size_t &total_ops)
{
    rounds.clear();
    size_t done = 0;
    for (int i = 0; i < n_segs; i++) {
        const iiv_segment &g = segs[i];
        if (g.n_ops < 0 || g.frame < 0 || g.frame >= n_frames || (g.is_aux != 0 && g.is_aux != 1) ||
            (e->mode == kHGR && g.is_aux))
            return set_error(IIV_ERR_INVALID, "iiv_encode: bad segment %d (stream %d)", i, stream);
        if (g.n_ops == 0) {
            // encode_frame() only creates a lazy generator (video.py:72-93): remember
            // it, run nothing.  A later restart == 0 segment will start it.
            if (g.restart) gs = GenState{2, g.is_aux, g.frame};
            continue;
        }
        bool need_prologue = g.restart != 0;
        if (!g.restart) {
            if (!gs.active)
                return set_error(IIV_ERR_INVALID, "iiv_encode: segment %d continues no generator (stream %d)", i, stream);
            if (gs.is_aux != g.is_aux || gs.frame != g.frame)
                return set_error(IIV_ERR_INVALID, "iiv_encode: segment %d continues a different target/bank (stream %d)", i,
                                 stream);
            if (gs.active == 2) need_prologue = true;
        }
        int need = -1;
        if (need_prologue) {
            need = 0;
            if (e->partial_sort) {
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
                // (a step consumes at most one primary and two -- with the fourth offset three -- secondaries resolved to zero)
                const long per_op = e->fourth_offset ? 4 : 3;
                if (closed && per_op * budget <= kSelNeedMax) need = (int)(per_op * budget);
            }
        }
        gs = GenState{1, g.is_aux, g.frame};
        rounds.push_back(LaunchSeg{g.frame, g.is_aux, g.n_ops, (int32_t)done, need});
        done += (size_t)g.n_ops;
    }
    total_ops = done;
    return IIV_OK;
}

// copy n descriptors to the device through the pinned staging ring (truly asynchronous)
static int upload_segs(Encoder *e, const std::vector<LaunchSeg> &host, hipStream_t st)
{
    const size_t n = host.size();
    if (n > e->d_cap) {
        IIV_HIP(hipStreamSynchronize(st));  // kernels of earlier calls may still read the old buffer
        if (e->d_segs) (void)hipFree(e->d_segs);
        e->d_segs = nullptr;
        e->d_cap = 0;
        IIV_HIP(hipMalloc(&e->d_segs, n * sizeof(LaunchSeg) * 2));
        e->d_cap = 2 * n;
    }
    const int k = e->seg_slot;
    e->seg_slot ^= 1;
    if (e->h_cap[k]) IIV_HIP(hipEventSynchronize(e->seg_ev[k]));  // the copy that last used this slot is done
    if (n > e->h_cap[k]) {
        if (e->h_segs[k]) (void)hipHostFree(e->h_segs[k]);
        e->h_segs[k] = nullptr;
        e->h_cap[k] = 0;
        IIV_HIP(hipHostMalloc(&e->h_segs[k], n * sizeof(LaunchSeg) * 2, hipHostMallocDefault));
        e->h_cap[k] = 2 * n;
    }
    memcpy(e->h_segs[k], host.data(), n * sizeof(LaunchSeg));
    IIV_HIP(hipMemcpyAsync(e->d_segs, e->h_segs[k], n * sizeof(LaunchSeg), hipMemcpyHostToDevice, st));
    IIV_HIP(hipEventRecord(e->seg_ev[k], st));
    return IIV_OK;
}

// zeroed stream counters for the n_rounds launches of a call
static int reset_queues(Encoder *e, size_t n_rounds, hipStream_t st)
{
    if (n_rounds > e->queue_cap) {
        IIV_HIP(hipStreamSynchronize(st));  // kernels of earlier calls may still count in the old buffer
        if (e->d_queue) (void)hipFree(e->d_queue);
        e->d_queue = nullptr;
        e->queue_cap = 0;
        IIV_HIP(hipMalloc(&e->d_queue, n_rounds * 2 * sizeof(int)));
        e->queue_cap = n_rounds * 2;
    }
    IIV_HIP(hipMemsetAsync(e->d_queue, 0, n_rounds * sizeof(int), st));
    return IIV_OK;
}

// the streams' stat_* totals summed (one small launch per iiv_encode call, in front of the copy)
__global__ void tie_stats_kernel(const StreamState *states, int n, unsigned long long *out)
{
    unsigned long long a = 0, b = 0, c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        a += states[i].stat_exact;
        b += states[i].stat_ops;
        c += states[i].stat_runs;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
        c += __shfl_xor(c, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], a);
        atomicAdd(&out[1], b);
        atomicAdd(&out[2], c);
    }
}

// tie statistics (kTieHeavyPercentDHGR / HGR): take in what an earlier call's copy has delivered, if it has
static void tie_stats_poll(Encoder *e)
{
    if (!e->tie_copy_pending || hipEventQuery(e->tie_ev) != hipSuccess) return;
    e->tie_copy_pending = false;
    if (e->h_tie_stats[2] < e->tie_seen[2] || e->h_tie_stats[1] < e->tie_seen[1] || e->h_tie_stats[0] < e->tie_seen[0]) {
        for (int k = 0; k < 3; k++) e->tie_seen[k] = e->h_tie_stats[k];   // (a rollback took the streams' totals back)
        return;
    }
    const unsigned long long ties = e->h_tie_stats[0] - e->tie_seen[0], ops = e->h_tie_stats[1] - e->tie_seen[1],
                             runs = e->h_tie_stats[2] - e->tie_seen[2];
    if (runs < 2ull * (unsigned long long)e->n_streams) return;   // (too little to judge by: keep counting)
    for (int k = 0; k < 3; k++) e->tie_seen[k] = e->h_tie_stats[k];
    e->tie_rate = ops ? (double)ties / (double)ops : 0.0;
    e->ops_per_launch = (double)ops / (double)runs;
    // (`tie_heavy` = the plain form is the better one for this input)
    const unsigned heavy = e->mode == kDHGR ? kTieHeavyPercentDHGR : kTieHeavyPercentHGR;
    e->tie_heavy = (ties * 100ull > (unsigned long long)heavy * ops || ops < (unsigned long long)kSharedMinOpsPerLaunch * runs) ? 1 : 0;
}

// ... and ask for the counters as this call leaves them (asynchronous; one copy in flight at a time)
static int tie_stats_request(Encoder *e, hipStream_t st)
{
    if (!e->d_tie_stats || e->tie_copy_pending || !tie_stats_wanted(e)) return IIV_OK;
    IIV_HIP(hipMemsetAsync(e->d_tie_stats, 0, 3 * sizeof(unsigned long long), st));
    const int blocks = e->n_streams < 64 * 256 ? (e->n_streams + 255) / 256 : 64;
    hipLaunchKernelGGL(tie_stats_kernel, dim3(blocks), dim3(256), 0, st, e->d_states, e->n_streams, e->d_tie_stats);
    IIV_HIP(hipGetLastError());
    IIV_HIP(hipMemcpyAsync(e->h_tie_stats, e->d_tie_stats, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    IIV_HIP(hipEventRecord(e->tie_ev, st));
    e->tie_copy_pending = true;
    return IIV_OK;
}

// which greedy kernel an encoder's launches run (launch_round)
static bool wave_kernels_run(const Encoder *e)
{
    return e->content_choice == IIV_CONTENT_TARGET &&
           (e->fourth_offset || (e->d_left && e->nt.exact && e->greedy_mode != IIV_GREEDY_WORKGROUP));
}
static bool team_kernel_runs(const Encoder *e)
{
    return wave_kernels_run(e) && (e->greedy_mode == IIV_GREEDY_TEAM || (e->greedy_mode == IIV_GREEDY_AUTO && e->n_streams <= kTeamMaxStreams));
}

// launch round r: descriptors at d + r * round_stride, stream i reads entry i * seg_stride
static int launch_round(Encoder *e, const uint8_t *d_main, const uint8_t *d_aux, int n_frames, const LaunchSeg *d_round,
                        int seg_stride, bool any_prologue, bool any_greedy, int uniform_bank, int *d_queue, uint8_t *d_ops,
                        size_t ops_stride, hipStream_t st)
{
    size_t slot = 0;
    if (any_prologue) {
        if (e->profiling) { int prc = prof_begin(e, 0, st, slot); if (prc) return prc; }
        const PrologueArgs pa{e->d_states, d_main, d_aux, n_frames, e->n_streams, d_round, seg_stride, e->d_table, e->d_strings,
                              e->d_sub, e->d_dwl, e->d_dwr, e->d_hgr_dots, e->d_dw_pieces};
        int prc2 = launch_prologue(e->mode, e->dw_mode, pa, st);
        if (prc2) return prc2;
        if (e->profiling) { int prc = prof_end(e, slot, st); if (prc) return prc; }
    }
    if (!any_greedy) return IIV_OK;
    if (e->profiling) { int prc = prof_begin(e, 1, st, slot); if (prc) return prc; }
    // the one-wave kernel (split store table, pipelined loads) wins at every batch size: 28
    // streams resident per CU, and 1.16 us per opcode for a single stream against the 1.6 us of
    // the 4-wave workgroup with its two barriers per opcode (tools/single_stream_probe.py); the
    // workgroup kernel remains for encoders created without dm, and as a second implementation
    // (the joint content choice, not a reference behaviour, exists in the workgroup kernel only)
    // (f4's fourth offset: in the plain one-wave kernel only, whatever the kernel option says)
    // (e->nt.exact: the folded narrow form reproduced every entry of the caller's store table at creation -- always so for
    // tables built from the same dm; otherwise the dense-table workgroup kernel runs)
    // (round 6: together with the joint content choice the fourth offset runs in the workgroup kernel, the joint choice's home)
    const bool use_wave = wave_kernels_run(e);
    // few streams: a team of eight waves per stream scores the next entries of the list
    // concurrently (iiv_team.hip); from ~900 streams on, one wave per stream fills the GPU
    const bool use_team = team_kernel_runs(e);
    if (use_wave) {
        // longest first: every kOrderEvery launches the streams are sorted by what their latest launch cost them (d_perm holds
        // the identity until the first sort)
        const bool ordered = e->d_perm && e->order_streams && !use_team;
        if (ordered && --e->order_countdown <= 0) {
            hipLaunchKernelGGL(order_streams_kernel, dim3(1), dim3(1024), 0, st, e->d_cost, e->n_streams, e->d_perm, 0);
            int orc = hip_check(hipGetLastError(), "order_streams_kernel launch");
            if (orc) return orc;
            e->order_countdown = kOrderEvery;
        }
        GreedyArgs a{e->d_states, d_main, d_aux, n_frames, e->n_streams, d_round, seg_stride, e->d_left, e->d_right, e->nt, d_ops,
                     ops_stride, e->greedy_lds_pad,
                     e->greedy_mode == IIV_GREEDY_WAVE_PLAIN ? -1 : uniform_bank,
                     // the LDS-shared form: on request; otherwise for batches that fill the GPU with its workgroups, unless the
                     // kernels have reported input on which the plain form is the faster one (kTieHeavyPercentDHGR / HGR; until they
                     // have reported: HGR shared, DHGR plain -- the better guess for each)
                     shared_form_now(e),
                     d_queue, e->fourth_offset != 0, e->d_tie_stats != nullptr && tie_stats_wanted(e),
                     ordered ? e->d_perm : (const int *)nullptr, ordered ? e->d_cost : (uint32_t *)nullptr};
        int form = 0;   // what launch_greedy_wave ran: 0 plain, 1 LDS-shared (it needs one bank per round and the stream counter)
        if (e->live_now) {
            if (!use_team) return set_error(IIV_ERR_INVALID, "iiv_encode_live: this encoder's launches do not run the team kernel");
            a.live = e->live_now;
            a.live_tag = e->live_tag;
        }
        int rc = use_team ? launch_greedy_team(e->mode, a, st) : launch_greedy_wave(e->mode, a, st, &form);
        if (rc) return rc;
        if (e->profiling) e->form_launches[use_team ? 2 : form]++;
    } else {
        if (e->live_now) return set_error(IIV_ERR_INVALID, "iiv_encode_live: this encoder's launches do not run the team kernel");
        const bool packed = e->content_choice == IIV_CONTENT_JOINT;
        const WorkgroupArgs wa{e->d_states, d_main, d_aux, n_frames, e->n_streams, d_round, seg_stride, e->d_store,
                               packed ? e->d_joint_l : e->d_left_t, packed ? e->d_joint_r : e->d_right_t, d_ops, ops_stride};
        int wrc = launch_greedy_workgroup(e->mode, packed ? 2 : e->content_choice == IIV_CONTENT_JOINT_SPLIT ? 1 : 0, e->fourth_offset != 0, wa, st);
        if (wrc) return wrc;
        if (e->profiling) e->form_launches[3]++;
    }
    if (e->profiling) { int prc = prof_end(e, slot, st); if (prc) return prc; }
    return IIV_OK;
}

// every stream runs the same segment list
int encode(Encoder *e, const uint8_t *d_main, const uint8_t *d_aux, int n_frames, const iiv_segment *segs, int n_segs,
           uint8_t *d_ops, hipStream_t st)
{
    if (!e || !d_main || !segs || n_segs < 0 || n_frames <= 0 || (!d_ops && n_segs > 0))
        return set_error(IIV_ERR_INVALID, "iiv_encode: bad argument");
    if (e->mode == kDHGR && !d_aux) return set_error(IIV_ERR_INVALID, "iiv_encode: DHGR needs aux frames");
    if (e->gens.size() != 1)
        return set_error(IIV_ERR_INVALID, "iiv_encode: the streams have run different schedules (iiv_encode_streams); "
                                          "keep using iiv_encode_streams");
    GenState gs = e->gens[0];
    std::vector<LaunchSeg> rounds;
    size_t total = 0;
    int rc = plan_stream(e, 0, segs, n_segs, n_frames, gs, rounds, total);
    if (rc) return rc;
    e->gens[0] = gs;
    if (rounds.empty()) return IIV_OK;
    if ((rc = upload_segs(e, rounds, st))) return rc;
    // (the stream counters are the LDS-shared form's: the team kernel's launches -- few streams, the drop-in Video's path,
    // where a 4 us memset in front of every generator is felt -- do without)
    if (!team_kernel_runs(e) && (rc = reset_queues(e, rounds.size(), st))) return rc;
    tie_stats_poll(e);
    for (size_t r = 0; r < rounds.size(); r++)
        if ((rc = launch_round(e, d_main, d_aux, n_frames, e->d_segs + r, 0, rounds[r].need >= 0, true, rounds[r].is_aux, e->d_queue ? e->d_queue + r : nullptr, d_ops,
                               total * 6, st)))
            return rc;
    return tie_stats_request(e, st);
}

// ---- live hand-over (include/iivision.h: iiv_encode_live)
constexpr int kLiveCap = 4096;   // opcodes per queue: what one call may emit

int encoder_live_queue(Encoder *e, int slot, uint64_t **host_queue, int *capacity)
{
    if (!e || slot < 0 || slot > 1 || !host_queue || !capacity) return set_error(IIV_ERR_INVALID, "live_queue: bad argument");
    if (e->n_streams != 1) return set_error(IIV_ERR_INVALID, "live_queue: a one-stream encoder's (this one has %d)", e->n_streams);
    if (!e->h_live[slot]) {
        // coherent (fine-grained) and mapped: a kernel's store is a write into host memory, seen by the host without any
        // synchronisation call
        void *p = nullptr;
        IIV_HIP(hipHostMalloc(&p, (size_t)kLiveCap * 8, hipHostMallocCoherent | hipHostMallocMapped));
        memset(p, 0, (size_t)kLiveCap * 8);   // (tag 0 is never used)
        void *dev = nullptr;
        IIV_HIP(hipHostGetDevicePointer(&dev, p, 0));
        e->h_live[slot] = static_cast<unsigned long long *>(p);
        e->d_live[slot] = static_cast<unsigned long long *>(dev);
    }
    *host_queue = reinterpret_cast<uint64_t *>(e->h_live[slot]);
    *capacity = kLiveCap;
    return IIV_OK;
}

int encode_live(Encoder *e, const uint8_t *d_main, const uint8_t *d_aux, int n_frames, const iiv_segment *segs, int n_segs,
                uint8_t *d_ops, int slot, uint32_t tag, hipStream_t st)
{
    if (!e || slot < 0 || slot > 1 || !segs) return set_error(IIV_ERR_INVALID, "iiv_encode_live: bad argument");
    if (!e->h_live[slot]) return set_error(IIV_ERR_INVALID, "iiv_encode_live: no queue in that slot (iiv_encoder_live_queue)");
    if (tag == 0 || tag > 0xffffu) return set_error(IIV_ERR_INVALID, "iiv_encode_live: the tag must be 1 .. 65535");
    if (!team_kernel_runs(e))   // (refused before anything is launched: the caller falls back to iiv_encode)
        return set_error(IIV_ERR_INVALID, "iiv_encode_live: this encoder's launches do not run the team kernel (options)");
    long long total = 0;
    for (int i = 0; i < n_segs; i++) total += segs[i].n_ops > 0 ? segs[i].n_ops : 0;
    if (total > kLiveCap) return set_error(IIV_ERR_INVALID, "iiv_encode_live: %lld opcodes, the queue holds %d", total, kLiveCap);
    e->live_now = e->d_live[slot];
    e->live_tag = tag;
    const int rc = encode(e, d_main, d_aux, n_frames, segs, n_segs, d_ops, st);
    e->live_now = nullptr;
    return rc;
}

// stream s runs segs[seg_begin[s] .. seg_begin[s + 1])
int encode_streams(Encoder *e, const uint8_t *d_main, const uint8_t *d_aux, int n_frames, const iiv_segment *segs,
                   const int32_t *seg_begin, uint8_t *d_ops, size_t ops_stride, hipStream_t st)
{
    if (!e || !d_main || !segs || !seg_begin || n_frames <= 0 || !d_ops)
        return set_error(IIV_ERR_INVALID, "iiv_encode_streams: bad argument");
    if (e->mode == kDHGR && !d_aux) return set_error(IIV_ERR_INVALID, "iiv_encode_streams: DHGR needs aux frames");
    const int S = e->n_streams;
    std::vector<GenState> gens = e->gens;
    if (gens.size() == 1) gens.assign((size_t)S, e->gens[0]);
    std::vector<std::vector<LaunchSeg>> plan((size_t)S);
    size_t n_rounds = 0;
    for (int s = 0; s < S; s++) {
        if (seg_begin[s + 1] < seg_begin[s]) return set_error(IIV_ERR_INVALID, "iiv_encode_streams: seg_begin must ascend");
        size_t total = 0;
        int rc = plan_stream(e, s, segs + seg_begin[s], seg_begin[s + 1] - seg_begin[s], n_frames, gens[(size_t)s],
                             plan[(size_t)s], total);
        if (rc) return rc;
        if (total * 6 > ops_stride)
            return set_error(IIV_ERR_INVALID, "iiv_encode_streams: stream %d emits %zu opcodes, ops_stride holds %zu", s,
                             total, ops_stride / 6);
        n_rounds = plan[(size_t)s].size() > n_rounds ? plan[(size_t)s].size() : n_rounds;
    }
    e->gens = gens;
    if (n_rounds == 0) return IIV_OK;
    std::vector<LaunchSeg> table(n_rounds * (size_t)S, LaunchSeg{0, 0, 0, 0, -1});
    std::vector<char> any_pro(n_rounds, 0);
    std::vector<int> bank(n_rounds, -2);   // -2: no stream emits in this round yet; -1: the streams' banks differ
    for (int s = 0; s < S; s++)
        for (size_t r = 0; r < plan[(size_t)s].size(); r++) {
            const LaunchSeg &g = plan[(size_t)s][r];
            table[r * (size_t)S + (size_t)s] = g;
            if (g.need >= 0) any_pro[r] = 1;
            if (g.n_ops > 0) bank[r] = bank[r] == -2 ? g.is_aux : (bank[r] == g.is_aux ? bank[r] : -1);
        }
    int rc = upload_segs(e, table, st);
    if (rc) return rc;
    if (!team_kernel_runs(e) && (rc = reset_queues(e, n_rounds, st))) return rc;
    tie_stats_poll(e);
    for (size_t r = 0; r < n_rounds; r++)
        if ((rc = launch_round(e, d_main, d_aux, n_frames, e->d_segs + r * (size_t)S, 1, any_pro[r] != 0, true, bank[r] < 0 ? -1 : bank[r],
                               e->d_queue ? e->d_queue + r : nullptr, d_ops,
                               ops_stride, st)))
            return rc;
    return tie_stats_request(e, st);
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
    int *d_res = e->d_result;
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
                                  "internal: prefix sort exhausted before the opcode budget",
                                  "internal: a stream's bank differs from its workgroup's shared table"};
    int is_overflow = code == kErrPushedOverflow;
    return set_error(is_overflow ? IIV_ERR_OVERFLOW : IIV_ERR_ASSERT, "stream %d: %s", first,
                     (code > 0 && code < 9) ? names[code] : "unknown error");
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
    return iiv::encoder_snapshot(enc->impl, 0, (hipStream_t)stream);
}

int iiv_encoder_rollback(iiv_encoder *enc, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_rollback(enc->impl, 0, (hipStream_t)stream);
}

int iiv_encoder_snapshot_slot(iiv_encoder *enc, int slot, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_snapshot(enc->impl, slot, (hipStream_t)stream);
}

int iiv_encoder_rollback_slot(iiv_encoder *enc, int slot, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_rollback(enc->impl, slot, (hipStream_t)stream);
}

int iiv_encoder_set_option(iiv_encoder *enc, int option, int value)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_option(enc->impl, option, value);
}

int iiv_encoder_info(iiv_encoder *enc, int *mode, int *n_streams)
{
    if (!enc || !enc->impl) return iiv::set_error(IIV_ERR_INVALID, "iiv_encoder_info: null encoder");
    if (mode) *mode = enc->impl->mode;
    if (n_streams) *n_streams = enc->impl->n_streams;
    return IIV_OK;
}

int iiv_encoder_get_state(iiv_encoder *enc, int stream_index, int what, void *host_buf, size_t bytes)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_get_state(enc->impl, stream_index, what, host_buf, bytes);
}

int iiv_encoder_set_state(iiv_encoder *enc, int stream_index, int what, const void *host_buf, size_t bytes)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_state(enc->impl, stream_index, 1, what, host_buf, bytes);
}

int iiv_encoder_set_state_async(iiv_encoder *enc, int stream_index, int what, const void *host_buf, size_t bytes, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_state_async(enc->impl, stream_index, what, host_buf, bytes, (hipStream_t)stream);
}

int iiv_encoder_get_video_state(iiv_encoder *enc, int stream_index, iiv_video_state *host_out)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_get_video_state(enc->impl, stream_index, host_out);
}

int iiv_encoder_get_video_brief(iiv_encoder *enc, int stream_index, iiv_video_brief *host_out)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_get_video_brief(enc->impl, stream_index, host_out);
}

int iiv_encoder_get_video_brief_async(iiv_encoder *enc, int stream_index, iiv_video_brief *host_out, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_get_video_brief_async(enc->impl, stream_index, host_out, (hipStream_t)stream);
}

int iiv_encoder_set_video_state(iiv_encoder *enc, int stream_index, const iiv_video_state *host_in)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_video_state(enc->impl, stream_index, host_in);
}

int iiv_encoder_set_state_range(iiv_encoder *enc, int first_stream, int n_streams, int what, const void *host_buf,
                                size_t bytes_per_stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_set_state(enc->impl, first_stream, n_streams, what, host_buf, bytes_per_stream);
}

int iiv_encode(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux, int n_frames,
               const iiv_segment *segments, int n_segments, uint8_t *d_ops_out, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encode(enc->impl, d_frames_main, d_frames_aux, n_frames, segments, n_segments, d_ops_out,
                       (hipStream_t)stream);
}

int iiv_encoder_live_queue(iiv_encoder *enc, int slot, uint64_t **host_queue, int *capacity)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_live_queue(enc->impl, slot, host_queue, capacity);
}

int iiv_encode_live(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux, int n_frames,
                    const iiv_segment *segments, int n_segments, uint8_t *d_ops_out, int slot, uint32_t tag, void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encode_live(enc->impl, d_frames_main, d_frames_aux, n_frames, segments, n_segments, d_ops_out, slot, tag,
                            (hipStream_t)stream);
}

int iiv_encode_streams(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux, int n_frames,
                       const iiv_segment *segments, const int32_t *seg_begin, uint8_t *d_ops_out, size_t ops_stride,
                       void *stream)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encode_streams(enc->impl, d_frames_main, d_frames_aux, n_frames, segments, seg_begin, d_ops_out, ops_stride,
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

int iiv_encoder_launch_forms(iiv_encoder *enc, int64_t counts[4])
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_launch_forms(enc->impl, counts);
}

int iiv_encoder_input_stats(iiv_encoder *enc, double stats[2], int *form)
{
    if (!enc) return iiv::set_error(IIV_ERR_INVALID, "null encoder");
    return iiv::encoder_input_stats(enc->impl, stats, form);
}

int iiv_check_split_diff_table(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches,
                               void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || !dm || !d_table || !mismatches)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_check_split_diff_table: bad argument");
    return iiv::check_split_dw_table(mode, dm, d_table, mismatches, (hipStream_t)stream);
}

int iiv_check_diff_weight_pieces(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches,
                                 void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || !dm || !d_table || !mismatches)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_check_diff_weight_pieces: bad argument");
    return iiv::check_dw_piece_table(mode, dm, d_table, mismatches, (hipStream_t)stream);
}

int iiv_build_split_store_table(int mode, const int32_t dm[256], uint32_t *d_left, uint32_t *d_right,
                                uint16_t *d_expanded, void *stream)
{
    if (mode != IIV_HGR && mode != IIV_DHGR) return iiv::set_error(IIV_ERR_INVALID, "mode must be IIV_HGR or IIV_DHGR");
    if (!dm) return iiv::set_error(IIV_ERR_INVALID, "dm is NULL");
    return iiv::build_split_store_table(mode, dm, d_left, d_right, d_expanded, (hipStream_t)stream);
}

int iiv_build_narrow_store_table(int mode, const int32_t dm[256], const uint16_t *d_store_table, uint16_t *d_expanded,
                                 unsigned long long *n_mismatch, void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || !dm || !d_store_table || !d_expanded || !n_mismatch)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_build_narrow_store_table: bad argument");
    return iiv::build_narrow_store_table(mode, dm, d_store_table, d_expanded, n_mismatch, (hipStream_t)stream);
}

size_t iiv_split_table_entries(int mode, int right_half) { return iiv::split_entries(mode, right_half ? 1 : 0); }

}  // extern "C"
