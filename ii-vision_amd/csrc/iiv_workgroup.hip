// iiv_workgroup.hip -- the greedy selection loop (transcoder/video.py:121-187, 275-301; transcoder/screen.py:256-293)
// as one 256-thread workgroup per stream, one lane per page byte, reading the dense u16 store table: the second,
// independent implementation beside iiv_greedy.hip / iiv_team.hip (every parity test runs both), the kernel of
// encoders created without a diff matrix (user-supplied tables), and the home of the joint content choice (f4).
#include "iiv_host.h"
#include "iiv_edit.h"
#include "iiv_stream.h"
#include "iiv_wave.h"

namespace iiv {

// ------------------------------------------------------------------------- greedy

constexpr int kChunk = 8;  // initial-list entries whose store-table rows are gathered together

// JOINT (f4, README.md:212-215 "Global optimization"; NOT reference behaviour, IIV_OPT_CONTENT_CHOICE):
// the content byte of a step is not the primary's target byte but the value c that maximises
//     R(c) = (dw[primary] - nd_c[primary]) - (d1 + d2),
// d1, d2 = the two smallest negative deltas nd_c[y] - dw[y] among the page's other bytes with non-zero
// priority (ties: the target byte, then the smallest c), and the primary keeps nd_c[primary] as its
// priority.  All else is the reference's step applied to the chosen byte.  Every wave scores all byte values against
// the eligible bytes of its quarter of the page.  JOINT == 2 (default): from the narrow form packed two byte values per
// word (iiv_tables.hip: joint_pack_kernel) with packed 16-bit arithmetic -- two loads and five vector instructions per
// byte of the page and PAIR of byte values; JOINT == 1: from the two-component split table, one byte value at a time
// (round 2's form, kept as the independent second implementation: IIV_CONTENT_JOINT_SPLIT).
// FOUR (round 6, joint choice only: IIV_OPT_FOURTH_OFFSET together with IIV_CONTENT_JOINT): up to three extra offsets per
// opcode; the joint score of a byte value then takes its THREE smallest negative deltas (oracle/iiv_oracle.c:
// choose_content_joint).  Without the joint choice the fourth offset runs in the one-wave / team kernels.
template <int MODE, int JOINT, bool FOUR = false>
__global__ __launch_bounds__(256, JOINT ? 5 : 1) void greedy_kernel(StreamState *__restrict__ states,
                                                     const uint8_t *__restrict__ frames_main,
                                                     const uint8_t *__restrict__ frames_aux, int n_frames,
                                                     const LaunchSeg *__restrict__ segs, int seg_stride,
                                                     const uint16_t *__restrict__ store,
                                                     const uint32_t *__restrict__ left_t,
                                                     const uint32_t *__restrict__ right_t,
                                                     uint8_t *__restrict__ ops_out, size_t ops_stride)
{
    const LaunchSeg seg = segs[(size_t)blockIdx.x * seg_stride];
    const int n_ops = __builtin_amdgcn_readfirstlane(seg.n_ops), is_aux = __builtin_amdgcn_readfirstlane(seg.is_aux);
    const int frame = __builtin_amdgcn_readfirstlane(seg.frame);
    const size_t ops_base = (size_t)__builtin_amdgcn_readfirstlane(seg.ops_base) * 6;
    if (n_ops <= 0) return;
    constexpr int BITS = ModeTraits<MODE>::kBits;
    constexpr int CB = ModeTraits<MODE>::kContentBits;
    constexpr int NB = ModeTraits<MODE>::kBanks;
    constexpr uint32_t INF = 0xffffffffu;
    // target bytes, [0] = bank being encoded, [1] = the other one.  JOINT: read where they lie (a step looks at two rows of them
    // against 32768 table values): without their 16 KiB five workgroups fit a CU instead of four, and the joint step is short
    // of waves, not of bandwidth
    constexpr bool kTgtLds = !JOINT;
    __shared__ __attribute__((aligned(16))) uint8_t tgt_lds[kTgtLds ? NB : 1][kTgtLds ? 8192 : 16];
    __shared__ __attribute__((aligned(16))) uint16_t dwf[8192];     // diff_weight | (priority != 0) << 15
    __shared__ uint32_t mt[2][624];
    __shared__ uint32_t xw_cnt[4];
    __shared__ uint32_t xw_key[12];
    __shared__ unsigned long long xw_pop[4];
    __shared__ int xw_joint[4];
    __shared__ uint32_t xw_m12[JOINT ? 4 : 1][JOINT ? (1 << ModeTraits<MODE>::kContentBits) : 1];
    __shared__ uint16_t xw_m3[(JOINT && FOUR) ? 4 : 1][(JOINT && FOUR) ? (1 << ModeTraits<MODE>::kContentBits) : 1];
    constexpr int CH = JOINT ? 1 : kChunk;  // (a joint step knows its content only after scoring every value)

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
    const uint8_t *tgt[2];
    if constexpr (kTgtLds) {
        for (int i = tid; i < 512 * NB; i += 256) {
            int b = i >> 9, k = i & 511;
            const uint8_t *src;
            if (MODE == kDHGR)
                src = ((b == 0) == (is_aux != 0) ? frames_aux : frames_main) + fbase;
            else
                src = frames_main + fbase;
            reinterpret_cast<uint4 *>(tgt_lds[b])[k] = reinterpret_cast<const uint4 *>(src)[k];
        }
        tgt[0] = tgt_lds[0];
        tgt[1] = tgt_lds[NB - 1];
    } else {
        tgt[0] = (MODE == kDHGR && is_aux ? frames_aux : frames_main) + fbase;
        tgt[1] = (MODE == kDHGR ? (is_aux ? frames_main : frames_aux) : frames_main) + fbase;
    }
    for (int i = tid; i < 8192; i += 256) {
        uint32_t bit = (S.nzbits[i >> 5] >> (i & 31)) & 1u;
        uint32_t dn = (S.pdone[i >> 5] >> (i & 31)) & 1u;
        dwf[i] = (uint16_t)((dn ? 0u : wd_dw(S.wd[i])) | (bit << 15));
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
        uint32_t ent[CH];
        int pos[CH];
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
            for (int m = 0; m < CH; m++) {
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
            chunk_end = mask ? pos[CH - 1] + 1 : window_end;
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
            for (int m = 0; m < CH; m++) {
                ent[m] = 0;
                pos[m] = 0;
            }
            ent[0] = bk & 0x1fff;
            cnt = 1;
            if (!(dwf[ent[0]] & 0x8000u)) continue;  // video.py:130
        }

        // ---- joint mode: choose the entry's content byte.  Lane l holds the byte values l, l + 64, ...
        // (two per lane DHGR, four HGR); wave w walks the eligible bytes of its quarter of the page: a
        // byte's row parts are wave-uniform, so the row of every byte value is one (left) or two (right)
        // cache lines read with lane-consecutive addresses, and each lane keeps the two smallest
        // deltas of its own byte values -- no reduction until the waves' quarters are merged in LDS.
        uint32_t joint_c = 0, joint_res = 0;
        if (JOINT) {
            using T = SplitTraits<MODE>;
            constexpr int NS = (1 << CB) / 64;
            const int p = ent[0] >> 8, x = ent[0] & 255;
            const uint32_t tc = tgt[0][ent[0]];
            const uint8_t *own_row = tgt[0] + p * 256;
            const uint8_t *oth_row = tgt[NB - 1] + p * 256;
            // content parts of this lane's byte values, for even and odd bytes
            uint32_t cl[NS][2], cr[NS][2];
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int od = 0; od < 2; od++) {
                    cl[j][od] = split_content_left<MODE>((uint32_t)(lane + 64 * j), od);
                    cr[j][od] = split_content_right<MODE>((uint32_t)(lane + 64 * j), od);
                }
            // values of every byte value for the byte with window `win` at parity `od`
            auto row_values = [&](uint32_t win, int yy, int (&nd)[NS]) {
                const int od = yy & 1, o = byte_offset<MODE>(yy, is_aux);
                const uint32_t *lrow = left_t + ((((size_t)o << T::kLeftRowBits) + split_row_left<MODE>(win, od)) << T::kLeftCBits);
                const uint32_t *rrow =
                    right_t + ((((size_t)o << T::kRightRowBits) + split_row_right<MODE>(win, od)) << T::kRightCBits);
#pragma unroll
                for (int j = 0; j < NS; j++) nd[j] = (int)combine(lrow[cl[j][od]], rrow[cr[j][od]]);
            };
            // this lane's byte of the wave's quarter: window, diff weight, eligibility
            const int ym = 64 * wave + lane;
            uint32_t pv, nx;
            neighbours<MODE>(own_row, oth_row, ym, is_aux, pv, nx);
            const uint32_t win_m = masked_window<MODE>(pv, own_row[ym], nx, ym & 1);
            const uint32_t wv = dwf[p * 256 + ym];
            const int dw_m = (int)(wv & 0x7fffu);
            // (a byte with priority 0, the primary itself, or a zero diff weight can never yield d < 0)
            unsigned long long todo = __ballot((wv & 0x8000u) && ym != x && dw_m != 0);
            typedef short v2s __attribute__((ext_vector_type(2)));
            constexpr int NP = JointPack<MODE>::kPairs;
            // JOINT == 2: the packed words of every byte value for the byte with window `win` at parity `od`: lane l gets the
            // sums L1 + RF (= store value + kNarrowBias) of byte values l + 128 j (low half) and l + 128 j + 64 (high half)
            auto row_packed = [&](uint32_t win, int yy, v2s (&nd)[NP]) {
                const int od = yy & 1, o = byte_offset<MODE>(yy, is_aux);
                const uint32_t *lrow = left_t + ((((size_t)o << T::kLeftRowBits) + split_row_left<MODE>(win, od)) * NP) * 64 + lane;
                const uint32_t *rrow = right_t + ((((size_t)o << T::kRightRowBits) + split_row_right<MODE>(win, od)) * NP) * 64 + lane;
#pragma unroll
                for (int j = 0; j < NP; j++) nd[j] = __builtin_bit_cast(v2s, lrow[64 * j]) + __builtin_bit_cast(v2s, rrow[64 * j]);
            };
            int m1[NS], m2[NS], m3[NS];  // per byte value: the two (FOUR: three) smallest negative deltas (m1 <= m2 <= m3 <= 0)
#pragma unroll
            for (int j = 0; j < NS; j++) m3[j] = 0;
            if constexpr (JOINT == 2) {
                // per lane, once per step: where the rows of its byte of the quarter start (byte offsets into the two packed
                // tables) and its diff weight plus the bias of the sums -- a trip then needs three v_readlane per byte and no
                // address arithmetic: the loads are buffer loads, row offset in a scalar register, 4 * lane in the vector one
                const int od_m = ym & 1, o_m = byte_offset<MODE>(ym, is_aux);
                const uint32_t lo_m = (uint32_t)((((size_t)o_m << T::kLeftRowBits) + split_row_left<MODE>(win_m, od_m)) * NP) << 8;
                const uint32_t ro_m = (uint32_t)((((size_t)o_m << T::kRightRowBits) + split_row_right<MODE>(win_m, od_m)) * NP) << 8;
                const int db_m = dw_m + (int)kNarrowBias;   // (diff weights are < 2^12: iiv_stream.h, kWdDwShift)
                const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void *)left_t, 0, (int)(joint_left_entries<MODE>() * 4), 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)right_t, 0, (int)(joint_right_entries<MODE>() * 4), 0x00020000);
                v2s q1[NP], q2[NP], q3[NP];
#pragma unroll
                for (int j = 0; j < NP; j++) q1[j] = q2[j] = q3[j] = v2s{0, 0};
#ifndef IIV_JOINT_U
#define IIV_JOINT_U 8
#endif
                constexpr int U = IIV_JOINT_U / NP;   // bytes per trip, their rows in flight together
                auto score = [&](const int (&lo)[U], const int (&ro)[U], const int (&db)[U]) {
                    v2s nd[U][NP];
#pragma unroll
                    for (int u = 0; u < U; u++)
#pragma unroll
                        for (int j = 0; j < NP; j++)
                            nd[u][j] = __builtin_bit_cast(v2s, __builtin_amdgcn_raw_buffer_load_b32(rs_l, 4 * lane + 256 * j, lo[u], 0)) +
                                       __builtin_bit_cast(v2s, __builtin_amdgcn_raw_buffer_load_b32(rs_r, 4 * lane + 256 * j, ro[u], 0));
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const short b = (short)db[u];
                        const v2s bb = v2s{b, b};
#pragma unroll
                        for (int j = 0; j < NP; j++) {
                            const v2s d = nd[u][j] - bb;
                            const v2s lo2 = __builtin_elementwise_min(d, q1[j]), hi2 = __builtin_elementwise_max(d, q1[j]);
                            q1[j] = lo2;
                            if constexpr (FOUR) {
                                const v2s hi3 = __builtin_elementwise_max(hi2, q2[j]);
                                q2[j] = __builtin_elementwise_min(hi2, q2[j]);
                                q3[j] = __builtin_elementwise_min(hi3, q3[j]);
                            } else {
                                q2[j] = __builtin_elementwise_min(hi2, q2[j]);
                            }
                        }
                    }
                };
                while (todo) {
                    // (a missing byte repeats the first with diff weight 0: d >= 0 changes nothing.  Walking all 64 bytes of the
                    // quarter without the mask bookkeeping was measured: no faster, DHGR, and 7 % slower, HGR; so was the left row
                    // offset riding in the diff-weight word's high half -- one v_readlane less, two scalar shifts more: 3 % slower)
                    int lo[U], ro[U], db[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (todo) {
                            const int k = __builtin_ctzll(todo);
                            todo &= todo - 1;
                            lo[u] = __builtin_amdgcn_readlane((int)lo_m, k);
                            ro[u] = __builtin_amdgcn_readlane((int)ro_m, k);
                            db[u] = __builtin_amdgcn_readlane(db_m, k);
                        } else {
                            lo[u] = lo[0];
                            ro[u] = ro[0];
                            db[u] = (int)kNarrowBias;
                        }
                    }
                    score(lo, ro, db);
                }
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    m1[2 * j] = q1[j].x, m1[2 * j + 1] = q1[j].y;
                    m2[2 * j] = q2[j].x, m2[2 * j + 1] = q2[j].y;
                    m3[2 * j] = q3[j].x, m3[2 * j + 1] = q3[j].y;
                }
            } else {
#pragma unroll
            for (int j = 0; j < NS; j++) m1[j] = m2[j] = 0;
            while (todo) {
                // four bytes per trip, their rows in flight together (a missing one repeats the first
                // with diff weight 0: d >= 0 changes nothing)
                constexpr int U = 4;
                uint32_t wins[U];
                int dws[U], ys[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (todo) {
                        const int k = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        wins[u] = __builtin_amdgcn_readlane(win_m, k);
                        dws[u] = __builtin_amdgcn_readlane(dw_m, k);
                        ys[u] = 64 * wave + k;
                    } else {
                        wins[u] = wins[0];
                        dws[u] = 0;
                        ys[u] = ys[0];
                    }
                }
                int nd[U][NS];
#pragma unroll
                for (int u = 0; u < U; u++) row_values(wins[u], ys[u], nd[u]);
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const int d = nd[u][j] - dws[u];
                        const int lo = d < m1[j] ? d : m1[j], hi = d < m1[j] ? m1[j] : d;
                        m1[j] = lo;
                        if constexpr (FOUR) {
                            const int hi3 = hi < m2[j] ? m2[j] : hi;
                            m3[j] = hi3 < m3[j] ? hi3 : m3[j];
                        }
                        m2[j] = hi < m2[j] ? hi : m2[j];
                    }
            }
            }
#pragma unroll
            for (int j = 0; j < NS; j++) {
                xw_m12[wave][lane + 64 * j] = (uint32_t)(uint16_t)m1[j] | ((uint32_t)(uint16_t)m2[j] << 16);
                if constexpr (FOUR) xw_m3[wave][lane + 64 * j] = (uint16_t)m3[j];
            }
            // the primary's own byte (uniform)
            const uint32_t winx = __builtin_amdgcn_readfirstlane(__shfl(win_m, x & 63, 64));  // valid in wave x >> 6 only
            if (wave == (x >> 6) && lane == 0) xw_joint[0] = (int)winx;
            __syncthreads();
            int ndx[NS];
            if constexpr (JOINT == 2) {
                v2s nx2[NP];
                row_packed((uint32_t)xw_joint[0], x, nx2);
#pragma unroll
                for (int j = 0; j < NP; j++) {
                    ndx[2 * j] = (int)nx2[j].x - (int)kNarrowBias;
                    ndx[2 * j + 1] = (int)nx2[j].y - (int)kNarrowBias;
                }
            } else {
                row_values((uint32_t)xw_joint[0], x, ndx);
            }
            const int dwx = (int)(dwf[p * 256 + x] & 0x7fffu);
            int best = -2147483647 - 1;
#pragma unroll
            for (int j = 0; j < NS; j++) {
                int a1 = 0, a2 = 0, a3 = 0;
                // (a value goes into the sorted triple a1 <= a2 <= a3: five minima / maxima; the fourth-offset form merges the
                // waves' triples that way, the reference form their pairs as before)
                auto put = [&](int b) {
                    const int lo = b < a1 ? b : a1, hi = b < a1 ? a1 : b;
                    a1 = lo;
                    const int lo2 = hi < a2 ? hi : a2, hi2 = hi < a2 ? a2 : hi;
                    a2 = lo2;
                    a3 = hi2 < a3 ? hi2 : a3;
                };
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t v = xw_m12[w][lane + 64 * j];
                    const int b1 = (int)(int16_t)(v & 0xffffu), b2 = (int)(int16_t)(v >> 16);
                    if constexpr (FOUR) {
                        put(b1);
                        put(b2);
                        put((int)(int16_t)xw_m3[w][lane + 64 * j]);
                    } else {
                        const int lo = b1 < a1 ? b1 : a1, hi = b1 < a1 ? a1 : b1;
                        a1 = lo;
                        a2 = hi < a2 ? hi : a2;
                        a2 = b2 < a2 ? b2 : a2;
                    }
                }
                const int c = lane + 64 * j;
                const int key = (dwx - ndx[j] - a1 - a2 - (FOUR ? a3 : 0)) * 512 + ((uint32_t)c == tc ? 256 : 0) + (255 - c);
                best = key > best ? key : best;
            }
            best = -wave_min_i32(-best);
            joint_c = 255u - ((uint32_t)best & 255u);
            // what the chosen value leaves at the primary: lane joint_c & 63 holds it in slot joint_c >> 6
            int res = 0;
#pragma unroll
            for (int j = 0; j < NS; j++) res = (int)(joint_c >> 6) == j ? ndx[j] : res;
            joint_res = (uint32_t)__builtin_amdgcn_readlane(res, joint_c & 63);
            __syncthreads();  // (xw_m12 / xw_joint are rewritten by the next step)
        }

        // ---- gather the store-table row of every chunk entry (all in flight together)
        uint32_t ndv[CH];
#pragma unroll
        for (int m = 0; m < CH; m++) {
            {   // branch-free: see greedy_wave_kernel
                int p = ent[m] >> 8;
                uint32_t c = JOINT ? joint_c : tgt[0][ent[m]];
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
        for (int m = 0; m < CH; m++) asm volatile("" : "+v"(ndv[m]));
        __builtin_amdgcn_sched_barrier(0);

        // ---- process the chunk sequentially
        uint32_t dead = 0;
#pragma unroll
        for (int m = 0; m < CH; m++) {
            if (m >= cnt || done >= n_ops || err) break;
            if (dead & (1u << m)) {
                head = pos[m] + 1;
                continue;
            }
            const int p = ent[m] >> 8, x = ent[m] & 255;
            if (MODE == kDHGR && tgt[0][ent[m]] >= 0x80) {   // video.py:137
                err = kErrPaletteBit;
                break;
            }
            const uint32_t c = JOINT ? joint_c : tgt[0][ent[m]];  // video.py:134
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
            uint32_t k1 = key, k2 = INF, k3 = INF;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                uint32_t o1 = __shfl_xor(k1, s, 64), o2 = __shfl_xor(k2, s, 64);
                uint32_t lo = k1 < o1 ? k1 : o1, hi = k1 < o1 ? o1 : k1;
                uint32_t m2 = k2 < o2 ? k2 : o2;
                if constexpr (FOUR) {   // third of the two sorted triples: the smallest of what is left behind the first two
                    const uint32_t o3 = __shfl_xor(k3, s, 64);
                    const uint32_t M2 = k2 < o2 ? o2 : k2, r = hi < m2 ? m2 : hi, m3 = k3 < o3 ? k3 : o3;
                    const uint32_t t = r < M2 ? r : M2;
                    k3 = t < m3 ? t : m3;
                }
                k1 = lo;
                k2 = hi < m2 ? hi : m2;
            }
            if (lane == 0) {
                xw_key[3 * wave] = k1;
                xw_key[3 * wave + 1] = k2;
                xw_key[3 * wave + 2] = FOUR ? k3 : INF;
            }
            __syncthreads();
            uint32_t K1 = INF, K2 = INF, K3 = INF;
#pragma unroll
            for (int q = 0; q < 12; q++) {
                uint32_t k = xw_key[q];
                if (k < K1) {
                    K3 = K2;
                    K2 = K1;
                    K1 = k;
                } else if (k < K2) {
                    K3 = K2;
                    K2 = k;
                } else if (k < K3) {
                    K3 = k;
                }
            }
            if (!FOUR) K3 = INF;
            const int y1 = K1 != INF ? (int)((K1 >> 1) & 255) : -1;
            const int f1 = K1 != INF ? (int)(K1 & 1) : 0;
            const int y2 = K2 != INF ? (int)((K2 >> 1) & 255) : -1;
            const int f2 = K2 != INF ? (int)(K2 & 1) : 0;
            const int y3 = K3 != INF ? (int)((K3 >> 1) & 255) : -1;
            const int f3 = K3 != INF ? (int)(K3 & 1) : 0;
            if (n_pushed + f1 + f2 + f3 > kPushedCap) {
                err = kErrPushedOverflow;
                break;
            }

            // ---- apply (video.py:140-144, 170-178; screen.py:256-293)
            if (y == x) {
                // (its diff weight counts as 0 from here on, video.py:141; its priority is 0 -- or, with the joint choice, the
                // error the chosen byte leaves: then the location stays live for a later pop of a stale bag entry, as in the
                // oracle's definition, video.py:130)
                dwf[p * 256 + x] = (JOINT && joint_res) ? (uint16_t)0x8000u : (uint16_t)0;
                S.up16[is_aux][p * 256 + x] = JOINT ? (uint16_t)joint_res : (uint16_t)0;   // (store values are <= 2047: the 16-bit copy)
                S.mem[is_aux][p * 256 + x] = (uint8_t)c;
            }
            if (y == y1 || y == y2 || y == y3) {
                const int before = (y == y2) ? f1 : (y == y3) ? f1 + f2 : 0;   // re-queued entries of this step in front of this one
                S.up16[is_aux][p * 256 + y] = (uint16_t)nd;  // byte_pair_difference == nd[y] (screen.py:383-398)
                S.mem[is_aux][p * 256 + y] = (uint8_t)c;
                dwf[p * 256 + y] = (uint16_t)((w & 0x7fffu) | (nd ? 0x8000u : 0u));
                if (nd) {
                    int j = mt_idx + C + before;
                    uint32_t word = j < 624 ? mt[cb][j] : mt[cb ^ 1][j - 624];
                    uint32_t nonce = mt_temper(word) >> 24;  // video.py:178
                    S.pushed[n_pushed + before] =
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
                q[5] = (uint8_t)(y3 >= 0 ? y3 : x);   // (FOUR only: y3 is -1 otherwise)
            }
            // later chunk entries that this step resolved exactly are now dead
#pragma unroll
            for (int m2 = 0; m2 < CH; m2++)
                if (m2 > m && m2 < cnt) {
                    if (y1 >= 0 && !f1 && ent[m2] == (uint32_t)((p << 8) | y1)) dead |= 1u << m2;
                    if (y2 >= 0 && !f2 && ent[m2] == (uint32_t)((p << 8) | y2)) dead |= 1u << m2;
                    if (y3 >= 0 && !f3 && ent[m2] == (uint32_t)((p << 8) | y3)) dead |= 1u << m2;
                }
            mt_idx += C + f1 + f2 + f3;
            draws += (unsigned long long)(C + f1 + f2 + f3);
            n_pushed += f1 + f2 + f3;
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
            if ((v & 0x7fffu) == 0 && wd_dw(S.wd[wi * 32 + b]) != 0) pdw |= 1u << b;
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
int launch_greedy_workgroup(int mode, int joint, bool fourth, const WorkgroupArgs &a, hipStream_t st)
{
#define IIV_GREEDY(K)                                                                                                     \
    hipLaunchKernelGGL(K, dim3(a.n_streams), dim3(256), 0, st, a.states, a.frames_main, a.frames_aux, a.n_frames, a.segs, \
                       a.seg_stride, a.store, a.left_t, a.right_t, a.ops_out, a.ops_stride)
    if (joint == 2 && fourth) {
        if (mode == kDHGR) IIV_GREEDY((greedy_kernel<kDHGR, 2, true>)); else IIV_GREEDY((greedy_kernel<kHGR, 2, true>));
    } else if (joint && fourth) {
        if (mode == kDHGR) IIV_GREEDY((greedy_kernel<kDHGR, 1, true>)); else IIV_GREEDY((greedy_kernel<kHGR, 1, true>));
    } else if (joint == 2) {
        if (mode == kDHGR) IIV_GREEDY((greedy_kernel<kDHGR, 2>)); else IIV_GREEDY((greedy_kernel<kHGR, 2>));
    } else if (joint) {
        if (mode == kDHGR) IIV_GREEDY((greedy_kernel<kDHGR, 1>)); else IIV_GREEDY((greedy_kernel<kHGR, 1>));
    } else {
        if (mode == kDHGR) IIV_GREEDY((greedy_kernel<kDHGR, 0>)); else IIV_GREEDY((greedy_kernel<kHGR, 0>));
    }
#undef IIV_GREEDY
    return hip_check(hipGetLastError(), "greedy_kernel launch");
}

}  // namespace iiv
