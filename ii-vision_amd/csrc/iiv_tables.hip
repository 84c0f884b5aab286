// iiv_tables.hip -- make_data_tables on gfx950 (reference: transcoder/make_data_tables.py).
//
// K3 cie2000_kernel     compute_diff_matrix            make_data_tables.py:55-70
// K1 pixel_kernel       to_dots + colour model         screen.py:743-789, colours.py:100-148
// K2 table_kernel       compute_edit_distance          make_data_tables.py:111-174
//    store_kernel       the (target window, content) sub-table the greedy loop reads
//
// The table build is integer ALU + streaming u16 stores: each thread owns 8
// consecutive j columns (their colour strings live in registers) and walks a
// tile of i rows whose string is wave-uniform, so every wave store is 1 KiB
// contiguous.  No MFMA: there is no contraction here, only a 10/18-step
// min-plus recurrence per pair.
#include "iiv_host.h"
#include "iiv_edit.h"
#include "iiv_stream.h"

namespace iiv {

// ------------------------------------------------------------------ CIE2000

// colormath 3.0.0 RGB_to_XYZ + XYZ_to_Lab (sRGB, D65/2deg, no adaptation).
__device__ static void rgb_to_lab(const uint8_t *rgb, double lab[3])
{
    double lin[3];
    for (int i = 0; i < 3; i++) {
        double v = rgb[i] / 255.0;
        lin[i] = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
    }
    const double M[3][3] = {{0.412424, 0.357579, 0.180464},
                            {0.212656, 0.715158, 0.0721856},
                            {0.0193324, 0.119193, 0.950444}};
    const double illum[3] = {0.95047, 1.00000, 1.08883};
    const double CIE_E = 216.0 / 24389.0;
    double t[3];
    for (int r = 0; r < 3; r++) {
        double s = M[r][0] * lin[0] + M[r][1] * lin[1] + M[r][2] * lin[2];
        s = s > 0.0 ? s : 0.0;
        double v = s / illum[r];
        t[r] = v > CIE_E ? pow(v, 1.0 / 3.0) : (7.787 * v) + (16.0 / 116.0);
    }
    lab[0] = (116.0 * t[1]) - 16.0;
    lab[1] = 500.0 * (t[0] - t[1]);
    lab[2] = 200.0 * (t[1] - t[2]);
}

__device__ static inline double deg(double r) { return r * (180.0 / 3.14159265358979323846); }
__device__ static inline double rad(double d) { return d * (3.14159265358979323846 / 180.0); }

// colormath 3.0.0 color_diff_matrix.delta_e_cie2000, Kl=Kc=Kh=1.
__device__ static double delta_e_cie2000(const double c1[3], const double c2[3])
{
    double L = c1[0], a = c1[1], b = c1[2];
    double L2 = c2[0], a2 = c2[1], b2 = c2[2];
    double avg_Lp = (L + L2) / 2.0;
    double C1 = sqrt(a * a + b * b);
    double C2 = sqrt(a2 * a2 + b2 * b2);
    double avg_C = (C1 + C2) / 2.0;
    double G = 0.5 * (1 - sqrt(pow(avg_C, 7.0) / (pow(avg_C, 7.0) + pow(25.0, 7.0))));
    double a1p = (1.0 + G) * a;
    double a2p = (1.0 + G) * a2;
    double C1p = sqrt(a1p * a1p + b * b);
    double C2p = sqrt(a2p * a2p + b2 * b2);
    double avg_Cp = (C1p + C2p) / 2.0;
    double h1p = deg(atan2(b, a1p));
    h1p += (h1p < 0) * 360;
    double h2p = deg(atan2(b2, a2p));
    h2p += (h2p < 0) * 360;
    double avg_Hp = (((fabs(h1p - h2p) > 180) * 360) + h1p + h2p) / 2.0;
    double T = 1 - 0.17 * cos(rad(avg_Hp - 30)) + 0.24 * cos(rad(2 * avg_Hp)) +
               0.32 * cos(rad(3 * avg_Hp + 6)) - 0.2 * cos(rad(4 * avg_Hp - 63));
    double diff = h2p - h1p;
    double delta_hp = diff + (fabs(diff) > 180) * 360;
    delta_hp -= (h2p > h1p) * 720;
    double dLp = L2 - L;
    double dCp = C2p - C1p;
    double dHp = 2 * sqrt(C2p * C1p) * sin(rad(delta_hp) / 2.0);
    double S_L = 1 + ((0.015 * pow(avg_Lp - 50, 2.0)) / sqrt(20 + pow(avg_Lp - 50, 2.0)));
    double S_C = 1 + 0.045 * avg_Cp;
    double S_H = 1 + 0.015 * avg_Cp * T;
    double d_ro = 30 * exp(-(pow(((avg_Hp - 275) / 25), 2.0)));
    double R_C = sqrt((pow(avg_Cp, 7.0)) / (pow(avg_Cp, 7.0) + pow(25.0, 7.0)));
    double R_T = -2 * R_C * sin(2 * rad(d_ro));
    return sqrt(pow(dLp / S_L, 2.0) + pow(dCp / S_C, 2.0) + pow(dHp / S_H, 2.0) +
                R_T * (dCp / S_C) * (dHp / S_H));
}

__global__ void cie2000_kernel(const uint8_t *rgb, double *out_f, int32_t *out_i)
{
    int t = threadIdx.x;  // 256 threads: (i, j)
    int i = t >> 4, j = t & 15;
    double l1[3], l2[3];
    rgb_to_lab(rgb + 3 * i, l1);
    rgb_to_lab(rgb + 3 * j, l2);
    double d = delta_e_cie2000(l1, l2);
    out_f[t] = d;
    out_i[t] = (int32_t)d;  // int(): truncation (make_data_tables.py:68)
}

// the same delta-E on caller-supplied Lab pairs (published CIEDE2000 test data, tests)
__global__ void delta_e_kernel(int n, const double *__restrict__ lab1, const double *__restrict__ lab2,
                               double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a[3] = {lab1[3 * i], lab1[3 * i + 1], lab1[3 * i + 2]};
    const double b[3] = {lab2[3 * i], lab2[3 * i + 1], lab2[3 * i + 2]};
    out[i] = delta_e_cie2000(a, b);
}

// ------------------------------------------------------------------ colour strings

template <int MODE>
__global__ void pixel_kernel(uint32_t *dots_out, uint8_t *pix_out, ulonglong2 *strings)
{
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;  // o * 2^bits + m
    if (idx >= (ModeTraits<MODE>::kOffsets << BITS)) return;
    int o = idx >> BITS;
    uint32_t m = idx & ((1u << BITS) - 1);
    uint64_t lo;
    uint32_t hi;
    colour_string<MODE>(m, o, lo, hi);
    if (strings) strings[idx] = make_ulonglong2(lo, (unsigned long long)hi);
    if (dots_out) dots_out[idx] = to_dots<MODE>(m, o);
    if (pix_out)
        for (int k = 0; k < ND; k++)
            pix_out[(size_t)idx * ND + k] = (uint8_t)(k < 16 ? (lo >> (4 * k)) & 0xf : (hi >> (4 * (k - 16))) & 0xf);
}


// ------------------------------------------------------------------ edit distance tables

constexpr int kTableRowsPerBlock = 32;
constexpr int kTableColsPerThread = 8;

// grid: x = column tile (2^bits / (256*8)), y = row tile (2^bits / 32), z = byte offset
template <int MODE>
__global__ __launch_bounds__(256) void table_kernel(const ulonglong2 *__restrict__ strings,
                                                    const uint16_t *__restrict__ sub, uint16_t *__restrict__ out,
                                                    int symmetric)
{
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots;
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    const int o = blockIdx.z;
    const ulonglong2 *S = strings + ((size_t)o << BITS);
    uint16_t *T = out + ((size_t)o << (2 * BITS));
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * kTableColsPerThread;
    uint64_t blo[kTableColsPerThread];
    uint32_t bhi[kTableColsPerThread];
#pragma unroll
    for (int c = 0; c < kTableColsPerThread; c++) {
        ulonglong2 v = S[j0 + c];
        blo[c] = v.x;
        bhi[c] = (uint32_t)v.y;
    }
    const int i0 = blockIdx.y * kTableRowsPerBlock;
    for (int r = 0; r < kTableRowsPerBlock; r++) {
        const int i = i0 + r;
        ulonglong2 av = S[i];  // wave-uniform
        uint64_t alo = av.x;
        uint32_t ahi = (uint32_t)av.y;
        uint32_t e[kTableColsPerThread];
#pragma unroll
        for (int c = 0; c < kTableColsPerThread; c++) {
            uint32_t v = edit_distance<ND>(alo, ahi, blo[c], bhi[c], lut);
            // make_data_tables.py:156: only j < i is stored; screen.py:358-365 mirrors it
            if (!symmetric && (j0 + c) >= i) v = 0;
            e[c] = v;
        }
        uint4 pk;
        pk.x = e[0] | (e[1] << 16);
        pk.y = e[2] | (e[3] << 16);
        pk.z = e[4] | (e[5] << 16);
        pk.w = e[6] | (e[7] << 16);
        *reinterpret_cast<uint4 *>(T + ((size_t)i << BITS) + j0) = pk;
    }
}

// S[o][content][m] = ED(string(poke(m, content)), string(m))
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(const ulonglong2 *__restrict__ strings,
                                                    const uint16_t *__restrict__ sub, uint16_t *__restrict__ out)
{
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots, CB = ModeTraits<MODE>::kContentBits;
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((o << CB) + content) << BITS) + m
    if (idx >= ((size_t)ModeTraits<MODE>::kOffsets << (CB + BITS))) return;
    uint32_t m = idx & ((1u << BITS) - 1);
    uint32_t content = (idx >> BITS) & ((1u << CB) - 1);
    int o = (int)(idx >> (BITS + CB));
    // byte parity inside the column: HGR o is the parity; DHGR poke ignores it
    uint32_t pm = poke_window<MODE>(m, content, o & 1);
    const ulonglong2 *S = strings + ((size_t)o << BITS);
    ulonglong2 a = S[pm], b = S[m];
    out[idx] = (uint16_t)edit_distance<ND>(a.x, (uint32_t)a.y, b.x, (uint32_t)b.y, lut);
}

// Bitmap.edit_distances' load-time mirror (screen.py:352-365): dist[transpose] += dist[identity],
// i.e. new[a][b] = old[a][b] + old[b][a] (u16 arithmetic), for a table as the .npz files hold it
template <int MODE>
__global__ __launch_bounds__(256) void symmetrise_kernel(uint16_t *__restrict__ table)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // (o, a, b) with a > b: one thread per unordered pair
    const size_t per_o = (size_t)1 << (2 * BITS);
    if (idx >= (size_t)ModeTraits<MODE>::kOffsets * per_o) return;
    const size_t o = idx / per_o, ab = idx % per_o;
    const uint32_t a = (uint32_t)(ab >> BITS), b = (uint32_t)(ab & ((1u << BITS) - 1));
    if (a < b) return;
    uint16_t *T = table + o * per_o;
    const uint16_t x = T[((size_t)a << BITS) + b], y = T[((size_t)b << BITS) + a];
    const uint16_t v = (uint16_t)(x + y);
    T[((size_t)a << BITS) + b] = v;   // (a == b: both reads and both writes hit the same cell: x + x, as numpy does)
    T[((size_t)b << BITS) + a] = v;
}

// the store sub-table out of a full symmetric table: S[o][c][m] = table[o][(poke(m, c) << bits) + m]
template <int MODE>
__global__ __launch_bounds__(256) void store_from_table_kernel(const uint16_t *__restrict__ table, uint16_t *__restrict__ out)
{
    constexpr int BITS = ModeTraits<MODE>::kBits, CB = ModeTraits<MODE>::kContentBits;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ((size_t)ModeTraits<MODE>::kOffsets << (CB + BITS))) return;
    const uint32_t m = idx & ((1u << BITS) - 1), c = (idx >> BITS) & ((1u << CB) - 1);
    const int o = (int)(idx >> (BITS + CB));
    const uint32_t pm = poke_window<MODE>(m, c, o & 1);
    out[idx] = table[((size_t)o << (2 * BITS)) + ((size_t)pm << BITS) + m];
}

// ------------------------------------------------------------------ split store table
// (see iiv_stream.h).  One thread per entry of either half: build a representative window
// from the entry's row bits (all other bits zero -- they cannot reach this half's pixels),
// poke the entry's content bits into it, and run the recurrence over this half's pixels:
// LEFT  = (E[M-1], E[M]) after pixels 1..M;
// RIGHT = cost of finishing pixels M+1..N from state (0, inf) and from state (inf, 0).

__device__ static inline uint32_t pdep32(uint32_t v, uint32_t mask)
{
    uint32_t out = 0;
    for (int b = 0, k = 0; b < 16; b++)
        if ((mask >> b) & 1u) out |= ((v >> k++) & 1u) << b;
    return out;
}

__device__ static inline uint32_t string_pixel(const ulonglong2 &s, int k)
{
    return k < 16 ? (uint32_t)(s.x >> (4 * k)) & 0xfu : (uint32_t)(s.y >> (4 * (k - 16))) & 0xfu;
}

template <int MODE>
__global__ __launch_bounds__(256) void split_kernel(const ulonglong2 *__restrict__ strings,
                                                    const uint16_t *__restrict__ sub, uint32_t *__restrict__ left,
                                                    uint32_t *__restrict__ right)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots;
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    const size_t nl = split_left_entries<MODE>(), nr = split_right_entries<MODE>();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool is_left = idx < nl;
    if (!is_left) idx -= nl;
    if (!is_left && idx >= nr) return;
    const int rb = is_left ? T::kLeftRowBits : T::kRightRowBits, cb = is_left ? T::kLeftCBits : T::kRightCBits;
    const uint32_t row = idx & ((1u << rb) - 1), cpart = (idx >> rb) & ((1u << cb) - 1);
    const int o = (int)(idx >> (rb + cb));
    const uint32_t mask = is_left ? split_mask_left<MODE>(o) : split_mask_right<MODE>(o);
    const uint32_t own = mask & split_mask_own<MODE>();
    const uint32_t wt = pdep32(row, mask);
    const uint32_t ws = (wt & ~own) | pdep32(cpart, own);
    const ulonglong2 a = strings[((size_t)o << BITS) + ws], b = strings[((size_t)o << BITS) + wt];
    auto step = [&](int k, uint32_t &e1, uint32_t &e2) {   // pixel k (0-based) of both strings
        const uint32_t ak = string_pixel(a, k), bk = string_pixel(b, k);
        uint32_t e = e1 + lut[ak * 16 + bk];
        if (k >= 1) {
            const uint32_t ap = string_pixel(a, k - 1), bp = string_pixel(b, k - 1);
            if (ap == bk && ak == bp && e2 + 1 < e) e = e2 + 1;
        }
        e = e < kSplitInf ? e : kSplitInf;
        e2 = e1;
        e1 = e;
    };
    if (is_left) {
        uint32_t e1 = 0, e2 = kSplitInf;
        for (int k = 0; k < T::kCut; k++) step(k, e1, e2);
        left[idx] = e2 | (e1 << 16);
    } else {
        uint32_t r[2];
        for (int j = 0; j < 2; j++) {
            uint32_t e2 = j == 0 ? 0u : kSplitInf, e1 = j == 0 ? kSplitInf : 0u;
            for (int k = T::kCut; k < ND; k++) step(k, e1, e2);
            r[j] = e1;
        }
        right[idx] = r[0] | (r[1] << 16);
    }
}

// ------------------------------------------------------------------ narrow form (iiv_stream.h)

// L1 (u16): component 1 of every entry of the left half.  RF (u16): min(r1, r0 - s) + kNarrowBias for every entry of
// the right half, s = the substitution cost of the last pixel of the LEFT half (pixel kCut - 1, which lies inside the
// right half's bits): one thread per entry, same representative windows and recurrence as split_kernel.
template <int MODE>
__global__ __launch_bounds__(256) void narrow_fold_kernel(const ulonglong2 *__restrict__ strings, const uint16_t *__restrict__ sub,
                                                          const uint32_t *__restrict__ left, uint8_t *__restrict__ buf)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots;
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    const size_t nl = split_left_entries<MODE>(), nr = split_right_entries<MODE>();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < nl) {
        reinterpret_cast<uint16_t *>(buf)[idx] = (uint16_t)(left[idx] >> 16);
        return;
    }
    idx -= nl;
    if (idx >= nr) return;
    const uint32_t row = idx & ((1u << T::kRightRowBits) - 1), cpart = (idx >> T::kRightRowBits) & ((1u << T::kRightCBits) - 1);
    const int o = (int)(idx >> (T::kRightRowBits + T::kRightCBits));
    const uint32_t mask = split_mask_right<MODE>(o), own = mask & split_mask_own<MODE>();
    const uint32_t wt = pdep32(row, mask);
    const uint32_t ws = (wt & ~own) | pdep32(cpart, own);
    const ulonglong2 a = strings[((size_t)o << BITS) + ws], b = strings[((size_t)o << BITS) + wt];
    uint32_t r[2];
    for (int j = 0; j < 2; j++) {
        uint32_t e2 = j == 0 ? 0u : kSplitInf, e1 = j == 0 ? kSplitInf : 0u;
        for (int k = T::kCut; k < ND; k++) {
            const uint32_t ak = string_pixel(a, k), bk = string_pixel(b, k);
            uint32_t e = e1 + lut[ak * 16 + bk];
            const uint32_t ap = string_pixel(a, k - 1), bp = string_pixel(b, k - 1);
            if (ap == bk && ak == bp && e2 + 1 < e) e = e2 + 1;
            e = e < kSplitInf ? e : kSplitInf;
            e2 = e1;
            e1 = e;
        }
        r[j] = e1;
    }
    const uint32_t s = lut[string_pixel(a, T::kCut - 1) * 16 + string_pixel(b, T::kCut - 1)];
    int v = (int)r[1];
    if (r[0] < kSplitInf && (int)r[0] - (int)s < v) v = (int)r[0] - (int)s;   // path 0: the transposition across the cut
    reinterpret_cast<uint16_t *>(buf + narrow_right_off<MODE>())[idx] = (uint16_t)(v + (int)kNarrowBias);
}

// every value as the kernels obtain it (wd word -> two offsets -> two u16 -> sum - bias), for the exactness check at
// encoder creation and in the tests; counts the entries that differ from the dense store table
template <int MODE>
__global__ __launch_bounds__(256) void narrow_expand_kernel(NarrowTables nt, const uint16_t *__restrict__ dense, uint16_t *__restrict__ out,
                                                            unsigned long long *__restrict__ n_mismatch)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits, CB = ModeTraits<MODE>::kContentBits;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ((size_t)ModeTraits<MODE>::kOffsets << (CB + BITS))) return;
    const uint32_t m = idx & ((1u << BITS) - 1), c = (idx >> BITS) & ((1u << CB) - 1);
    const int o = (int)(idx >> (BITS + CB)), odd = o & 1;
    const uint32_t wd = wd_word(split_row_left<MODE>(m, odd), split_row_right<MODE>(m, odd), 0);
    const uint32_t slab_l = (uint32_t)((((size_t)o << T::kLeftCBits) + split_content_left<MODE>(c, odd)) << (T::kLeftRowBits + 1));
    const uint32_t slab_r =
        nt.right_off + (uint32_t)((((size_t)o << T::kRightCBits) + split_content_right<MODE>(c, odd)) << (T::kRightRowBits + 1));
    const uint32_t v = (uint32_t)*reinterpret_cast<const uint16_t *>(nt.base + slab_l + wd_off_left(wd)) +
                       (uint32_t)*reinterpret_cast<const uint16_t *>(nt.base + slab_r + wd_off_right(wd)) - kNarrowBias;
    if (out) out[idx] = (uint16_t)v;
    if (dense && v != (uint32_t)dense[idx]) atomicAdd(n_mismatch, 1ull);
}

int expand_narrow_tables(int mode, const NarrowTables &nt, const uint16_t *d_store, uint16_t *d_out, unsigned long long *n_mismatch,
                         hipStream_t st)
{
    const size_t n = (size_t)num_offsets(mode) << (content_bits(mode) + masked_bits(mode));
    unsigned long long *d_cnt = nullptr;
    IIV_HIP(hipMalloc(&d_cnt, 8));
    int rc = hip_check(hipMemsetAsync(d_cnt, 0, 8, st), "memset");
    if (!rc) {
        const dim3 grid((unsigned)((n + 255) / 256));
        if (mode == kDHGR) hipLaunchKernelGGL(narrow_expand_kernel<kDHGR>, grid, dim3(256), 0, st, nt, d_store, d_out, d_cnt);
        else hipLaunchKernelGGL(narrow_expand_kernel<kHGR>, grid, dim3(256), 0, st, nt, d_store, d_out, d_cnt);
        rc = hip_check(hipGetLastError(), "narrow_expand_kernel launch");
    }
    if (!rc) rc = hip_check(hipMemcpyAsync(n_mismatch, d_cnt, 8, hipMemcpyDeviceToHost, st), "copy count");
    if (!rc) rc = hip_check(hipStreamSynchronize(st), "sync");
    (void)hipFree(d_cnt);
    return rc;
}

// d_strings / d_sub: build_strings; d_left: the u32 left half (build_split_tables); d_store: the dense store table the
// folded form is held to -- out->exact says whether every one of its entries is reproduced
int build_narrow_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, const uint32_t *d_left, const uint16_t *d_store,
                        NarrowTables *out, hipStream_t st)
{
    const size_t total = mode == kDHGR ? narrow_total_bytes<kDHGR>() : narrow_total_bytes<kHGR>();
    const size_t n_fill = split_entries(mode, 0) + split_entries(mode, 1);
    uint8_t *buf = nullptr;
    IIV_HIP(hipMalloc(&buf, total));
    if (mode == kDHGR)
        hipLaunchKernelGGL(narrow_fold_kernel<kDHGR>, dim3((unsigned)((n_fill + 255) / 256)), dim3(256), 0, st, d_strings, d_sub, d_left, buf);
    else
        hipLaunchKernelGGL(narrow_fold_kernel<kHGR>, dim3((unsigned)((n_fill + 255) / 256)), dim3(256), 0, st, d_strings, d_sub, d_left, buf);
    int rc = hip_check(hipGetLastError(), "narrow_fold_kernel launch");
    out->base = buf;
    out->right_off = mode == kDHGR ? narrow_right_off<kDHGR>() : narrow_right_off<kHGR>();
    out->exact = 0;
    unsigned long long bad = 0;
    if (!rc) rc = expand_narrow_tables(mode, *out, d_store, nullptr, &bad, st);
    if (rc) {
        (void)hipFree(buf);
        out->base = nullptr;
        return rc;
    }
    out->exact = bad == 0 ? 1 : 0;
    return IIV_OK;
}

void free_narrow_tables(NarrowTables *nt)
{
    if (nt->base) (void)hipFree(const_cast<uint8_t *>(nt->base));
    nt->base = nullptr;
}

// The halves with content innermost -- T[o][row][content part] -- for the joint content choice
// (iiv_workgroup.hip, greedy_kernel<MODE, true>): one row's values for every byte value are one or
// two cache lines.
template <int MODE>
__global__ __launch_bounds__(256) void split_transpose_kernel(const uint32_t *__restrict__ left,
                                                              const uint32_t *__restrict__ right,
                                                              uint32_t *__restrict__ left_t, uint32_t *__restrict__ right_t)
{
    using T = SplitTraits<MODE>;
    const size_t nl = split_left_entries<MODE>(), nr = split_right_entries<MODE>();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool is_left = idx < nl;
    if (!is_left) idx -= nl;
    if (!is_left && idx >= nr) return;
    const int rb = is_left ? T::kLeftRowBits : T::kRightRowBits, cb = is_left ? T::kLeftCBits : T::kRightCBits;
    const uint32_t row = idx & ((1u << rb) - 1), cpart = (idx >> rb) & ((1u << cb) - 1);
    const size_t o = idx >> (rb + cb);
    (is_left ? left_t : right_t)[(((o << rb) + row) << cb) + cpart] = (is_left ? left : right)[idx];
}

int transpose_split_tables(int mode, const uint32_t *d_left, const uint32_t *d_right, uint32_t *d_left_t,
                           uint32_t *d_right_t, hipStream_t st)
{
    const size_t n = mode == kDHGR ? split_left_entries<kDHGR>() + split_right_entries<kDHGR>()
                                   : split_left_entries<kHGR>() + split_right_entries<kHGR>();
    const dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(split_transpose_kernel<kDHGR>, grid, dim3(256), 0, st, d_left, d_right, d_left_t, d_right_t);
    else
        hipLaunchKernelGGL(split_transpose_kernel<kHGR>, grid, dim3(256), 0, st, d_left, d_right, d_left_t, d_right_t);
    return hip_check(hipGetLastError(), "split_transpose_kernel launch");
}

// The narrow form with content innermost and two byte values per word, for the joint content choice's packed scoring
// (iiv_workgroup.hip, greedy_kernel<MODE, 2>): lane l of a wave scores byte values l + 128 j (low half) and l + 128 j + 64
// (high half), j < kPairs, so
//     JL[o][row][j][l] = L1[o][left part of l + 128 j][row] | L1[o][left part of l + 128 j + 64][row] << 16
// and JR likewise from RF: the values of every byte value for one byte of the page are 2 x kPairs coalesced 256-byte
// loads, and v_pk_add_u16 adds both halves at once.
template <int MODE>
__global__ __launch_bounds__(256) void joint_pack_kernel(const uint8_t *__restrict__ narrow, uint32_t right_off,
                                                         uint32_t *__restrict__ jl, uint32_t *__restrict__ jr)
{
    using T = SplitTraits<MODE>;
    constexpr int NP = JointPack<MODE>::kPairs;
    const size_t nl = joint_left_entries<MODE>(), nr = joint_right_entries<MODE>();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool is_left = idx < nl;
    if (!is_left) idx -= nl;
    if (!is_left && idx >= nr) return;
    const int rb = is_left ? T::kLeftRowBits : T::kRightRowBits, cb = is_left ? T::kLeftCBits : T::kRightCBits;
    const uint32_t l = idx & 63u, j = (uint32_t)(idx >> 6) % NP;
    const size_t orow = (idx >> 6) / NP;
    const uint32_t row = (uint32_t)(orow & ((1u << rb) - 1));
    const int o = (int)(orow >> rb), od = MODE == kDHGR ? o >> 1 : o;   // (byte_offset: the parity of the page offset)
    const uint16_t *half = reinterpret_cast<const uint16_t *>(narrow + (is_left ? 0u : right_off));
    uint32_t v = 0;
    for (int h = 0; h < 2; h++) {
        const uint32_t c = l + 128u * j + 64u * (uint32_t)h;
        const uint32_t part = is_left ? split_content_left<MODE>(c, od) : split_content_right<MODE>(c, od);
        v |= (uint32_t)half[((((size_t)o << cb) + part) << rb) + row] << (16 * h);
    }
    (is_left ? jl : jr)[idx] = v;
}

int build_joint_tables(int mode, const NarrowTables &nt, uint32_t **d_jl, uint32_t **d_jr, hipStream_t st)
{
    const size_t nl = mode == kDHGR ? joint_left_entries<kDHGR>() : joint_left_entries<kHGR>();
    const size_t nr = mode == kDHGR ? joint_right_entries<kDHGR>() : joint_right_entries<kHGR>();
    IIV_HIP(hipMalloc(d_jl, nl * 4));
    IIV_HIP(hipMalloc(d_jr, nr * 4));
    const dim3 grid((unsigned)((nl + nr + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(joint_pack_kernel<kDHGR>, grid, dim3(256), 0, st, nt.base, nt.right_off, *d_jl, *d_jr);
    else
        hipLaunchKernelGGL(joint_pack_kernel<kHGR>, grid, dim3(256), 0, st, nt.base, nt.right_off, *d_jl, *d_jr);
    return hip_check(hipGetLastError(), "joint_pack_kernel launch");
}

// The halves of the diff-weight table (iiv_stream.h): the same recurrence between two arbitrary
// windows, each built from its row part alone.
template <int MODE>
__global__ __launch_bounds__(256) void split_dw_kernel(const ulonglong2 *__restrict__ strings,
                                                       const uint16_t *__restrict__ sub, uint32_t *__restrict__ left,
                                                       uint32_t *__restrict__ right)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits, ND = ModeTraits<MODE>::kDots;
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    const size_t nl = split_dw_left_entries<MODE>(), nr = split_dw_right_entries<MODE>();
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool is_left = idx < nl;
    if (!is_left) idx -= nl;
    if (!is_left && idx >= nr) return;
    const int rb = is_left ? T::kLeftRowBits : T::kRightRowBits;
    const uint32_t row_t = idx & ((1u << rb) - 1), row_s = (idx >> rb) & ((1u << rb) - 1);
    const int o = (int)(idx >> (2 * rb));
    const uint32_t mask = is_left ? split_mask_left<MODE>(o) : split_mask_right<MODE>(o);
    const ulonglong2 a = strings[((size_t)o << BITS) + pdep32(row_s, mask)],
                     b = strings[((size_t)o << BITS) + pdep32(row_t, mask)];
    auto step = [&](int k, uint32_t &e1, uint32_t &e2) {
        const uint32_t ak = string_pixel(a, k), bk = string_pixel(b, k);
        uint32_t e = e1 + lut[ak * 16 + bk];
        if (k >= 1) {
            const uint32_t ap = string_pixel(a, k - 1), bp = string_pixel(b, k - 1);
            if (ap == bk && ak == bp && e2 + 1 < e) e = e2 + 1;
        }
        e = e < kSplitInf ? e : kSplitInf;
        e2 = e1;
        e1 = e;
    };
    if (is_left) {
        uint32_t e1 = 0, e2 = kSplitInf;
        for (int k = 0; k < T::kCut; k++) step(k, e1, e2);
        left[idx] = e2 | (e1 << 16);
    } else {
        uint32_t r[2];
        for (int j = 0; j < 2; j++) {
            uint32_t e2 = j == 0 ? 0u : kSplitInf, e1 = j == 0 ? kSplitInf : 0u;
            for (int k = T::kCut; k < ND; k++) step(k, e1, e2);
            r[j] = e1;
        }
        right[idx] = r[0] | (r[1] << 16);
    }
}

// the dense store table rebuilt from the two halves with the index arithmetic the encoder
// kernels use (tests compare it with store_kernel's output, entry for entry)
template <int MODE>
__global__ __launch_bounds__(256) void split_expand_kernel(const uint32_t *__restrict__ left,
                                                           const uint32_t *__restrict__ right, uint16_t *__restrict__ out)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits, CB = ModeTraits<MODE>::kContentBits;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((o << CB) + content) << BITS) + m
    if (idx >= ((size_t)ModeTraits<MODE>::kOffsets << (CB + BITS))) return;
    const uint32_t m = idx & ((1u << BITS) - 1), c = (idx >> BITS) & ((1u << CB) - 1);
    const int o = (int)(idx >> (BITS + CB)), odd = o & 1;
    const uint32_t l = left[(((size_t)o << T::kLeftCBits) + split_content_left<MODE>(c, odd) << T::kLeftRowBits) +
                            split_row_left<MODE>(m, odd)];
    const uint32_t r = right[(((size_t)o << T::kRightCBits) + split_content_right<MODE>(c, odd) << T::kRightRowBits) +
                             split_row_right<MODE>(m, odd)];
    out[idx] = (uint16_t)split_combine(l, r);
}

// ------------------------------------------------------------------ host side

// compute_substitute_costs (make_data_tables.py:73-89): the fill loop writes
// (c,d) and (d,c) on every iteration, so the result is dm's lower triangle
// mirrored: sub[u][v] = dm[max(u,v)][min(u,v)].
static void substitute_costs(const int32_t dm[256], uint16_t sub[256])
{
    int32_t s[256];
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            s[i * 16 + j] = dm[i * 16 + j];
            s[j * 16 + i] = dm[i * 16 + j];
        }
    for (int i = 0; i < 256; i++) sub[i] = (uint16_t)s[i];
}

template <int MODE> static int pixel_strings_impl(uint32_t *d_dots, uint8_t *d_pixels, ulonglong2 *d_strings, hipStream_t st)
{
    int n = ModeTraits<MODE>::kOffsets << ModeTraits<MODE>::kBits;
    hipLaunchKernelGGL(pixel_kernel<MODE>, dim3((n + 255) / 256), dim3(256), 0, st, d_dots, d_pixels, d_strings);
    return hip_check(hipGetLastError(), "pixel_kernel launch");
}

int pixel_strings(int mode, uint32_t *d_dots, uint8_t *d_pixels, ulonglong2 *d_strings, hipStream_t st)
{
    return mode == kDHGR ? pixel_strings_impl<kDHGR>(d_dots, d_pixels, d_strings, st)
                         : pixel_strings_impl<kHGR>(d_dots, d_pixels, d_strings, st);
}

// The HGR prologue's dot lookups (iiv_edit.h: hgr_dot_slot_lo / _hi): entry t = to_dots of a window that has slot t's bits
// and no others, shifted up by one.
__global__ __launch_bounds__(64) void hgr_dot_lut_kernel(uint32_t *__restrict__ out)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= kHgrDotLutEntries) return;
    int odd;
    const uint32_t m = hgr_dot_slot_window(t, odd);
    out[t] = to_dots<kHGR>(m, odd) << 1;
}
// ... and that OR-ing the two parts gives to_dots for EVERY window of both parities (counted, not assumed)
__global__ __launch_bounds__(256) void hgr_dot_lut_check_kernel(const uint32_t *__restrict__ lut, unsigned long long *__restrict__ mismatches)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;   // odd << 14 | window
    if (idx >= (2 << 14)) return;
    const int odd = idx >> 14;
    const uint32_t m = (uint32_t)idx & 0x3fffu;
    if ((lut[hgr_dot_slot_lo(m, odd)] | lut[hgr_dot_slot_hi(m, odd)]) != (to_dots<kHGR>(m, odd) << 1)) atomicAdd(mismatches, 1ull);
}

int build_hgr_dot_lut(uint32_t **d_out, hipStream_t st)
{
    IIV_HIP(hipMalloc(d_out, kHgrDotLutEntries * sizeof(uint32_t)));
    hipLaunchKernelGGL(hgr_dot_lut_kernel, dim3((kHgrDotLutEntries + 63) / 64), dim3(64), 0, st, *d_out);
    return hip_check(hipGetLastError(), "hgr_dot_lut_kernel launch");
}

// ------------------------------------------------------------------ diff weights as a sum of local terms (the prologue's default)
// The recurrence E[k] = min(E[k-1] + s_k, E[k-2] + 1 if pixels (k-1, k) transpose) (make_data_tables.py:92-108; iiv_edit.h)
// collapses to a SUM for the strings it is applied to.  A colour string is a sliding 4-dot window, so a_{k-1}, a_k, a_{k+1}
// cannot be X, Y, X with X != Y (the step k-1 -> k replaces the dot of one position class, the step k -> k+1 that of the
// next): two transpositions never overlap, E[k-1] = E[k-2] + s_{k-1} wherever pixels (k-1, k) transpose, and
//     distance = sum over k of g_k,    g_k = s_k, or min(s_k, 1 - s_{k-1}) where pixels (k-1, k) transpose.
// g_k depends on dots k-1 .. k+3 of both windows, so two pixels' terms are one lookup by 6 + 6 dots -- five lookups and
// five additions per DHGR distance instead of ten dependent steps of the recurrence, nine per HGR distance instead of
// eighteen (round 6: HGR's windows are turned into their 21 dots first, two small lookups each -- iiv_edit.h:
// hgr_dot_slot_lo -- and from there on a dot is a dot: the SAME table form serves both modes).  (Checked against the
// recurrence for every pair of windows: iiv_check_diff_weight_pieces / tests/test_gpu_tables.py, both modes, both palettes
// and random matrices.)
// Table: G[bank][cur6 << 6 | tgt6] = (g-sum of the pixel pair with rotation ph_e) | (... with rotation ph_e + 2) << 16, each
// plus kDwPieceBias; ph_e = the phase of the bank's even bytes (its odd bytes' differs by 2: DHGR [1, 0, 3, 2], HGR [1, 3]),
// bit j of cur6 / tgt6 = dot k - 1 + j, the pair being pixels (k, k + 1).  A window's dot -1 is 0 in both strings, which makes
// the term a pixel -1 would contribute vanish (a transposition of pixels (-1, 0) then needs equal colours all round).
// DHGR: two banks (aux, then main), G[bank][cur6 << 6 | tgt6]; HGR: one, G[cur6 * kHgrPieceStride + tgt6] (iiv_stream.h).
__global__ __launch_bounds__(256) void dw_piece_kernel(int mode, const uint16_t *__restrict__ sub, uint32_t *__restrict__ out)
{
    __shared__ uint16_t lut[256];
    load_cost_lut(lut, sub, threadIdx.x);
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;   // bank << 12 | cur6 << 6 | tgt6
    if (idx >= (mode == kDHGR ? 2 : 1) * 4096) return;
    const int bank = idx >> 12;
    const uint32_t cur6 = (idx >> 6) & 63u, tgt6 = idx & 63u;
    const int ph_e = mode == kDHGR ? phase_of(kDHGR, byte_offset<kDHGR>(0, bank)) : phase_of(kHGR, 0);
    uint32_t v = 0;
    for (int c = 0; c < 2; c++) {
        const int r = (ph_e + 2 * c) & 3;   // rotation of pixel k; k - 1 has r - 1, k + 1 has r + 1 (colours.py:100-134)
        int a[3], b[3];
        for (int j = 0; j < 3; j++) {
            const int rot = (r - 1 + j) & 3;
            const uint32_t wa = (cur6 >> j) & 0xfu, wb = (tgt6 >> j) & 0xfu;
            a[j] = (int)(((wa | (wa << 4)) >> (4 - rot)) & 0xfu);
            b[j] = (int)(((wb | (wb << 4)) >> (4 - rot)) & 0xfu);
        }
        int s[3];
        for (int j = 0; j < 3; j++) s[j] = (int)lut[a[j] * 16 + b[j]];
        int g = 0;
        for (int j = 1; j < 3; j++) {
            const bool t = a[j - 1] == b[j] && a[j] == b[j - 1];
            g += (t && 1 - s[j - 1] < s[j]) ? 1 - s[j - 1] : s[j];
        }
        v |= (uint32_t)(g + (int)kDwPieceBias) << (16 * c);
    }
    // (HGR: rows of kHgrPieceStride words, iiv_stream.h -- the LDS banks of its doubled dots; the gaps stay zero)
    out[mode == kDHGR ? (uint32_t)idx : cur6 * kHgrPieceStride + tgt6] = v;
}

int build_dw_piece_table(int mode, const uint16_t *d_sub, uint32_t **d_out, hipStream_t st)
{
    const int n = (mode == kDHGR ? 2 : 1) * 4096;
    const size_t words = mode == kDHGR ? (size_t)n : (size_t)kHgrPieceWords;
    IIV_HIP(hipMalloc(d_out, words * sizeof(uint32_t)));
    IIV_HIP(hipMemsetAsync(*d_out, 0, words * sizeof(uint32_t), st));
    hipLaunchKernelGGL(dw_piece_kernel, dim3(n / 256), dim3(256), 0, st, mode, d_sub, *d_out);
    return hip_check(hipGetLastError(), "dw_piece_kernel launch");
}

// every entry of the full symmetric DHGR table against the sum of its five pair terms, read the way the prologue reads them
__global__ __launch_bounds__(256) void dw_piece_check_kernel(const uint32_t *__restrict__ pieces, const uint16_t *__restrict__ table,
                                                             unsigned long long *__restrict__ mismatches)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // ((o << 13) + current) << 13) + target
    if (idx >= ((size_t)4 << 26)) return;
    const uint32_t tm = idx & 8191u, cm = (idx >> 13) & 8191u;
    const int o = (int)(idx >> 26), parity = o >> 1, is_aux = (o & 1) ? 0 : 1;   // byte_offset<kDHGR>: aux 0 / 2, main 1 / 3
    const uint32_t W = (cm << 17) | (tm << 1);
    const unsigned char *g = reinterpret_cast<const unsigned char *>(pieces + 4096 * is_aux);
    uint32_t acc = 0;
    for (int i = 0; i < 5; i++) {
        const uint32_t cur6 = (W >> (16 + 2 * i)) & 63u, tgt6 = ((W << 2) >> (2 * i)) & 0xfcu;   // (as the prologue forms them)
        const uint32_t v = *reinterpret_cast<const uint32_t *>(g + ((cur6 << 8) | tgt6));
        acc += ((i + parity) & 1) ? v >> 16 : v & 0xffffu;
    }
    if (acc - 5u * kDwPieceBias != (uint32_t)table[idx]) atomicAdd(mismatches, 1ull);
}

// every entry of the full symmetric HGR table (2 x 2^28) against the sum of its nine pair terms: the prologue's own function
// (iiv_stream.h: hgr_dw_pieces_sum) on the tables where they lie
__global__ __launch_bounds__(256) void dw_piece_check_hgr_kernel(const uint32_t *__restrict__ pieces, const uint32_t *__restrict__ dots,
                                                                 const uint16_t *__restrict__ table, unsigned long long *__restrict__ mismatches)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // ((o << 14) + current) << 14) + target
    if (idx >= ((size_t)2 << 28)) return;
    const uint32_t tm = idx & 16383u, cm = (idx >> 14) & 16383u;
    const int odd = (int)(idx >> 28);
    if (hgr_dw_pieces_sum(dots, reinterpret_cast<const unsigned char *>(pieces), cm, tm, odd) != (uint32_t)table[idx]) atomicAdd(mismatches, 1ull);
}

int check_dw_piece_table(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches, hipStream_t st)
{
    uint16_t sub[256], *d_sub = nullptr;
    uint32_t *d_p = nullptr, *d_dots = nullptr;
    unsigned long long *d_cnt = nullptr;
    substitute_costs(dm, sub);
    int rc = IIV_OK;
    do {
        if ((rc = hip_check(hipMalloc(&d_sub, sizeof(sub)), "hipMalloc(sub)"))) break;
        if ((rc = hip_check(hipMemcpy(d_sub, sub, sizeof(sub), hipMemcpyHostToDevice), "copy sub"))) break;
        if ((rc = hip_check(hipMalloc(&d_cnt, 8), "hipMalloc(count)"))) break;
        if ((rc = hip_check(hipMemsetAsync(d_cnt, 0, 8, st), "memset"))) break;
        if ((rc = build_dw_piece_table(mode, d_sub, &d_p, st))) break;
        if (mode == kDHGR) {
            hipLaunchKernelGGL(dw_piece_check_kernel, dim3((unsigned)(((size_t)4 << 26) / 256)), dim3(256), 0, st, d_p, d_table, d_cnt);
        } else {
            if ((rc = build_hgr_dot_lut(&d_dots, st))) break;
            hipLaunchKernelGGL(hgr_dot_lut_check_kernel, dim3((2 << 14) / 256), dim3(256), 0, st, d_dots, d_cnt);
            hipLaunchKernelGGL(dw_piece_check_hgr_kernel, dim3((unsigned)(((size_t)2 << 28) / 256)), dim3(256), 0, st, d_p, d_dots, d_table, d_cnt);
        }
        if ((rc = hip_check(hipGetLastError(), "dw_piece_check_kernel launch"))) break;
        if ((rc = hip_check(hipMemcpyAsync(mismatches, d_cnt, 8, hipMemcpyDeviceToHost, st), "copy count"))) break;
        rc = hip_check(hipStreamSynchronize(st), "sync");
    } while (0);
    if (d_sub) (void)hipFree(d_sub);
    if (d_p) (void)hipFree(d_p);
    if (d_dots) (void)hipFree(d_dots);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

struct TableScratch {
    ulonglong2 *strings = nullptr;
    uint16_t *sub = nullptr;
    ~TableScratch()
    {
        if (strings) (void)hipFree(strings);
        if (sub) (void)hipFree(sub);
    }
};

static int prepare_scratch(int mode, const int32_t dm[256], TableScratch &sc, hipStream_t st)
{
    size_t n = (size_t)num_offsets(mode) << masked_bits(mode);
    IIV_HIP(hipMalloc(&sc.strings, n * sizeof(ulonglong2)));
    IIV_HIP(hipMalloc(&sc.sub, 256 * sizeof(uint16_t)));
    uint16_t sub[256];
    substitute_costs(dm, sub);
    IIV_HIP(hipMemcpyAsync(sc.sub, sub, sizeof(sub), hipMemcpyHostToDevice, st));
    IIV_HIP(hipStreamSynchronize(st));  // `sub` is a stack buffer
    return pixel_strings(mode, nullptr, nullptr, sc.strings, st);
}

int build_strings(int mode, const int32_t dm[256], ulonglong2 **d_strings, uint16_t **d_sub, hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    IIV_HIP(hipStreamSynchronize(st));
    *d_strings = sc.strings;
    *d_sub = sc.sub;
    sc.strings = nullptr;  // ownership passes to the caller
    sc.sub = nullptr;
    return IIV_OK;
}

int build_table(int mode, const int32_t dm[256], uint16_t *d_out, int symmetric, hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    int bits = masked_bits(mode);
    dim3 grid((1u << bits) / (256 * kTableColsPerThread), (1u << bits) / kTableRowsPerBlock, num_offsets(mode));
    if (mode == kDHGR)
        hipLaunchKernelGGL(table_kernel<kDHGR>, grid, dim3(256), 0, st, sc.strings, sc.sub, d_out, symmetric);
    else
        hipLaunchKernelGGL(table_kernel<kHGR>, grid, dim3(256), 0, st, sc.strings, sc.sub, d_out, symmetric);
    rc = hip_check(hipGetLastError(), "table_kernel launch");
    if (rc) return rc;
    IIV_HIP(hipStreamSynchronize(st));  // scratch is freed on return
    return IIV_OK;
}

int build_store_table(int mode, const int32_t dm[256], uint16_t *d_out, hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    size_t n = (size_t)num_offsets(mode) << (content_bits(mode) + masked_bits(mode));
    dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(store_kernel<kDHGR>, grid, dim3(256), 0, st, sc.strings, sc.sub, d_out);
    else
        hipLaunchKernelGGL(store_kernel<kHGR>, grid, dim3(256), 0, st, sc.strings, sc.sub, d_out);
    rc = hip_check(hipGetLastError(), "store_kernel launch");
    if (rc) return rc;
    IIV_HIP(hipStreamSynchronize(st));
    return IIV_OK;
}

// left / right: device buffers of split_left_entries / split_right_entries u32 (iiv_stream.h)
int build_split_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, uint32_t *d_left, uint32_t *d_right,
                       hipStream_t st)
{
    const size_t n = mode == kDHGR ? split_left_entries<kDHGR>() + split_right_entries<kDHGR>()
                                   : split_left_entries<kHGR>() + split_right_entries<kHGR>();
    dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(split_kernel<kDHGR>, grid, dim3(256), 0, st, d_strings, d_sub, d_left, d_right);
    else
        hipLaunchKernelGGL(split_kernel<kHGR>, grid, dim3(256), 0, st, d_strings, d_sub, d_left, d_right);
    return hip_check(hipGetLastError(), "split_kernel launch");
}

size_t split_dw_entries(int mode, int right)
{
    if (mode == kDHGR) return right ? split_dw_right_entries<kDHGR>() : split_dw_left_entries<kDHGR>();
    return right ? split_dw_right_entries<kHGR>() : split_dw_left_entries<kHGR>();
}

int build_split_dw_tables(int mode, const ulonglong2 *d_strings, const uint16_t *d_sub, uint32_t *d_left, uint32_t *d_right,
                          hipStream_t st)
{
    const size_t n = split_dw_entries(mode, 0) + split_dw_entries(mode, 1);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR)
        hipLaunchKernelGGL(split_dw_kernel<kDHGR>, grid, dim3(256), 0, st, d_strings, d_sub, d_left, d_right);
    else
        hipLaunchKernelGGL(split_dw_kernel<kHGR>, grid, dim3(256), 0, st, d_strings, d_sub, d_left, d_right);
    return hip_check(hipGetLastError(), "split_dw_kernel launch");
}

// every entry of the full symmetric table against the combination of the two halves
template <int MODE>
__global__ __launch_bounds__(256) void split_dw_check_kernel(const uint32_t *__restrict__ left,
                                                             const uint32_t *__restrict__ right,
                                                             const uint16_t *__restrict__ table,
                                                             unsigned long long *__restrict__ mismatches)
{
    using T = SplitTraits<MODE>;
    constexpr int BITS = ModeTraits<MODE>::kBits;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // ((o << BITS) + source) << BITS) + target
    if (idx >= ((size_t)ModeTraits<MODE>::kOffsets << (2 * BITS))) return;
    const uint32_t t = idx & ((1u << BITS) - 1), s = (idx >> BITS) & ((1u << BITS) - 1);
    const int o = (int)(idx >> (2 * BITS)), odd = o & 1;
    const uint32_t l = left[(((size_t)o << T::kLeftRowBits) + split_row_left<MODE>(s, odd) << T::kLeftRowBits) +
                            split_row_left<MODE>(t, odd)];
    const uint32_t r = right[(((size_t)o << T::kRightRowBits) + split_row_right<MODE>(s, odd) << T::kRightRowBits) +
                             split_row_right<MODE>(t, odd)];
    if (split_combine(l, r) != table[idx]) atomicAdd(mismatches, 1ull);
}

int check_split_dw_table(int mode, const int32_t dm[256], const uint16_t *d_table, unsigned long long *mismatches,
                         hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    uint32_t *d_l = nullptr, *d_r = nullptr;
    unsigned long long *d_cnt = nullptr;
    do {
        if ((rc = hip_check(hipMalloc(&d_l, split_dw_entries(mode, 0) * 4), "hipMalloc(dw left)"))) break;
        if ((rc = hip_check(hipMalloc(&d_r, split_dw_entries(mode, 1) * 4), "hipMalloc(dw right)"))) break;
        if ((rc = hip_check(hipMalloc(&d_cnt, 8), "hipMalloc(count)"))) break;
        if ((rc = hip_check(hipMemsetAsync(d_cnt, 0, 8, st), "memset"))) break;
        if ((rc = build_split_dw_tables(mode, sc.strings, sc.sub, d_l, d_r, st))) break;
        const size_t n = (size_t)num_offsets(mode) << (2 * masked_bits(mode));
        const dim3 grid((unsigned)((n + 255) / 256));
        if (mode == kDHGR)
            hipLaunchKernelGGL(split_dw_check_kernel<kDHGR>, grid, dim3(256), 0, st, d_l, d_r, d_table, d_cnt);
        else
            hipLaunchKernelGGL(split_dw_check_kernel<kHGR>, grid, dim3(256), 0, st, d_l, d_r, d_table, d_cnt);
        if ((rc = hip_check(hipGetLastError(), "split_dw_check_kernel launch"))) break;
        if ((rc = hip_check(hipMemcpyAsync(mismatches, d_cnt, 8, hipMemcpyDeviceToHost, st), "copy count"))) break;
        rc = hip_check(hipStreamSynchronize(st), "sync");
    } while (0);
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

size_t split_entries(int mode, int right)
{
    if (mode == kDHGR) return right ? split_right_entries<kDHGR>() : split_left_entries<kDHGR>();
    return right ? split_right_entries<kHGR>() : split_left_entries<kHGR>();
}

int build_split_store_table(int mode, const int32_t dm[256], uint32_t *d_left, uint32_t *d_right, uint16_t *d_expanded,
                            hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    uint32_t *tmp_l = nullptr, *tmp_r = nullptr;
    if (!d_left) { IIV_HIP(hipMalloc(&tmp_l, split_entries(mode, 0) * 4)); d_left = tmp_l; }
    if (!d_right) {
        hipError_t he = hipMalloc(&tmp_r, split_entries(mode, 1) * 4);
        if (he != hipSuccess) { if (tmp_l) (void)hipFree(tmp_l); return hip_check(he, "hipMalloc(split right)"); }
        d_right = tmp_r;
    }
    rc = build_split_tables(mode, sc.strings, sc.sub, d_left, d_right, st);
    if (!rc && d_expanded) {
        size_t n = (size_t)num_offsets(mode) << (content_bits(mode) + masked_bits(mode));
        dim3 grid((unsigned)((n + 255) / 256));
        if (mode == kDHGR)
            hipLaunchKernelGGL(split_expand_kernel<kDHGR>, grid, dim3(256), 0, st, d_left, d_right, d_expanded);
        else
            hipLaunchKernelGGL(split_expand_kernel<kHGR>, grid, dim3(256), 0, st, d_left, d_right, d_expanded);
        rc = hip_check(hipGetLastError(), "split_expand_kernel launch");
    }
    hipError_t he = hipStreamSynchronize(st);  // scratch is freed on return
    if (tmp_l) (void)hipFree(tmp_l);
    if (tmp_r) (void)hipFree(tmp_r);
    if (rc) return rc;
    return hip_check(he, "sync");
}

// for the exactness test: the left half and the folded narrow form from dm, every value re-read the way the kernels
// read it into d_expanded; *n_mismatch = entries that differ from d_store
int build_narrow_store_table(int mode, const int32_t dm[256], const uint16_t *d_store, uint16_t *d_expanded,
                             unsigned long long *n_mismatch, hipStream_t st)
{
    TableScratch sc;
    int rc = prepare_scratch(mode, dm, sc, st);
    if (rc) return rc;
    uint32_t *d_l = nullptr, *d_r = nullptr;
    NarrowTables nt{};
    do {
        if ((rc = hip_check(hipMalloc(&d_l, split_entries(mode, 0) * 4), "hipMalloc(split left)"))) break;
        if ((rc = hip_check(hipMalloc(&d_r, split_entries(mode, 1) * 4), "hipMalloc(split right)"))) break;
        if ((rc = build_split_tables(mode, sc.strings, sc.sub, d_l, d_r, st))) break;
        if ((rc = build_narrow_tables(mode, sc.strings, sc.sub, d_l, d_store, &nt, st))) break;
        rc = expand_narrow_tables(mode, nt, d_store, d_expanded, n_mismatch, st);
    } while (0);
    free_narrow_tables(&nt);
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    return rc;
}

int symmetrise_table(int mode, uint16_t *d_table, hipStream_t st)
{
    const size_t n = (size_t)num_offsets(mode) << (2 * masked_bits(mode));
    dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR) hipLaunchKernelGGL(symmetrise_kernel<kDHGR>, grid, dim3(256), 0, st, d_table);
    else hipLaunchKernelGGL(symmetrise_kernel<kHGR>, grid, dim3(256), 0, st, d_table);
    return hip_check(hipGetLastError(), "symmetrise_kernel launch");
}

int store_table_from_table(int mode, const uint16_t *d_table, uint16_t *d_store, hipStream_t st)
{
    const size_t n = (size_t)num_offsets(mode) << (content_bits(mode) + masked_bits(mode));
    dim3 grid((unsigned)((n + 255) / 256));
    if (mode == kDHGR) hipLaunchKernelGGL(store_from_table_kernel<kDHGR>, grid, dim3(256), 0, st, d_table, d_store);
    else hipLaunchKernelGGL(store_from_table_kernel<kHGR>, grid, dim3(256), 0, st, d_table, d_store);
    return hip_check(hipGetLastError(), "store_from_table_kernel launch");
}

int delta_e_pairs(int n, const double *lab1, const double *lab2, double *out, hipStream_t st)
{
    double *d = nullptr;
    IIV_HIP(hipMalloc(&d, (size_t)n * 7 * sizeof(double)));
    int rc = hip_check(hipMemcpyAsync(d, lab1, (size_t)n * 24, hipMemcpyHostToDevice, st), "copy lab1");
    if (!rc) rc = hip_check(hipMemcpyAsync(d + 3 * (size_t)n, lab2, (size_t)n * 24, hipMemcpyHostToDevice, st), "copy lab2");
    if (!rc) {
        hipLaunchKernelGGL(delta_e_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, d, d + 3 * (size_t)n, d + 6 * (size_t)n);
        rc = hip_check(hipGetLastError(), "delta_e_kernel launch");
    }
    if (!rc) rc = hip_check(hipMemcpyAsync(out, d + 6 * (size_t)n, (size_t)n * 8, hipMemcpyDeviceToHost, st), "copy out");
    if (!rc) rc = hip_check(hipStreamSynchronize(st), "sync");
    (void)hipFree(d);
    return rc;
}

int cie2000_matrix(const uint8_t rgb[48], double out_f[256], int32_t out_i[256], hipStream_t st)
{
    uint8_t *d_rgb = nullptr;
    double *d_f = nullptr;
    int32_t *d_i = nullptr;
    IIV_HIP(hipMalloc(&d_rgb, 48));
    IIV_HIP(hipMalloc(&d_f, 256 * sizeof(double)));
    IIV_HIP(hipMalloc(&d_i, 256 * sizeof(int32_t)));
    int rc = IIV_OK;
    do {
        if ((rc = hip_check(hipMemcpyAsync(d_rgb, rgb, 48, hipMemcpyHostToDevice, st), "copy rgb"))) break;
        hipLaunchKernelGGL(cie2000_kernel, dim3(1), dim3(256), 0, st, d_rgb, d_f, d_i);
        if ((rc = hip_check(hipGetLastError(), "cie2000_kernel launch"))) break;
        double f[256];
        int32_t iv[256];
        if ((rc = hip_check(hipMemcpyAsync(f, d_f, sizeof(f), hipMemcpyDeviceToHost, st), "copy f"))) break;
        if ((rc = hip_check(hipMemcpyAsync(iv, d_i, sizeof(iv), hipMemcpyDeviceToHost, st), "copy i"))) break;
        if ((rc = hip_check(hipStreamSynchronize(st), "sync"))) break;
        if (out_f) memcpy(out_f, f, sizeof(f));
        if (out_i) memcpy(out_i, iv, sizeof(iv));
    } while (0);
    (void)hipFree(d_rgb);
    (void)hipFree(d_f);
    (void)hipFree(d_i);
    return rc;
}

}  // namespace iiv
