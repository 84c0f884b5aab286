// Microbenchmark: the memory-side floor of the greedy step's access pattern, with none of its
// arithmetic.  One wave per "stream" (28 resident per CU, like greedy_wave_kernel), per "opcode":
// one coalesced 1 KiB row (streamed from HBM, like the wd row), then eight 4-byte gathers whose
// lane addresses are row-dependent indices into the four (offset, content) slices of a table with
// the split store table's shape (DHGR: left 4 x 32 x 256, right 4 x 64 x 512 u32 = 640 KiB), the
// content changing every opcode.  Reports lookups per second and per CU-cycle; the greedy kernel's
// own rate (512 lookups per opcode) is to be read against this.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_ceiling tools/gather_ceiling.hip && tools/gather_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(64) void pattern_kernel(const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                     const uint4 *__restrict__ rows, int n_ops, int gathers,
                                                     uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t pad[];  // occupancy: the greedy kernel's 5.7 KiB per wave
    const int lane = threadIdx.x;
    const uint4 *my = rows + (size_t)blockIdx.x * n_ops * 64 + lane;
    uint32_t acc = 0, h = blockIdx.x * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;  // wave-uniform content byte
        const uint32_t *le = left + (((0u << 5) | (c & 31u)) << 8), *lo = left + (((2u << 5) | (c & 31u)) << 8);
        const uint32_t *re = right + (((0u << 6) | ((c >> 1) & 63u)) << 9), *ro = right + (((2u << 6) | ((c >> 1) & 63u)) << 9);
        if (gathers) {
            const uint32_t a0 = le[row.x & 0xffu], b0 = re[(row.x >> 8) & 0x1ffu];
            const uint32_t a1 = lo[row.y & 0xffu], b1 = ro[(row.y >> 8) & 0x1ffu];
            const uint32_t a2 = le[row.z & 0xffu], b2 = re[(row.z >> 8) & 0x1ffu];
            const uint32_t a3 = lo[row.w & 0xffu], b3 = ro[(row.w >> 8) & 0x1ffu];
            acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
        } else {
            acc += row.x ^ row.y ^ row.z ^ row.w;
        }
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    sink[blockIdx.x * 64 + lane] = acc;
}

// Variant A: the same eight gathers from a table of half the entry size (u16): half the slice bytes.
__global__ __launch_bounds__(64) void pattern_u16_kernel(const uint16_t *__restrict__ left, const uint16_t *__restrict__ right,
                                                         const uint4 *__restrict__ rows, int n_ops, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x;
    const uint4 *my = rows + (size_t)blockIdx.x * n_ops * 64 + lane;
    uint32_t acc = 0, h = blockIdx.x * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;
        const uint16_t *le = left + (((0u << 5) | (c & 31u)) << 8), *lo = left + (((2u << 5) | (c & 31u)) << 8);
        const uint16_t *re = right + (((0u << 6) | ((c >> 1) & 63u)) << 9), *ro = right + (((2u << 6) | ((c >> 1) & 63u)) << 9);
        const uint32_t a0 = le[row.x & 0xffu], b0 = re[(row.x >> 8) & 0x1ffu];
        const uint32_t a1 = lo[row.y & 0xffu], b1 = ro[(row.y >> 8) & 0x1ffu];
        const uint32_t a2 = le[row.z & 0xffu], b2 = re[(row.z >> 8) & 0x1ffu];
        const uint32_t a3 = lo[row.w & 0xffu], b3 = ro[(row.w >> 8) & 0x1ffu];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    sink[blockIdx.x * 64 + lane] = acc;
}

// Variant D: A plus, for one byte in 64 (the bytes whose colour strings can be transposed across the
// cut), a 2-byte gather from the 8 MiB dense table under an execution mask.
template <bool HGR>
__global__ __launch_bounds__(64) void pattern_u16_rare_kernel(const uint16_t *__restrict__ left, const uint16_t *__restrict__ right,
                                                              const uint16_t *__restrict__ dense, const uint4 *__restrict__ rows,
                                                              int n_ops, uint32_t *__restrict__ sink)
{
    // DHGR: left 4 x 32 x 256, right 4 x 64 x 512, dense 4 x 128 x 2^13; HGR: left / right 2 x 64 x 512, dense 2 x 256 x 2^14
    constexpr uint32_t LR = HGR ? 9 : 8, LM = (1u << LR) - 1, CM = HGR ? 255u : 127u, WB = HGR ? 14 : 13;
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x;
    const uint4 *my = rows + (size_t)blockIdx.x * n_ops * 64 + lane;
    uint32_t acc = 0, h = blockIdx.x * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & CM;
        const uint32_t o_d = HGR ? 1u : 2u, cl = HGR ? (c & 63u) : (c & 31u), cr = HGR ? (c >> 2) : ((c >> 1) & 63u);
        const uint32_t lcb = HGR ? 6 : 5;
        const uint16_t *le = left + (((0u << lcb) | cl) << LR), *lo = left + (((o_d << lcb) | cl) << LR);
        const uint16_t *re = right + (((0u << 6) | cr) << 9), *ro = right + (((o_d << 6) | cr) << 9);
        const uint16_t *de = dense + ((size_t)c << WB), *dd = dense + ((size_t)((CM + 1) * (HGR ? 1 : 2) + c) << WB);
        uint32_t a0 = le[row.x & LM], b0 = re[(row.x >> 9) & 0x1ffu];
        uint32_t a1 = lo[row.y & LM], b1 = ro[(row.y >> 9) & 0x1ffu];
        uint32_t a2 = le[row.z & LM], b2 = re[(row.z >> 9) & 0x1ffu];
        uint32_t a3 = lo[row.w & LM], b3 = ro[(row.w >> 9) & 0x1ffu];
        if (((row.x >> 20) & 63u) == 0) a0 = de[row.x & ((1u << WB) - 1)];
        if (((row.y >> 20) & 63u) == 0) a1 = dd[row.y & ((1u << WB) - 1)];
        if (((row.z >> 20) & 63u) == 0) a2 = de[row.z & ((1u << WB) - 1)];
        if (((row.w >> 20) & 63u) == 0) a3 = dd[row.w & ((1u << WB) - 1)];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    sink[blockIdx.x * 64 + lane] = acc;
}


// Variant E: a workgroup of W one-wave streams shares the bank's whole L1 half (DHGR: 2 offsets x 32 x 256
// u16 = 32 KiB), copied into LDS once per launch: four of the eight gathers become ds_read_u16, the other
// four (R1) and the 1-in-64 dense load stay with the L1/TA.  LDS per workgroup = 32 KiB + W x per-stream bytes.
template <int W>
__global__ __launch_bounds__(64 * W) void pattern_shared_l1_kernel(const uint16_t *__restrict__ left, const uint16_t *__restrict__ right,
                                                                   const uint16_t *__restrict__ dense, const uint4 *__restrict__ rows,
                                                                   int n_ops, int n_streams, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds[];   // [0, 8192): L1 of offsets 0 and 2; the rest: per-stream padding
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint4 *src0 = reinterpret_cast<const uint4 *>(left), *src2 = reinterpret_cast<const uint4 *>(left + 2 * 32 * 256);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < 1024; i += 64 * W) {
            dst[i] = src0[i];
            dst[1024 + i] = src2[i];
        }
    }
    __syncthreads();
    const int stream = blockIdx.x * W + wave;
    if (stream >= n_streams) return;
    const uint16_t *l16 = reinterpret_cast<const uint16_t *>(lds);
    const uint4 *my = rows + (size_t)stream * n_ops * 64 + lane;
    uint32_t acc = 0, h = stream * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;
        const uint16_t *le = l16 + ((c & 31u) << 8), *lo = l16 + 8192 + ((c & 31u) << 8);
        const uint16_t *re = right + (((0u << 6) | ((c >> 1) & 63u)) << 9), *ro = right + (((2u << 6) | ((c >> 1) & 63u)) << 9);
        const uint16_t *de = dense + ((size_t)c << 13), *dd = dense + ((size_t)(256 + c) << 13);
        uint32_t b0 = re[(row.x >> 9) & 0x1ffu], b1 = ro[(row.y >> 9) & 0x1ffu], b2 = re[(row.z >> 9) & 0x1ffu], b3 = ro[(row.w >> 9) & 0x1ffu];
        uint32_t a0 = le[row.x & 255u], a1 = lo[row.y & 255u], a2 = le[row.z & 255u], a3 = lo[row.w & 255u];
        if (((row.x >> 20) & 63u) == 0) a0 = de[row.x & 8191u];
        if (((row.y >> 20) & 63u) == 0) a1 = dd[row.y & 8191u];
        if (((row.z >> 20) & 63u) == 0) a2 = de[row.z & 8191u];
        if (((row.w >> 20) & 63u) == 0) a3 = dd[row.w & 8191u];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
    }
    sink[stream * 64 + lane] = acc;
}


// Variant E for HGR: the L1 table of ONE of the bank's two byte offsets (64 content parts x 512 rows x 2 B = 64 KiB) shared in
// LDS by a workgroup of W one-wave streams (one workgroup per CU): two of the eight gathers become ds_read_u16.
template <int W>
__global__ __launch_bounds__(64 * W) void pattern_shared_l1_hgr_kernel(const uint16_t *__restrict__ left, const uint16_t *__restrict__ right,
                                                                       const uint16_t *__restrict__ dense, const uint4 *__restrict__ rows,
                                                                       int n_ops, int n_streams, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds[];   // [0, 16384): L1 of offset 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint4 *src0 = reinterpret_cast<const uint4 *>(left);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < 4096; i += 64 * W) dst[i] = src0[i];
    }
    __syncthreads();
    const int stream = blockIdx.x * W + wave;
    if (stream >= n_streams) return;
    const uint16_t *l16 = reinterpret_cast<const uint16_t *>(lds);
    const uint4 *my = rows + (size_t)stream * n_ops * 64 + lane;
    uint32_t acc = 0, h = stream * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 255u;
        const uint32_t cl = c & 63u, cr = c >> 2;
        const uint16_t *le = l16 + (cl << 9), *lo = left + (((1u << 6) | cl) << 9);
        const uint16_t *re = right + (cr << 9), *ro = right + (((1u << 6) | cr) << 9);
        const uint16_t *de = dense + ((size_t)c << 14), *dd = dense + ((size_t)(256 + c) << 14);
        uint32_t b0 = re[(row.x >> 9) & 0x1ffu], b1 = ro[(row.y >> 9) & 0x1ffu], b2 = re[(row.z >> 9) & 0x1ffu], b3 = ro[(row.w >> 9) & 0x1ffu];
        uint32_t a1 = lo[row.y & 511u], a3 = lo[row.w & 511u];
        uint32_t a0 = le[row.x & 511u], a2 = le[row.z & 511u];
        if (((row.x >> 20) & 63u) == 0) a0 = de[row.x & 16383u];
        if (((row.y >> 20) & 63u) == 0) a1 = dd[row.y & 16383u];
        if (((row.z >> 20) & 63u) == 0) a2 = de[row.z & 16383u];
        if (((row.w >> 20) & 63u) == 0) a3 = dd[row.w & 16383u];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
    }
    sink[stream * 64 + lane] = acc;
}

// Variant E2 for HGR (round 6): the L1 tables of BOTH byte offsets (128 KiB) shared in LDS by sixteen one-wave streams (one
// workgroup per CU, 2 KiB of LDS left per stream: the two bitmaps, MT19937 in registers): four of the eight gathers are ds_read_u16.
template <int W>
__global__ __launch_bounds__(64 * W) void pattern_shared_l1x2_hgr_kernel(const uint16_t *__restrict__ left, const uint16_t *__restrict__ right,
                                                                         const uint16_t *__restrict__ dense, const uint4 *__restrict__ rows,
                                                                         int n_ops, int n_streams, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds[];   // [0, 32768): L1 of offsets 0 and 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint4 *src0 = reinterpret_cast<const uint4 *>(left);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < 8192; i += 64 * W) dst[i] = src0[i];
    }
    __syncthreads();
    const int stream = blockIdx.x * W + wave;
    if (stream >= n_streams) return;
    const uint16_t *l16 = reinterpret_cast<const uint16_t *>(lds);
    const uint4 *my = rows + (size_t)stream * n_ops * 64 + lane;
    uint32_t acc = 0, h = stream * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 255u;
        const uint32_t cl = c & 63u, cr = c >> 2;
        const uint16_t *le = l16 + (cl << 9), *lo = l16 + (((1u << 6) | cl) << 9);
        const uint16_t *re = right + (cr << 9), *ro = right + (((1u << 6) | cr) << 9);
        uint32_t b0 = re[(row.x >> 9) & 0x1ffu], b1 = ro[(row.y >> 9) & 0x1ffu], b2 = re[(row.z >> 9) & 0x1ffu], b3 = ro[(row.w >> 9) & 0x1ffu];
        uint32_t a1 = lo[row.y & 511u], a3 = lo[row.w & 511u];
        uint32_t a0 = le[row.x & 511u], a2 = le[row.z & 511u];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
    }
    sink[stream * 64 + lane] = acc;
}

// Variant G: the store value as a sum of FOUR pixel-group terms (DESIGN 3.10 applied to the store table: groups of 2 / 3 / 3 / 2
// pixels depend on 5 / 7 / 7 / 6 target dots and 2 / 5 / 6 / 3 content bits -- 128 + 4096 + 8192 + 512 entries per offset
// class, 25.3 KiB, both classes of a bank 50.5 KiB), all of it in LDS, shared by the W one-wave streams of one workgroup
// per CU: sixteen ds_read_u16 per lane and opcode, no divergent global load at all.  Layout [content part][target field], so
// that a step's lanes (one content, 64 fields) spread over the banks.
template <int W>
__global__ __launch_bounds__(64 * W) void pattern_groups_lds_kernel(const uint16_t *__restrict__ tab, const uint4 *__restrict__ rows,
                                                                    int n_ops, int n_streams, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds[];   // [0, 12928 u32): the two classes' tables; the rest: per-stream padding
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kClass = 128 + 4096 + 8192 + 512;   // u16 entries per class
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(tab);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        for (int i = threadIdx.x; i < 2 * kClass * 2 / 16; i += 64 * W) dst[i] = src[i];
    }
    __syncthreads();
    const int stream = blockIdx.x * W + wave;
    if (stream >= n_streams) return;
    const unsigned char *t8 = reinterpret_cast<const unsigned char *>(lds);
    const uint4 *my = rows + (size_t)stream * n_ops * 64 + lane;
    uint32_t acc = 0, h = stream * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;
        // scalar byte offsets of this content's slices: class base + group base + content part << field bits, times 2
        uint32_t sa[2], sb[2], sc[2], sd[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t cls = (uint32_t)k * kClass;
            sa[k] = __builtin_amdgcn_readfirstlane(2u * (cls + ((c & 3u) << 5)));
            sb[k] = __builtin_amdgcn_readfirstlane(2u * (cls + 128u + ((c & 31u) << 7)));
            sc[k] = __builtin_amdgcn_readfirstlane(2u * (cls + 128u + 4096u + (((c >> 1) & 63u) << 7)));
            sd[k] = __builtin_amdgcn_readfirstlane(2u * (cls + 128u + 4096u + 8192u + ((c >> 4) << 6)));
        }
        const uint32_t w[4] = {row.x, row.y, row.z, row.w};
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = r & 1;
            const uint32_t fa = __builtin_amdgcn_ubfe(w[r], 0, 5), fb = __builtin_amdgcn_ubfe(w[r], 1, 7), fc = __builtin_amdgcn_ubfe(w[r], 4, 7),
                           fd = __builtin_amdgcn_ubfe(w[r], 7, 6);
            v[r] = *reinterpret_cast<const uint16_t *>(t8 + (fa * 2 + sa[k])) + *reinterpret_cast<const uint16_t *>(t8 + (fb * 2 + sb[k])) +
                   *reinterpret_cast<const uint16_t *>(t8 + (fc * 2 + sc[k])) + *reinterpret_cast<const uint16_t *>(t8 + (fd * 2 + sd[k]));
        }
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    sink[stream * 64 + lane] = acc;
}

// Variant B: four 8-byte gathers (as if both halves of a byte's value sat side by side in one
// 6 KiB slice per opcode): what a layout that serves a byte with ONE load would buy.
__global__ __launch_bounds__(64) void pattern_x2_kernel(const uint2 *__restrict__ both, const uint4 *__restrict__ rows,
                                                        int n_ops, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x;
    const uint4 *my = rows + (size_t)blockIdx.x * n_ops * 64 + lane;
    uint32_t acc = 0, h = blockIdx.x * 2654435761u + 977u;
    uint4 next = my[0];
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;
        const uint2 *e = both + ((0u << 7 | c) * 384u), *o = both + ((1u << 7 | c) * 384u);   // 384 rows x 8 B = 3 KiB per parity
        const uint2 v0 = e[(row.x & 0x1ffu) % 384u], v1 = o[(row.y & 0x1ffu) % 384u], v2 = e[(row.z & 0x1ffu) % 384u],
                    v3 = o[(row.w & 0x1ffu) % 384u];
        acc += (v0.x + v0.y) ^ (v1.x + v1.y) ^ (v2.x + v2.y) ^ (v3.x + v3.y);
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    sink[blockIdx.x * 64 + lane] = acc;
}

// Variant C: the opcode's four slices (6 KiB) copied into LDS with coalesced 16-byte loads, the
// eight gathers served by LDS; LDS per wave 5.7 + 6 KiB, i.e. 13 waves per CU.
__global__ __launch_bounds__(64) void pattern_lds_kernel(const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                         const uint4 *__restrict__ rows, int n_ops, uint32_t *__restrict__ sink)
{
    extern __shared__ uint32_t lds[];   // [0, 1536): the slices; the rest: padding
    const int lane = threadIdx.x;
    const uint4 *my = rows + (size_t)blockIdx.x * n_ops * 64 + lane;
    uint32_t acc = 0, h = blockIdx.x * 2654435761u + 977u;
    uint4 next = my[0];
    uint4 *l4 = reinterpret_cast<uint4 *>(lds);
    for (int op = 0; op < n_ops; op++) {
        const uint4 row = next;
        if (op + 1 < n_ops) next = my[(size_t)(op + 1) * 64];
        h = h * 1664525u + 1013904223u;
        const uint32_t c = (h >> 16) & 127u;
        const uint4 *le = reinterpret_cast<const uint4 *>(left + (((0u << 5) | (c & 31u)) << 8));
        const uint4 *lo = reinterpret_cast<const uint4 *>(left + (((2u << 5) | (c & 31u)) << 8));
        const uint4 *re = reinterpret_cast<const uint4 *>(right + (((0u << 6) | ((c >> 1) & 63u)) << 9));
        const uint4 *ro = reinterpret_cast<const uint4 *>(right + (((2u << 6) | ((c >> 1) & 63u)) << 9));
        const uint4 s0 = le[lane], s1 = lo[lane], s2 = re[lane], s3 = re[64 + lane], s4 = ro[lane], s5 = ro[64 + lane];
        l4[lane] = s0; l4[64 + lane] = s1; l4[128 + lane] = s2; l4[192 + lane] = s3; l4[256 + lane] = s4; l4[320 + lane] = s5;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t a0 = lds[row.x & 0xffu], b0 = lds[512 + ((row.x >> 8) & 0x1ffu)];
        const uint32_t a1 = lds[256 + (row.y & 0xffu)], b1 = lds[1024 + ((row.y >> 8) & 0x1ffu)];
        const uint32_t a2 = lds[row.z & 0xffu], b2 = lds[512 + ((row.z >> 8) & 0x1ffu)];
        const uint32_t a3 = lds[256 + (row.w & 0xffu)], b3 = lds[1024 + ((row.w >> 8) & 0x1ffu)];
        acc += (a0 + b0) ^ (a1 + b1) ^ (a2 + b2) ^ (a3 + b3);
        __builtin_amdgcn_wave_barrier();
    }
    sink[blockIdx.x * 64 + lane] = acc;
}

// `gather_ceiling D <waves> [HGR]`: only variant D (the kernels' present access pattern) with that many
// waves, one line "D <waves> <ms per launch> <G table loads per s>" -- what bench.py runs.
static int run_d_only(int waves, bool hgr)
{
    const int n_ops = hgr ? 490 : 183;   // opcodes per generator under Movie pacing
    const size_t nl = hgr ? 2 * 64 * 512 : 4 * 32 * 256, nr = hgr ? 2 * 64 * 512 : 4 * 64 * 512, n_rows = (size_t)waves * n_ops * 64;
    const size_t dense_bytes = hgr ? (size_t)16 << 20 : (size_t)8 << 20;
    uint32_t *left, *right, *sink;
    uint16_t *dense;
    uint4 *rows;
    if (hipMalloc(&left, nl * 4) != hipSuccess || hipMalloc(&right, nr * 4) != hipSuccess ||
        hipMalloc(&rows, n_rows * sizeof(uint4)) != hipSuccess || hipMalloc(&sink, (size_t)waves * 64 * 4) != hipSuccess ||
        hipMalloc(&dense, dense_bytes) != hipSuccess)
        return 1;
    (void)hipMemset(left, 1, nl * 4);
    (void)hipMemset(right, 1, nr * 4);
    (void)hipMemset(dense, 1, dense_bytes);
    {
        const size_t chunk = 1 << 24;
        uint32_t *h = (uint32_t *)malloc(chunk * 4);
        uint32_t s = 12345;
        for (size_t o = 0; o < n_rows * 4; o += chunk) {
            const size_t n = n_rows * 4 - o < chunk ? n_rows * 4 - o : chunk;
            for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 7; }
            (void)hipMemcpy((uint32_t *)rows + o, h, n * 4, hipMemcpyHostToDevice);
        }
        free(h);
    }
    auto launch = [&] {
        if (hgr)
            hipLaunchKernelGGL(pattern_u16_rare_kernel<true>, dim3(waves), dim3(64), 5824, 0, (const uint16_t *)left,
                               (const uint16_t *)right, dense, rows, n_ops, sink);
        else
            hipLaunchKernelGGL(pattern_u16_rare_kernel<false>, dim3(waves), dim3(64), 5824, 0, (const uint16_t *)left,
                               (const uint16_t *)right, dense, rows, n_ops, sink);
    };
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    const int reps = hgr ? 4 : 10;
    for (int r = 0; r < reps; r++) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    if (hipGetLastError() != hipSuccess) return 1;
    if (hgr && getenv("IIV_GATHER_E")) {   // variant E for HGR beside D (diagnostic)
        (void)hipFuncSetAttribute((const void *)pattern_shared_l1_hgr_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)pattern_shared_l1_hgr_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int w : {16, 12}) {
            auto le = [&] {
                if (w == 16)
                    hipLaunchKernelGGL(pattern_shared_l1_hgr_kernel<16>, dim3((waves + 15) / 16), dim3(1024), 65536 + 16 * 5632, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense, rows, n_ops, waves, sink);
                else
                    hipLaunchKernelGGL(pattern_shared_l1_hgr_kernel<12>, dim3((waves + 11) / 12), dim3(768), 65536 + 12 * 5632, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense, rows, n_ops, waves, sink);
            };
            le();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            for (int r = 0; r < reps; r++) le();
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms2;
            (void)hipEventElapsedTime(&ms2, a, b);
            printf("# E-HGR W=%d (one offset's L1 in LDS, 1 workgroup per CU): %.4f ms per launch  err=%d\n", w, ms2 / reps, (int)hipGetLastError());
        }
        {
            (void)hipFuncSetAttribute((const void *)pattern_shared_l1x2_hgr_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            auto le2 = [&] {
                hipLaunchKernelGGL(pattern_shared_l1x2_hgr_kernel<16>, dim3((waves + 15) / 16), dim3(1024), 131072 + 16 * 2048, 0, (const uint16_t *)left,
                                   (const uint16_t *)right, dense, rows, n_ops, waves, sink);
            };
            le2();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            for (int r = 0; r < reps; r++) le2();
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms3;
            (void)hipEventElapsedTime(&ms3, a, b);
            printf("# E2-HGR W=16 (both offsets' L1 in LDS, 160 KiB, 1 workgroup per CU): %.4f ms per launch  err=%d\n", ms3 / reps, (int)hipGetLastError());
        }
        printf("D %d %.4f %.1f\n", waves, ms, (double)waves * n_ops * 512 / ms * 1e-6);
        return 0;
    }
    printf("D %d %.4f %.1f\n", waves, ms, (double)waves * n_ops * 512 / ms * 1e-6);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && argv[1][0] == 'D') return run_d_only(atoi(argv[2]), argc >= 4 && argv[3][0] == 'H');
    const int n_ops = 183;
    const size_t nl = 4 * 32 * 256, nr = 4 * 64 * 512;
    uint32_t *left, *right, *sink;
    uint4 *rows;
    for (int waves : {7168, 14336}) {
        const size_t n_rows = (size_t)waves * n_ops * 64;
        (void)hipMalloc(&left, nl * 4);
        (void)hipMalloc(&right, nr * 4);
        (void)hipMalloc(&rows, n_rows * sizeof(uint4));
        (void)hipMalloc(&sink, (size_t)waves * 64 * 4);
        (void)hipMemset(left, 1, nl * 4);
        (void)hipMemset(right, 1, nr * 4);
        {   // random rows (the gathers' addresses)
            const size_t chunk = 1 << 24;
            uint32_t *h = (uint32_t *)malloc(chunk * 4);
            uint32_t s = 12345;
            for (size_t o = 0; o < n_rows * 4; o += chunk) {
                const size_t n = n_rows * 4 - o < chunk ? n_rows * 4 - o : chunk;
                for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 7; }
                (void)hipMemcpy((uint32_t *)rows + o, h, n * 4, hipMemcpyHostToDevice);
            }
            free(h);
        }
        for (int lds : {5824, 5120}) {
            for (int gathers : {1, 0}) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                hipLaunchKernelGGL(pattern_kernel, dim3(waves), dim3(64), lds, 0, left, right, rows, n_ops, gathers, sink);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a);
                const int reps = 5;
                for (int r = 0; r < reps; r++)
                    hipLaunchKernelGGL(pattern_kernel, dim3(waves), dim3(64), lds, 0, left, right, rows, n_ops, gathers, sink);
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                ms /= reps;
                const double lookups = (double)waves * n_ops * 512;
                printf("waves %5d  LDS/wave %4d B  %-28s %7.3f ms per launch", waves, lds,
                       gathers ? "row + 8 gathers per opcode:" : "row only (HBM stream):", ms);
                if (gathers)
                    printf("  %7.1f G lookups/s  (%.2f per CU-cycle @2.4 GHz)\n", lookups / ms * 1e-6, lookups / (ms * 1e-3) / 256 / 2.4e9);
                else
                    printf("  %7.1f GB/s\n", (double)waves * n_ops * 1024 / ms * 1e-6);
            }
        }
        {
            auto time = [&](const char *name, auto launch) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a);
                (void)hipEventCreate(&b);
                launch();
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(a);
                for (int r = 0; r < 5; r++) launch();
                (void)hipEventRecord(b);
                (void)hipEventSynchronize(b);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                ms /= 5;
                const double lookups = (double)waves * n_ops * 512;
                printf("waves %5d  %-58s %7.3f ms per launch  %7.1f G lookups/s  err=%d\n", waves, name, ms, lookups / ms * 1e-6,
                       (int)hipGetLastError());
            };
            uint2 *both;
            (void)hipMalloc(&both, (size_t)2 * 128 * 384 * 8);
            (void)hipMemset(both, 1, (size_t)2 * 128 * 384 * 8);
            time("A: eight 2-byte gathers (half-size slices)", [&] {
                hipLaunchKernelGGL(pattern_u16_kernel, dim3(waves), dim3(64), 5824, 0, (const uint16_t *)left, (const uint16_t *)right,
                                   rows, n_ops, sink);
            });
            uint16_t *dense;
            (void)hipMalloc(&dense, (size_t)8 << 20);
            (void)hipMemset(dense, 1, (size_t)8 << 20);
            time("D: A + a dense-table gather for 1 byte in 64", [&] {
                hipLaunchKernelGGL(pattern_u16_rare_kernel<false>, dim3(waves), dim3(64), 5824, 0, (const uint16_t *)left,
                                   (const uint16_t *)right, dense, rows, n_ops, sink);
            });
            (void)hipFree(dense);
            time("B: four 8-byte gathers (one load per byte, 6 KiB slices)", [&] {
                hipLaunchKernelGGL(pattern_x2_kernel, dim3(waves), dim3(64), 5824, 0, both, rows, n_ops, sink);
            });
            time("C: slices copied to LDS, gathers from LDS (13 waves per CU)", [&] {
                hipLaunchKernelGGL(pattern_lds_kernel, dim3(waves), dim3(64), 5824 + 6144, 0, left, right, rows, n_ops, sink);
            });
            {
                uint16_t *dense2;
                (void)hipMalloc(&dense2, (size_t)8 << 20);
                (void)hipMemset(dense2, 1, (size_t)8 << 20);
                // per-stream LDS: 5.8 KiB (today's), 3.8 KiB (bitmaps in registers)
                (void)hipFuncSetAttribute((const void *)pattern_shared_l1_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pattern_shared_l1_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pattern_shared_l1_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)pattern_shared_l1_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                time("E: L1 half shared in LDS, W=8,  2 wg/CU (16 waves/CU, 5.8 KiB/stream)", [&] {
                    hipLaunchKernelGGL(pattern_shared_l1_kernel<8>, dim3((waves + 7) / 8), dim3(512), 32768 + 8 * 5824, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense2, rows, n_ops, waves, sink);
                });
                time("E: L1 half shared in LDS, W=16, 1 wg/CU (16 waves/CU, 5.8 KiB/stream)", [&] {
                    hipLaunchKernelGGL(pattern_shared_l1_kernel<16>, dim3((waves + 15) / 16), dim3(1024), 32768 + 16 * 5824, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense2, rows, n_ops, waves, sink);
                });
                time("E: L1 half shared in LDS, W=12, 2 wg/CU (24 waves/CU, 3.8 KiB/stream)", [&] {
                    hipLaunchKernelGGL(pattern_shared_l1_kernel<12>, dim3((waves + 11) / 12), dim3(768), 32768 + 12 * 3840, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense2, rows, n_ops, waves, sink);
                });
                time("E: L1 half shared in LDS, W=10, 2 wg/CU (20 waves/CU, 4.7 KiB/stream)", [&] {
                    hipLaunchKernelGGL(pattern_shared_l1_kernel<10>, dim3((waves + 9) / 10), dim3(640), 32768 + 10 * 4800, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense2, rows, n_ops, waves, sink);
                });
                time("E: L1 half shared in LDS, W=16, 1 wg/CU, then D with 16 waves/CU for comparison", [&] {
                    hipLaunchKernelGGL(pattern_u16_rare_kernel<false>, dim3(waves), dim3(64), 10240, 0, (const uint16_t *)left,
                                       (const uint16_t *)right, dense2, rows, n_ops, sink);
                });
                {
                    uint16_t *gt;
                    const size_t gbytes = (size_t)2 * (128 + 4096 + 8192 + 512) * 2;
                    (void)hipMalloc(&gt, gbytes);
                    (void)hipMemset(gt, 1, gbytes);
                    (void)hipFuncSetAttribute((const void *)pattern_groups_lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute((const void *)pattern_groups_lds_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    time("G: four pixel-group tables in LDS (50.5 KiB), W=16, 1 wg/CU, no gathers", [&] {
                        hipLaunchKernelGGL(pattern_groups_lds_kernel<16>, dim3((waves + 15) / 16), dim3(1024), (int)gbytes + 16 * 5824, 0, gt, rows, n_ops,
                                           waves, sink);
                    });
                    time("G: the same, W=12 (12 waves/CU)", [&] {
                        hipLaunchKernelGGL(pattern_groups_lds_kernel<12>, dim3((waves + 11) / 12), dim3(768), (int)gbytes + 12 * 5824, 0, gt, rows, n_ops,
                                           waves, sink);
                    });
                    (void)hipFree(gt);
                }
                (void)hipFree(dense2);
            }
            (void)hipFree(both);
        }
        (void)hipFree(left); (void)hipFree(right); (void)hipFree(rows); (void)hipFree(sink);
    }
    return 0;
}
