// iiv_bitmap.hip -- screen.Bitmap operations on gfx950, batched over independent
// screens (reference: transcoder/screen.py).
//
// K4 pack_kernel          Bitmap._pack                     screen.py:207-226
// K5 diff_weights_kernel  Bitmap._diff_weights             screen.py:409-449
// K7 delta_pages_kernel   Bitmap.compute_delta_page        screen.py:525-547
//
// These are the stand-alone forms behind screen.HGRBitmap / DHGRBitmap; the
// encoder kernels (iiv_prologue.hip, iiv_greedy.hip) fuse the same arithmetic into their prologue
// and greedy loop and never materialise the packed array.
#include "iiv_host.h"

namespace iiv {

template <int MODE> __device__ static inline uint64_t body_of(const uint8_t *m, const uint8_t *a, int col)
{
    if (MODE == kDHGR) {
        // screen.py:939-947
        uint64_t a0 = a[2 * col] & 0x7f, m0 = m[2 * col] & 0x7f;
        uint64_t a1 = a[2 * col + 1] & 0x7f, m1 = m[2 * col + 1] & 0x7f;
        return (a0 << 3) + (m0 << 10) + (a1 << 17) + (m1 << 24);
    }
    // screen.py:672-677
    uint64_t even = m[2 * col], odd = m[2 * col + 1];
    return (even << 3) + ((odd & 0x7f) << 12) + ((odd & 0x80) << 4);
}

template <int MODE> __device__ static inline uint64_t make_header(uint64_t col)
{
    if (MODE == kDHGR) return (col >> 28) & 7;                    // screen.py:924
    return (((col >> 11) & 1) << 2) ^ ((col >> 17) & 3);          // screen.py:658-661
}

template <int MODE> __device__ static inline uint64_t make_footer(uint64_t col)
{
    if (MODE == kDHGR) return (col & (7ull << 3)) << 28;          // screen.py:952
    return (((col >> 10) & 1) ^ (((col >> 3) & 3) << 1)) << 19;   // screen.py:687-690
}

// one thread per packed column; grid.x = n * 32 * 128 / 256
template <int MODE>
__global__ __launch_bounds__(256) void pack_kernel(int n, const uint8_t *__restrict__ main_mem,
                                                   const uint8_t *__restrict__ aux_mem, uint64_t *__restrict__ packed)
{
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * 4096) return;
    int col = idx & 127;
    size_t row = idx >> 7;  // screen * 32 + page
    const uint8_t *m = main_mem + row * 256;
    const uint8_t *a = MODE == kDHGR ? aux_mem + row * 256 : nullptr;
    uint64_t body = body_of<MODE>(m, a, col);
    // header from column c-1, footer from column c+1; only the page edges are
    // zeroed (screen.py:217,224)
    uint64_t header = col == 0 ? 0 : make_header<MODE>(body_of<MODE>(m, a, col - 1));
    uint64_t footer = col == 127 ? 0 : make_footer<MODE>(body_of<MODE>(m, a, col + 1));
    packed[idx] = header ^ body ^ footer;
}

template <int MODE> __device__ static inline uint32_t mask_and_shift(uint64_t packed, int o)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    int shift = MODE == kDHGR ? 7 * o : 8 * o;  // BYTE_SHIFTS screen.py:636,910
    return (uint32_t)(packed >> shift) & ((1u << BITS) - 1);
}

template <int MODE> __device__ static inline uint64_t masked_update(int o, uint64_t old, uint32_t v)
{
    if (MODE == kDHGR) {
        int sh = 7 * o + 3;  // screen.py:1001-1007
        return (old & ~(0x7full << sh)) ^ ((uint64_t)(v & 0x7f) << sh);
    }
    if (o == 0) return (old & ~(0xffull << 3)) ^ ((uint64_t)v << 3);  // screen.py:801-805
    uint64_t sv = ((v & 0x7f) << 1) ^ ((v & 0x80) >> 7);               // screen.py:807-816
    return (old & ~(0xffull << 11)) ^ (sv << 11);
}

// one thread per screen byte
template <int MODE>
__global__ __launch_bounds__(256) void diff_weights_kernel(const uint16_t *__restrict__ table, int n,
                                                           const uint64_t *__restrict__ src,
                                                           const uint64_t *__restrict__ tgt, int is_aux,
                                                           int32_t *__restrict__ out)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * 8192) return;
    int y = idx & 255;
    size_t col = idx >> 1;  // (screen*32+page)*128 + y/2
    int o = byte_offset<MODE>(y, is_aux);
    uint32_t sp = mask_and_shift<MODE>(src[col], o);
    uint32_t tp = mask_and_shift<MODE>(tgt[col], o);
    size_t pair = ((size_t)sp << BITS) + tp;  // screen.py:441
    out[idx] = table[((size_t)o << (2 * BITS)) + pair];
}

// one block per (page, content) query
template <int MODE>
__global__ __launch_bounds__(256) void delta_pages_kernel(const uint16_t *__restrict__ table,
                                                          const uint64_t *__restrict__ tgt,
                                                          const int32_t *__restrict__ pages,
                                                          const int32_t *__restrict__ contents,
                                                          const int32_t *__restrict__ dw_rows, int is_aux,
                                                          int32_t *__restrict__ out)
{
    constexpr int BITS = ModeTraits<MODE>::kBits;
    int q = blockIdx.x, y = threadIdx.x;
    int page = pages[q];
    uint32_t content = (uint32_t)contents[q] & 0xff;
    int o = byte_offset<MODE>(y, is_aux);
    uint64_t t = tgt[page * 128 + (y >> 1)];
    // _diff_weights_page(packed_page, packed_page, is_aux, content): source =
    // target with `content` stored at this byte (screen.py:476-486).  The
    // _fix_array_neighbours call there only rewrites header/footer bits that
    // lie outside byte o's mask, so it does not reach the lookup.
    uint32_t sp = mask_and_shift<MODE>(masked_update<MODE>(o, t, content), o);
    uint32_t tp = mask_and_shift<MODE>(t, o);
    size_t pair = ((size_t)sp << BITS) + tp;
    int32_t nd = table[((size_t)o << (2 * BITS)) + pair];
    out[q * 256 + y] = nd - dw_rows[q * 256 + y];  // screen.py:547
}

#define IIV_DISPATCH(mode, KERNEL, grid, block, st, ...)                                   \
    do {                                                                                   \
        if ((mode) == kDHGR)                                                               \
            hipLaunchKernelGGL(KERNEL<kDHGR>, grid, block, 0, st, __VA_ARGS__);            \
        else                                                                               \
            hipLaunchKernelGGL(KERNEL<kHGR>, grid, block, 0, st, __VA_ARGS__);             \
    } while (0)

int pack(int mode, int n, const uint8_t *d_main, const uint8_t *d_aux, uint64_t *d_packed, hipStream_t st)
{
    if (n <= 0) return IIV_OK;
    dim3 grid((unsigned)(((size_t)n * 4096 + 255) / 256));
    IIV_DISPATCH(mode, pack_kernel, grid, dim3(256), st, n, d_main, d_aux, d_packed);
    return hip_check(hipGetLastError(), "pack_kernel launch");
}

int diff_weights(int mode, const uint16_t *d_table, int n, const uint64_t *d_src, const uint64_t *d_tgt,
                 int is_aux, int32_t *d_out, hipStream_t st)
{
    if (n <= 0) return IIV_OK;
    dim3 grid((unsigned)(((size_t)n * 8192 + 255) / 256));
    IIV_DISPATCH(mode, diff_weights_kernel, grid, dim3(256), st, d_table, n, d_src, d_tgt, is_aux, d_out);
    return hip_check(hipGetLastError(), "diff_weights_kernel launch");
}

int compute_delta_pages(int mode, const uint16_t *d_table, int n, const uint64_t *d_tgt, const int32_t *d_pages,
                        const int32_t *d_contents, const int32_t *d_dw_rows, int is_aux, int32_t *d_out,
                        hipStream_t st)
{
    if (n <= 0) return IIV_OK;
    IIV_DISPATCH(mode, delta_pages_kernel, dim3(n), dim3(256), st, d_table, d_tgt, d_pages, d_contents, d_dw_rows,
                 is_aux, d_out);
    return hip_check(hipGetLastError(), "delta_pages_kernel launch");
}

}  // namespace iiv
