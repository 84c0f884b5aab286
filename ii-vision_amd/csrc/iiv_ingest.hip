// iiv_ingest.hip -- RGB frames -> HGR / DHGR memory maps on gfx950.
//
// The reference shells out to an external C tool, /usr/local/bin/bmp2dhr, for this step
// (transcoder/frame_grabber.py:68-115; README.md:217-221 wishes for a "direct image
// encoding").  That tool is not part of the reference's source and is absent here, so this
// conversion has no reference output to match: it is specified, in integer arithmetic, in
// include/iivision.h (iiv_frames_to_memory_maps), and the tests hold this kernel to a CPU
// restatement of that specification bit for bit.
//
// One thread per screen byte (HGR 40 x 192, DHGR 80 x 192 per frame): it resolves the colour
// pixels its seven dots belong to (mean of two source pixels + 4x4 ordered dither -> nearest
// palette colour in weighted integer RGB) and writes the byte at its memory-map position
// (screen.py:16-69: y_to_base_addr).  Frames are independent; the work is a coalesced read of
// 161 KB per frame and 8 / 16 KiB of stores: HBM-bound, nothing to tile.
#include "iiv_host.h"

namespace iiv {

__device__ static inline int y_to_offset(int y)  // y_to_base_addr(y, 0) - 0x2000 (screen.py:16-22)
{
    return 1024 * (y % 8) + 128 * ((y % 64) / 8) + 40 * (y / 64);
}

struct IngestPixel {
    int r, g, b;
};

__device__ static inline IngestPixel ingest_pixel(const uint8_t *__restrict__ rgb, int y, int k, int dither)
{
    constexpr int bayer[16] = {0, 8, 2, 10, 12, 4, 14, 6, 3, 11, 1, 9, 15, 7, 13, 5};
    const uint8_t *p = rgb + ((size_t)y * 280 + 2 * k) * 3;
    const int d = ((2 * bayer[(y & 3) * 4 + (k & 3)] - 15) * dither + 16 * 256) / 16 - 256;
    IngestPixel o;
    int v = ((int)p[0] + (int)p[3] + 1) / 2 + d;
    o.r = v < 0 ? 0 : v > 255 ? 255 : v;
    v = ((int)p[1] + (int)p[4] + 1) / 2 + d;
    o.g = v < 0 ? 0 : v > 255 ? 255 : v;
    v = ((int)p[2] + (int)p[5] + 1) / 2 + d;
    o.b = v < 0 ? 0 : v > 255 ? 255 : v;
    return o;
}

__device__ static inline int ingest_err(const uint8_t *pal, int c, const IngestPixel &px)
{
    const int dr = px.r - pal[3 * c], dg = px.g - pal[3 * c + 1], db = px.b - pal[3 * c + 2];
    return 2 * dr * dr + 4 * dg * dg + 3 * db * db;
}

template <int MODE>
__global__ __launch_bounds__(256) void ingest_kernel(int n, const uint8_t *__restrict__ rgb_frames, int dither,
                                                     const uint8_t *__restrict__ palette, uint8_t *__restrict__ main_mem,
                                                     uint8_t *__restrict__ aux_mem)
{
    constexpr int BPR = MODE == kDHGR ? 80 : 40;  // screen bytes per row, in dot order
    __shared__ uint8_t pal[48];
    if (threadIdx.x < 48) pal[threadIdx.x] = palette[threadIdx.x];
    __syncthreads();
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * 192 * BPR) return;
    const int j = (int)(idx % BPR), y = (int)((idx / BPR) % 192);
    const size_t f = idx / ((size_t)BPR * 192);
    const uint8_t *rgb = rgb_frames + f * (size_t)(192 * 280 * 3);
    const int base = y_to_offset(y);
    if (MODE == kDHGR) {
        // dots 7j .. 7j+6 lie in the quads (7j) >> 2 .. (7j + 6) >> 2 (two or three of them)
        const int q0 = (7 * j) >> 2, q1 = (7 * j + 6) >> 2;
        int quad[3] = {0, 0, 0};
        for (int q = q0; q <= q1; q++) {
            const IngestPixel px = ingest_pixel(rgb, y, q, dither);
            int best = 0, be = 0x7fffffff;
            for (int c = 0; c < 16; c++) {
                const int e = ingest_err(pal, c, px);
                if (e < be) {
                    be = e;
                    best = c;
                }
            }
            quad[q - q0] = best;
        }
        int v = 0;
        for (int i = 0; i < 7; i++) {
            const int X = 7 * j + i;
            v |= ((quad[(X >> 2) - q0] >> (X & 3)) & 1) << i;
        }
        // bytes alternate aux, main in dot order (screen.py:822-826)
        ((j & 1) ? main_mem : aux_mem)[f * 8192 + base + (j >> 1)] = (uint8_t)v;
    } else {
        // the colour values a pixel can take under either palette bit, pattern bit 0 = even dot
        // column (colours.py:18-44): black, violet | blue, green | orange, white
        constexpr int colour[2][4] = {{0, 3, 12, 15}, {0, 6, 9, 15}};
        const int k0 = (7 * j) >> 1, k1 = (7 * j + 6) >> 1;  // four pixels touch the byte
        int pat[2][4], err[2][4];
        for (int k = k0; k <= k1; k++) {
            const IngestPixel px = ingest_pixel(rgb, y, k, dither);
            for (int pb = 0; pb < 2; pb++) {
                int best = 0, be = 0x7fffffff;
                for (int q = 0; q < 4; q++) {
                    const int e = ingest_err(pal, colour[pb][q], px);
                    if (e < be) {
                        be = e;
                        best = q;
                    }
                }
                pat[pb][k - k0] = best;
                err[pb][k - k0] = be;
            }
        }
        long e0 = 0, e1 = 0;
        for (int i = 0; i < 7; i++) {
            e0 += err[0][((7 * j + i) >> 1) - k0];
            e1 += err[1][((7 * j + i) >> 1) - k0];
        }
        const int pb = e1 < e0 ? 1 : 0;
        int v = pb << 7;
        for (int i = 0; i < 7; i++) {
            const int X = 7 * j + i;
            v |= ((pat[pb][(X >> 1) - k0] >> (X & 1)) & 1) << i;
        }
        main_mem[f * 8192 + base + j] = (uint8_t)v;
    }
}

int frames_to_memory_maps(int mode, const uint8_t palette_rgb[48], int n, const uint8_t *d_rgb, int dither, uint8_t *d_main,
                          uint8_t *d_aux, hipStream_t st)
{
    uint8_t *d_pal = nullptr;
    IIV_HIP(hipMalloc(&d_pal, 48));
    int rc = hip_check(hipMemcpyAsync(d_pal, palette_rgb, 48, hipMemcpyHostToDevice, st), "copy palette");
    // screen holes (and everything else) start as zero, as bmp2dhr's files hold them
    if (!rc) rc = hip_check(hipMemsetAsync(d_main, 0, (size_t)n * 8192, st), "clear main");
    if (!rc && mode == kDHGR) rc = hip_check(hipMemsetAsync(d_aux, 0, (size_t)n * 8192, st), "clear aux");
    if (!rc) {
        const size_t total = (size_t)n * 192 * (mode == kDHGR ? 80 : 40);
        dim3 grid((unsigned)((total + 255) / 256));
        if (mode == kDHGR)
            hipLaunchKernelGGL(ingest_kernel<kDHGR>, grid, dim3(256), 0, st, n, d_rgb, dither, d_pal, d_main, d_aux);
        else
            hipLaunchKernelGGL(ingest_kernel<kHGR>, grid, dim3(256), 0, st, n, d_rgb, dither, d_pal, d_main, d_aux);
        rc = hip_check(hipGetLastError(), "ingest_kernel launch");
    }
    if (!rc) rc = hip_check(hipStreamSynchronize(st), "ingest sync");  // palette_rgb is caller memory; d_pal freed below
    (void)hipFree(d_pal);
    return rc;
}

}  // namespace iiv

extern "C" int iiv_frames_to_memory_maps(int mode, const uint8_t palette_rgb[48], int n_frames, const uint8_t *d_rgb,
                                         int dither, uint8_t *d_main, uint8_t *d_aux, void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || !palette_rgb || n_frames < 0 || !d_rgb || !d_main ||
        (mode == IIV_DHGR && !d_aux) || dither < 0 || dither > 255)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_frames_to_memory_maps: bad argument");
    if (n_frames == 0) return IIV_OK;
    return iiv::frames_to_memory_maps(mode, palette_rgb, n_frames, d_rgb, dither, d_main, d_aux, (hipStream_t)stream);
}
