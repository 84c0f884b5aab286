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

// dither == IIV_DITHER_DIFFUSION: Floyd-Steinberg error diffusion (include/iivision.h).  A pixel needs the errors of
// its left neighbour and of three pixels of the row above, so the rows of a frame advance as a skewed wavefront: one
// thread per row, row y works on pixel t - 6 y at step t (six behind the row above: HGR fixes a screen byte's palette
// bit from the accumulated error of up to four pixels ahead, which must have received everything the row above
// sends them), one workgroup barrier per step, 140 + 6 * 191 steps per frame.  The accumulators of a row live in an
// eight-slot ring in LDS (pixel k in slot k & 7): the row above writes slots k + 5 .. k + 7 while the row itself
// reads k .. k + 3 and adds to k + 1.
template <int MODE>
__global__ __launch_bounds__(192) void ingest_diffusion_kernel(const uint8_t *__restrict__ rgb_frames, const uint8_t *__restrict__ palette,
                                                               uint8_t *__restrict__ main_mem, uint8_t *__restrict__ aux_mem)
{
    constexpr int colour4[2][4] = {{0, 3, 12, 15}, {0, 6, 9, 15}};
    __shared__ int ring[192][8][3];
    __shared__ uint8_t patt[192][140];   // DHGR: colour value = dot quad; HGR: 2-dot pattern | palette bit << 2
    __shared__ uint8_t pal[48];
    const int y = threadIdx.x;
    const size_t f = blockIdx.x;
    const uint8_t *rgb = rgb_frames + f * (size_t)(192 * 280 * 3) + (size_t)y * 280 * 3;
    if (y < 48) pal[y] = palette[y];
    for (int i = 0; i < 24; i++) (&ring[y][0][0])[i] = 0;
    __syncthreads();
    auto value = [&](int k, IngestPixel &o) {
        const uint8_t *p = rgb + 6 * k;
        const int *a = ring[y][k & 7];
        int v = ((int)p[0] + (int)p[3] + 1) / 2 + (a[0] >> 4);   // (>> 4 of a negative int: floor)
        o.r = v < 0 ? 0 : v > 255 ? 255 : v;
        v = ((int)p[1] + (int)p[4] + 1) / 2 + (a[1] >> 4);
        o.g = v < 0 ? 0 : v > 255 ? 255 : v;
        v = ((int)p[2] + (int)p[5] + 1) / 2 + (a[2] >> 4);
        o.b = v < 0 ? 0 : v > 255 ? 255 : v;
    };
    int pb = 0;
    for (int t = 0; t < 140 + 6 * 191; t++) {
        const int k = t - 6 * y;
        if (k >= 0 && k < 140) {
            IngestPixel px;
            int colour, pattern;
            if (MODE == kDHGR) {
                value(k, px);
                int best = 0, be = 0x7fffffff;
                for (int c = 0; c < 16; c++) {
                    const int e = ingest_err(pal, c, px);
                    if (e < be) {
                        be = e;
                        best = c;
                    }
                }
                pattern = colour = best;
            } else {
                if (k == 0 || (2 * k) / 7 != (2 * k - 2) / 7) {   // the first dot of this pixel opens screen byte b
                    const int b = (2 * k) / 7;
                    long err0 = 0, err1 = 0;
                    for (int kk = k; kk < 140 && (2 * kk) / 7 == b; kk++) {
                        IngestPixel u;
                        value(kk, u);
                        const int w = (2 * kk + 1) / 7 == b ? 2 : 1;
                        int b0 = 0x7fffffff, b1 = 0x7fffffff;
                        for (int i = 0; i < 4; i++) {
                            const int e0 = ingest_err(pal, colour4[0][i], u), e1 = ingest_err(pal, colour4[1][i], u);
                            b0 = e0 < b0 ? e0 : b0;
                            b1 = e1 < b1 ? e1 : b1;
                        }
                        err0 += (long)w * b0;
                        err1 += (long)w * b1;
                    }
                    pb = err1 < err0 ? 1 : 0;
                }
                value(k, px);
                int best = 0, be = 0x7fffffff;
                for (int i = 0; i < 4; i++) {
                    const int e = ingest_err(pal, colour4[pb][i], px);
                    if (e < be) {
                        be = e;
                        best = i;
                    }
                }
                pattern = best | (pb << 2);
                colour = colour4[pb][best];
            }
            patt[y][k] = (uint8_t)pattern;
            const int e[3] = {px.r - pal[3 * colour], px.g - pal[3 * colour + 1], px.b - pal[3 * colour + 2]};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                ring[y][k & 7][c] = 0;   // the slot is pixel k + 8's from now on
                if (k + 1 < 140) ring[y][(k + 1) & 7][c] += 7 * e[c];
                if (y + 1 < 192) {
                    if (k > 0) ring[y + 1][(k - 1) & 7][c] += 3 * e[c];
                    ring[y + 1][k & 7][c] += 5 * e[c];
                    if (k + 1 < 140) ring[y + 1][(k + 1) & 7][c] += e[c];
                }
            }
        }
        __syncthreads();
    }
    // the row's bytes (this thread wrote every pattern of its row itself)
    const int base = y_to_offset(y);
    if (MODE == kDHGR) {
        for (int j = 0; j < 80; j++) {
            int v = 0;
            for (int i = 0; i < 7; i++) {
                const int X = 7 * j + i;
                v |= ((patt[y][X >> 2] >> (X & 3)) & 1) << i;
            }
            ((j & 1) ? main_mem : aux_mem)[f * 8192 + base + (j >> 1)] = (uint8_t)v;
        }
    } else {
        for (int b = 0; b < 40; b++) {
            // the byte's palette bit is that of the first pixel whose first dot lies in it
            int v = ((patt[y][(7 * b + 1) >> 1] >> 2) & 1) << 7;
            for (int i = 0; i < 7; i++) {
                const int X = 7 * b + i;
                v |= ((patt[y][X >> 1] >> (X & 1)) & 1) << i;
            }
            main_mem[f * 8192 + base + b] = (uint8_t)v;
        }
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
    if (!rc && dither == IIV_DITHER_DIFFUSION) {
        if (mode == kDHGR)
            hipLaunchKernelGGL(ingest_diffusion_kernel<kDHGR>, dim3((unsigned)n), dim3(192), 0, st, d_rgb, d_pal, d_main, d_aux);
        else
            hipLaunchKernelGGL(ingest_diffusion_kernel<kHGR>, dim3((unsigned)n), dim3(192), 0, st, d_rgb, d_pal, d_main, d_aux);
        rc = hip_check(hipGetLastError(), "ingest_diffusion_kernel launch");
    } else if (!rc) {
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
        (mode == IIV_DHGR && !d_aux) || dither < 0 || dither > IIV_DITHER_DIFFUSION)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_frames_to_memory_maps: bad argument");
    if (n_frames == 0) return IIV_OK;
    return iiv::frames_to_memory_maps(mode, palette_rgb, n_frames, d_rgb, dither, d_main, d_aux, (hipStream_t)stream);
}
