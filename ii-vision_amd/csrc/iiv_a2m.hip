// iiv_a2m.hip -- byte emission of the opcode stream (".a2m" framing), batched.
// Reference: movie.Movie.emit_stream / done (transcoder/movie.py:113-161),
// opcodes.Header / BaseTick / Ack / Terminate (opcodes.py:64-139),
// machine.Machine.emit (machine.py:11-25).
//
// The reference emits byte by byte through generators; the layout is closed form:
// a 7-byte header, 7 bytes per tick opcode, and a 4-byte ACK whenever the stream
// position reaches 2044 mod 2048 -- i.e. after opcode 290, then every 292 opcodes
// (movie.py:139-148).  One thread per opcode writes its 7 bytes (and the ACK that
// follows it, if any) straight to its final position.
#include "iiv_host.h"

namespace iiv {

// stream position at which tick opcode k starts
__host__ __device__ static inline size_t tick_offset(long k)
{
    if (k < 291) return 7 + 7 * (size_t)k;
    long g = (k - 291) / 292, r = (k - 291) % 292;
    return 2048 * (size_t)(1 + g) + 7 * (size_t)r;
}

// opcodes emitted before max_bytes_out stops the stream (movie.py:132-134)
static long emitted_ops(long n_ops, long max_bytes_out)
{
    if (max_bytes_out <= 0) return n_ops;
    long lo = 0, hi = n_ops;  // largest n with tick_offset(k) < max for all k < n
    while (lo < hi) {
        long mid = (lo + hi + 1) / 2;
        if ((long)tick_offset(mid - 1) < max_bytes_out)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

static size_t stream_size(long n_emit)
{
    size_t pos = tick_offset(n_emit) + 2;  // + Terminate
    return pos + (2048 - pos % 2048);      // done() pads to the next 2 KiB boundary (a full frame if already on it)
}

// Opcodes [first_op, first_op + n_ops) of every stream -> their bytes, written at
// out + s * out_stride + (stream position - out_base).  ops / ticks point at opcode first_op of
// stream 0.  ticks == nullptr: every opcode carries const_tick.  header / finish: also write
// the 7-byte header (first_op must be 0) / Terminate + zero padding up to `total`.
// A tick that is not an even number in 4..66 or a page outside 32..63 cannot be an opcode of
// the player (opcodes.py:11-26): the lookup index is clamped and *err is set.
__global__ __launch_bounds__(256) void emit_kernel(int mode, long first_op, long n_ops, const uint8_t *__restrict__ ops,
                                                   size_t ops_stride, const uint8_t *__restrict__ ticks,
                                                   size_t ticks_stride, uint32_t const_tick,
                                                   const uint16_t *__restrict__ tick_addr, uint32_t ack_addr,
                                                   uint32_t term_addr, uint8_t *__restrict__ out, size_t out_stride,
                                                   size_t out_base, int header, int finish, size_t total,
                                                   int *__restrict__ err)
{
    const int s = blockIdx.y;
    uint8_t *o = out + (size_t)s * out_stride - out_base;
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < n_ops) {
        const uint8_t *q = ops + (size_t)s * ops_stride + (size_t)k * 6;
        const uint32_t tick = ticks ? ticks[(size_t)s * ticks_stride + k] : const_tick;
        uint32_t ti = (tick - 4) >> 1, pi = (uint32_t)q[0] - 32u;
        if (tick < 4 || tick > 66 || (tick & 1) || pi > 31u) {
            if (err) *err = 1;
            ti = ti > 31u ? 0u : ti;
            pi = pi > 31u ? 0u : pi;
        }
        const uint32_t a = tick_addr[ti * 32 + pi];
        size_t p = tick_offset(first_op + k);
        o[p + 0] = (uint8_t)(a >> 8);  // emit_command (opcodes.py:49-53)
        o[p + 1] = (uint8_t)a;
        o[p + 2] = q[1];               // content, then the 4 offsets (opcodes.py:136-138)
        o[p + 3] = q[2];
        o[p + 4] = q[3];
        o[p + 5] = q[4];
        o[p + 6] = q[5];
        if ((p + 7) % 2048 >= 2044) {
            // ACK: DHGR flips the bank first (movie.py:143-147); ack i carries the bank after i+1 flips
            const long i = (first_op + k - 290) / 292;
            const bool aux = mode == kDHGR && ((i + 1) & 1);
            o[p + 7] = (uint8_t)(ack_addr >> 8);
            o[p + 8] = (uint8_t)ack_addr;
            o[p + 9] = aux ? 0x55 : 0x54;
            o[p + 10] = 0xff;
        }
    }
    if (k == 0 && header) {
        for (int i = 0; i < 6; i++) o[i] = 0xff;  // Header (opcodes.py:77-90)
        o[6] = (uint8_t)mode;
    }
    if (finish) {
        const size_t pt = tick_offset(first_op + n_ops);
        if (k == 0) {
            o[pt] = (uint8_t)(term_addr >> 8);  // Terminate
            o[pt + 1] = (uint8_t)term_addr;
        }
        // zero padding after Terminate (movie.py:159-161), spread over the grid
        for (size_t i = pt + 2 + (size_t)k; i < total; i += (size_t)gridDim.x * 256) o[i] = 0;
    }
}

int emit_stream(int mode, int n_streams, long n_ops, const uint8_t *d_ops, const uint8_t *d_ticks,
                const uint16_t tick_addr[1024], uint16_t ack_addr, uint16_t term_addr, long max_bytes_out,
                uint8_t *d_out, size_t out_stride, size_t *out_len, hipStream_t st)
{
    const long n_emit = emitted_ops(n_ops, max_bytes_out);
    const size_t total = stream_size(n_emit);
    if (out_len) *out_len = total;
    if (!d_out) return IIV_OK;  // size query
    if (out_stride < total) return set_error(IIV_ERR_INVALID, "iiv_emit_stream: out_stride %zu < %zu", out_stride, total);
    uint16_t *d_addr = nullptr;
    IIV_HIP(hipMalloc(&d_addr, 1024 * sizeof(uint16_t) + sizeof(int)));
    int *d_err = reinterpret_cast<int *>(d_addr + 1024);
    int h_err = 0;
    int rc = hip_check(hipMemcpyAsync(d_addr, tick_addr, 1024 * sizeof(uint16_t), hipMemcpyHostToDevice, st), "copy addr");
    if (!rc) rc = hip_check(hipMemsetAsync(d_err, 0, sizeof(int), st), "clear flag");
    if (!rc) {
        dim3 grid((unsigned)((n_emit + 255) / 256 > 0 ? (n_emit + 255) / 256 : 1), (unsigned)n_streams);
        hipLaunchKernelGGL(emit_kernel, grid, dim3(256), 0, st, mode, 0L, n_emit, d_ops, (size_t)n_ops * 6, d_ticks,
                           (size_t)n_ops, 0u, d_addr, (uint32_t)ack_addr, (uint32_t)term_addr, d_out, out_stride, (size_t)0, 1,
                           1, total, d_err);
        rc = hip_check(hipGetLastError(), "emit_kernel launch");
    }
    if (!rc) rc = hip_check(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st), "read flag");
    if (!rc) rc = hip_check(hipStreamSynchronize(st), "emit sync");  // tick_addr is caller memory; d_addr freed below
    (void)hipFree(d_addr);
    if (!rc && h_err)
        rc = set_error(IIV_ERR_INVALID, "iiv_emit_stream: a tick is not an even number in 4..66, or a page is outside 32..63");
    return rc;
}

// byte range of opcodes [first_op, first_op + n_ops) in a stream (the header belongs to opcode 0)
void emit_chunk_range(long first_op, long n_ops, size_t *first_byte, size_t *n_bytes)
{
    const size_t b0 = first_op == 0 ? 0 : tick_offset(first_op), b1 = tick_offset(first_op + n_ops);
    if (first_byte) *first_byte = b0;
    if (n_bytes) *n_bytes = b1 - b0;
}

int emit_chunk(int mode, int n_streams, long first_op, long n_ops, const uint8_t *d_ops, size_t ops_stride,
               const uint8_t *d_ticks, size_t ticks_stride, int const_tick, const uint16_t *d_tick_addr, uint16_t ack_addr,
               uint8_t *d_out, size_t out_stride, int *d_err, hipStream_t st)
{
    size_t b0, nb;
    emit_chunk_range(first_op, n_ops, &b0, &nb);
    if (out_stride < nb) return set_error(IIV_ERR_INVALID, "iiv_emit_chunk: out_stride %zu < %zu", out_stride, nb);
    if (n_ops == 0 && first_op != 0) return IIV_OK;
    dim3 grid((unsigned)((n_ops + 255) / 256 > 0 ? (n_ops + 255) / 256 : 1), (unsigned)n_streams);
    hipLaunchKernelGGL(emit_kernel, grid, dim3(256), 0, st, mode, first_op, n_ops, d_ops, ops_stride, d_ticks, ticks_stride,
                       (uint32_t)const_tick, d_tick_addr, (uint32_t)ack_addr, 0u, d_out, out_stride, b0, first_op == 0 ? 1 : 0,
                       0, (size_t)0, d_err);
    return hip_check(hipGetLastError(), "emit_kernel launch");
}

}  // namespace iiv

extern "C" int iiv_emit_stream(int mode, int n_streams, long n_ops, const uint8_t *d_ops, const uint8_t *d_ticks,
                               const uint16_t tick_addr[1024], uint16_t ack_addr, uint16_t terminate_addr,
                               long max_bytes_out, uint8_t *d_out, size_t out_stride, size_t *out_len, void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || n_streams <= 0 || n_ops < 0 || !tick_addr ||
        (d_out && n_ops > 0 && (!d_ops || !d_ticks)))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_emit_stream: bad argument");
    return iiv::emit_stream(mode, n_streams, n_ops, d_ops, d_ticks, tick_addr, ack_addr, terminate_addr, max_bytes_out,
                            d_out, out_stride, out_len, (hipStream_t)stream);
}

extern "C" int iiv_emit_chunk(int mode, int n_streams, long first_op, long n_ops, const uint8_t *d_ops, size_t ops_stride,
                              const uint8_t *d_ticks, size_t ticks_stride, int const_tick, const uint16_t *d_tick_addr,
                              uint16_t ack_addr, uint8_t *d_out, size_t out_stride, size_t *first_byte, size_t *n_bytes,
                              int *d_err, void *stream)
{
    if ((mode != IIV_HGR && mode != IIV_DHGR) || n_streams <= 0 || n_ops < 0 || first_op < 0)
        return iiv::set_error(IIV_ERR_INVALID, "iiv_emit_chunk: bad argument");
    iiv::emit_chunk_range(first_op, n_ops, first_byte, n_bytes);
    if (!d_out) return IIV_OK;  // size query
    if (!d_ops || !d_tick_addr || (!d_ticks && (const_tick < 4 || const_tick > 66 || (const_tick & 1))))
        return iiv::set_error(IIV_ERR_INVALID, "iiv_emit_chunk: bad argument");
    return iiv::emit_chunk(mode, n_streams, first_op, n_ops, d_ops, ops_stride, d_ticks, ticks_stride, const_tick,
                           d_tick_addr, ack_addr, d_out, out_stride, d_err, (hipStream_t)stream);
}
