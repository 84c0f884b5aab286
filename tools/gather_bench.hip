// Microbenchmark: how many random 2-byte (or 4-byte) table lookups per second can one
// MI355X sustain from a table of a given size?  This is the structural ceiling of the
// greedy step (256 store-table lookups per opcode).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gather_bench tools/gather_bench.hip && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

// SHARE: lanes in groups of SHARE consecutive lanes look up neighbouring entries (same line);
// ACTIVE: lanes >= ACTIVE are switched off for the loads.
template <typename T, int G, int SHARE = 1, int ACTIVE = 64>
__global__ __launch_bounds__(64, 4) void gather_kernel(const T *__restrict__ table, uint32_t mask, int iters,
                                                       uint32_t *__restrict__ sink)
{
    uint32_t s = (blockIdx.x * 64 + threadIdx.x / SHARE) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            s = s * 1664525u + 1013904223u;
            v[g] = 0;
            if (ACTIVE == 64 || threadIdx.x < ACTIVE)
                v[g] = table[(((s >> 8) & mask) & ~(uint32_t)(SHARE - 1)) | (threadIdx.x % SHARE)];
        }
#pragma unroll
        for (int g = 0; g < G; g++) acc += v[g];
        s ^= acc & 1u;  // next batch depends on this one (like a chunk depends on the previous)
    }
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <typename T, int G, int SHARE = 1, int ACTIVE = 64> static void run(size_t bytes, int waves, int iters, const char *name)
{
    T *d;
    uint32_t *sink;
    hipMalloc(&d, bytes);
    hipMemset(d, 1, bytes);
    hipMalloc(&sink, (size_t)waves * 64 * 4);
    uint32_t mask = (uint32_t)(bytes / sizeof(T) - 1);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((gather_kernel<T, G, SHARE, ACTIVE>), dim3(waves), dim3(64), 0, 0, d, mask, 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((gather_kernel<T, G, SHARE, ACTIVE>), dim3(waves), dim3(64), 0, 0, d, mask, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double lookups = (double)waves * 64 * iters * G;
    printf("%-6s share %d active %2d table %8.0f KiB  waves %6d  batch %2d: %8.3f ms  %7.1f G lookups/s  (%.3f per CU-cycle @2.4GHz)\n", name, SHARE, ACTIVE,
           bytes / 1024.0, waves, G, ms, lookups / ms * 1e-6, lookups / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
    hipFree(sink);
}

int main()
{
    // L1-resident .. L2-resident tables: is an L1 hit cheaper than an L2 hit for a divergent gather?
    for (size_t kib : {8, 16, 32, 64, 256, 1024, 4096}) run<uint16_t, 32>(kib << 10, 4096, 400, "u16");
    for (size_t kib : {16, 4096}) run<uint32_t, 32>(kib << 10, 4096, 400, "u32");
    run<uint16_t, 32, 2>(4 << 20, 4096, 400, "u16");
    run<uint16_t, 32, 4>(4 << 20, 4096, 400, "u16");
    run<uint16_t, 32, 1, 32>(4 << 20, 4096, 400, "u16");
    run<uint16_t, 32>(4 << 20, 256, 400, "u16");
    run<uint16_t, 32>(8 << 20, 4096, 400, "u16");
    run<uint16_t, 32>(16 << 20, 4096, 400, "u16");
    return 0;
}
