// Microbenchmark 2: does a cache-policy modifier (sc0 / nt / sc1) change the cost of a
// fully divergent 2-byte gather?  (raw buffer loads, aux bits: 1 = sc0, 2 = nt, 16 = sc1)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int AUX, int G>
__global__ __launch_bounds__(64, 4) void gather_kernel(const uint16_t *__restrict__ table, uint32_t mask, int iters,
                                                       uint32_t bytes, uint32_t *__restrict__ sink)
{
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, (int)bytes, 0x00020000);
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            s = s * 1664525u + 1013904223u;
            v[g] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)(((s >> 8) & mask) * 2), 0, AUX);
        }
#pragma unroll
        for (int g = 0; g < G; g++) acc += v[g];
        s ^= acc & 1u;
    }
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int AUX> static void run(size_t bytes, int waves, int iters)
{
    uint16_t *d;
    uint32_t *sink;
    (void)hipMalloc(&d, bytes);
    (void)hipMemset(d, 1, bytes);
    (void)hipMalloc(&sink, (size_t)waves * 64 * 4);
    uint32_t mask = (uint32_t)(bytes / 2 - 1);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL((gather_kernel<AUX, 32>), dim3(waves), dim3(64), 0, 0, d, mask, 4, (uint32_t)bytes, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((gather_kernel<AUX, 32>), dim3(waves), dim3(64), 0, 0, d, mask, iters, (uint32_t)bytes, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    double lookups = (double)waves * 64 * iters * 32;
    printf("aux %2d table %5.1f MiB: %8.3f ms  %7.1f G lookups/s (%.3f per CU-cycle @2.4GHz) err=%d\n", AUX, bytes / 1048576.0, ms,
           lookups / ms * 1e-6, lookups / (ms * 1e-3) / 256 / 2.4e9, (int)hipGetLastError());
    (void)hipFree(d);
    (void)hipFree(sink);
}

int main()
{
    for (size_t mib : {4, 8, 16}) {
        run<0>(mib << 20, 4096, 200);
        run<1>(mib << 20, 4096, 200);
        run<2>(mib << 20, 4096, 200);
        run<3>(mib << 20, 4096, 200);
        run<16>(mib << 20, 4096, 200);
        run<17>(mib << 20, 4096, 200);
        run<18>(mib << 20, 4096, 200);
        run<19>(mib << 20, 4096, 200);
    }
    return 0;
}
