// What this box's HBM moves with plain streaming kernels: read (sum), write (fill), copy (read + write), 4 GiB each,
// 16 bytes per lane per access, grid-stride, several launch shapes; the best of each is printed last.
// (tools/hbm_copy_bench.py measures the same through torch's own kernels; bench.py's HBM_MEASURED_* constants quote a run.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t v4 __attribute__((ext_vector_type(4)));   // (the nontemporal builtins want a native vector type)

template <int U>
__global__ void copy_k(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], b + i + u * stride);
    }
    for (; i < n; i += stride) b[i] = a[i];
}
template <int U>
__global__ void read_k(const v4 *__restrict__ a, uint32_t *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int U>
__global__ void fill_k(v4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const v4 v = {1, 2, 3, 4};
    for (; i + (U - 1) * stride < n; i += U * stride)
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v, b + i + u * stride);
}

int main()
{
    const size_t bytes = (size_t)4 << 30, n = bytes / 16;
    v4 *a, *b;
    uint32_t *o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best[3] = {0, 0, 0};
    auto time = [&](const char *name, int which, double moved, int grid, int block, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double gbs = moved / ms * 1e-6;
        printf("%-6s grid %6d x %4d  %7.3f ms  %7.1f GB/s\n", name, grid, block, ms, gbs);
        if (gbs > best[which]) best[which] = gbs;
    };
    for (int block : {256, 512, 1024})
        for (int per_cu : {2, 4, 8, 16, 32}) {
            const int grid = 256 * per_cu * 256 / block;
            if (grid < 256) continue;
            time("copy", 0, 2.0 * bytes, grid, block, [&] { hipLaunchKernelGGL(copy_k<4>, dim3(grid), dim3(block), 0, 0, a, b, n); });
            time("read", 1, 1.0 * bytes, grid, block, [&] { hipLaunchKernelGGL(read_k<4>, dim3(grid), dim3(block), 0, 0, a, o, n); });
            time("fill", 2, 1.0 * bytes, grid, block, [&] { hipLaunchKernelGGL(fill_k<4>, dim3(grid), dim3(block), 0, 0, b, n); });
        }
    printf("best: copy (read + write bytes) %.1f GB/s, read %.1f GB/s, write %.1f GB/s; err=%d\n", best[0], best[1], best[2], (int)hipGetLastError());
    return 0;
}
