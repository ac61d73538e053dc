// membw.hip - streaming-read / copy ceilings on this GPU for the access shapes the step kernels use (tool, not product).
//   hipcc -O3 --offload-arch=gfx950 tools/membw.hip -o gpurun_out/membw && gpurun_out/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one quad per lane per row, U rows per lane in flight (the register-tile shape): block covers 256 quads x U rows
template <int U>
__global__ __launch_bounds__(256) void k_read_tile(const float4 *__restrict__ a, float *out, size_t rowq, size_t nrows)
{
    size_t bx = blockIdx.x % (rowq / 256), by = blockIdx.x / (rowq / 256);
    size_t q = bx * 256 + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[(by * U + u) * rowq + q];
    float s = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    if (s == 1.2345f) out[0] = s;
}
// grid-stride, U independent loads per iteration
template <int U>
__global__ __launch_bounds__(256) void k_read_stride(const float4 *__restrict__ a, float *out, size_t n)
{
    size_t stride = (size_t)gridDim.x * 256;
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { size_t k = i + u * stride; v[u] = k < n ? a[k] : make_float4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 1.2345f) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void k_copy_tile(const float4 *__restrict__ a, float4 *__restrict__ b, size_t rowq, size_t nrows)
{
    size_t bx = blockIdx.x % (rowq / 256), by = blockIdx.x / (rowq / 256);
    size_t q = bx * 256 + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[(by * U + u) * rowq + q];
#pragma unroll
    for (int u = 0; u < U; ++u) b[(by * U + u) * rowq + q] = v[u];
}

template <typename F>
static float time_it(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t rowq = 8192 / 4 * 2;           // a 2-channel row of the res-4096 grid, in quads
    const size_t nrows = 4096 * 2;              // 2 fields -> 536 MB (beyond the 256 MiB MALL)
    const size_t n = rowq * nrows;
    float4 *a, *b; float *out;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16));
    const double GB = n * 16 / 1e9;
    printf("buffer %.1f MB\n", GB * 1e3);
#define TILE(U) { float ms = time_it([&] { k_read_tile<U><<<(rowq / 256) * (nrows / U), 256>>>(a, out, rowq, nrows); }, 20); \
                  printf("read  tile U=%d            : %7.1f us  %6.0f GB/s\n", U, ms * 1e3, GB / (ms * 1e-3)); }
    TILE(1) TILE(2) TILE(4) TILE(8) TILE(16)
#define STR(U, NB) { float ms = time_it([&] { k_read_stride<U><<<NB, 256>>>(a, out, n); }, 20); \
                  printf("read  stride U=%d blocks=%5d: %7.1f us  %6.0f GB/s\n", U, NB, ms * 1e3, GB / (ms * 1e-3)); }
    STR(1, 2048) STR(2, 2048) STR(4, 2048) STR(8, 2048) STR(4, 1024) STR(4, 4096) STR(4, 8192) STR(8, 4096) STR(4, 16384)
#define CPY(U) { float ms = time_it([&] { k_copy_tile<U><<<(rowq / 256) * (nrows / U), 256>>>(a, b, rowq, nrows); }, 20); \
                  printf("copy  tile U=%d            : %7.1f us  %6.0f GB/s (read+write)\n", U, ms * 1e3, 2 * GB / (ms * 1e-3)); }
    CPY(1) CPY(2) CPY(4) CPY(8)
    { float ms = time_it([&] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); }, 20);
      printf("hipMemcpy D2D              : %7.1f us  %6.0f GB/s (read+write)\n", ms * 1e3, 2 * GB / (ms * 1e-3)); }
    return 0;
}
