// mallbw.hip - does a producer -> consumer pair of streaming kernels run faster when executed band by band, so that the
// band written by the producer is still in the 256 MiB Infinity Cache (MALL) when the consumer reads it?  (tool, not product)
//   hipcc -O3 --offload-arch=gfx950 tools/mallbw.hip -o tools/mallbw.bin && tools/mallbw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_produce(const float4 *__restrict__ x, float4 *__restrict__ y, size_t n)   // y = f(x)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 v = x[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; y[i] = v; }
}
__global__ __launch_bounds__(256) void k_consume(const float4 *__restrict__ x, const float4 *__restrict__ y, float4 *__restrict__ z, size_t n)   // z = g(x, y)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 a = x[i], b = y[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; z[i] = a; }
}

int main()
{
    const size_t n = (size_t)8192 * 4096 * 2 / 4;          // one 2-channel f32 field of the res-4096 grid, in float4: 268 MB
    float4 *x, *y, *z;
    CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMalloc(&z, n * 16));
    CK(hipMemset(x, 0, n * 16)); CK(hipMemset(y, 0, n * 16)); CK(hipMemset(z, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("field %.0f MB; traffic per pass: produce 2 fields, consume 3 fields = %.0f MB\n", n * 16 / 1e6, 5 * n * 16 / 1e6);
    for (int nb : {1, 2, 4, 8, 16, 32, 64}) {
        const size_t band = n / nb;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it)
                for (int b = 0; b < nb; ++b) {
                    const size_t o = b * band;
                    k_produce<<<(band + 255) / 256, 256>>>(x + o, y + o, band);
                    k_consume<<<(band + 255) / 256, 256>>>(x + o, y + o, z + o, band);
                }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("bands %3d (%6.1f MB per field band): %8.1f us per pass  -> %6.0f GB/s algorithmic\n", nb, band * 16 / 1e6, ms / 10 * 1e3, 5 * n * 16 / (ms / 10 * 1e-3) / 1e9);
    }
    return 0;
}
