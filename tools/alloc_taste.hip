// tools/alloc_taste.hip - does the streaming rate of a buffer depend on the allocation it lives in?  K buffers of 256 MiB from K hipMalloc calls, then K
// slices of one 8 GiB allocation: read rate (float4 loads, grid-stride) and copy rate of every one.   hipcc --offload-arch=gfx950 -O3 tools/alloc_taste.hip -o /tmp/alloc_taste
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_read(const float4 *p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *out = acc;
}
__global__ __launch_bounds__(256) void k_copy(float4 *d, const float4 *s, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
static float timed(hipEvent_t a, hipEvent_t b, int reps, void (*launch)(void *), void *arg)
{
    launch(arg); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch(arg);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
struct Arg { float4 *a, *b; size_t n; float *out; };
static void l_read(void *x) { Arg *g = (Arg *)x; hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, g->a, g->n, g->out); }
static void l_copy(void *x) { Arg *g = (Arg *)x; hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, g->b, g->a, g->n); }
int main()
{
    const size_t bytes = (size_t)256 << 20, n = bytes / 16; const int K = 24;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float *out; hipMalloc(&out, 4);
    std::vector<float4 *> bufs(K);
    for (auto &p : bufs) { if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(p, 0, bytes); }
    printf("separate hipMalloc, 256 MiB each (read GB/s | copy GB/s to the next buffer):\n");
    for (int k = 0; k < K; ++k) {
        Arg g{bufs[k], bufs[(k + 1) % K], n, out};
        const float r = timed(e0, e1, 20, l_read, &g), c = timed(e0, e1, 20, l_copy, &g);
        printf("  buf %2d @%p  read %6.0f  copy %6.0f\n", k, (void *)bufs[k], bytes / r / 1e6, 2.0 * bytes / c / 1e6);
    }
    for (auto p : bufs) hipFree(p);
    char *chunk; if (hipMalloc(&chunk, (size_t)K * bytes) != hipSuccess) { printf("chunk alloc failed\n"); return 1; }
    hipMemset(chunk, 0, (size_t)K * bytes);
    printf("slices of ONE %d x 256 MiB allocation:\n", K);
    for (int k = 0; k < K; ++k) {
        Arg g{(float4 *)(chunk + k * bytes), (float4 *)(chunk + ((k + 1) % K) * bytes), n, out};
        const float r = timed(e0, e1, 20, l_read, &g), c = timed(e0, e1, 20, l_copy, &g);
        printf("  slice %2d  read %6.0f  copy %6.0f\n", k, bytes / r / 1e6, 2.0 * bytes / c / 1e6);
    }
    return 0;
}
