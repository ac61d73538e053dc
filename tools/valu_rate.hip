// valu_rate.hip - how many f32 VALU wave-instructions does one SIMD of this GPU issue per cycle?  (tool, not product)
//   hipcc -O3 -fno-slp-vectorize -ffp-contract=off --offload-arch=gfx950 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// (-fno-slp-vectorize matters: without it the vectoriser packs the "scalar" loop as well - round 3's "packed issues at half rate" was that artefact;
//  the library's own probes k_box_valu / k_box_valu_pk, built with its flags, are what bench.py reports: profiles/r5_valu_rate.txt)
// Chains of independent multiplies / adds (no FMA contraction), scalar f32 and packed (float2: v_pk_mul_f32 / v_pk_add_f32), at
// 1 .. 8 waves per SIMD.  Prints wave-instructions per SIMD per cycle (clock from hipDeviceProp) and the equivalent TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int U>
__global__ __launch_bounds__(256) void k_scalar(float *out, float a, float b, int iters)
{
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a + (float)(threadIdx.x + u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = x[u] * a;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = x[u] + b;
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u];
    if (s == 1.2345f) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void k_packed(float *out, float a, float b, int iters)
{
    v2f x[U];
    const v2f va = {a, a * 1.0001f}, vb = {b, b * 0.9999f};
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = va + (float)(threadIdx.x + u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = x[u] * va;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = x[u] + vb;
    }
    v2f s = {0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u];
    if (s.x + s.y == 1.2345f) out[0] = s.x;
}
template <int U>
__global__ __launch_bounds__(256) void k_f64(float *out, float a, double r, int iters)     // the f64-multiply division: cvt, mul_f64, cvt
{
    float x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = a + (float)(threadIdx.x + u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (float)((double)x[u] * r);
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += x[u];
    if (s == 1.2345f) out[0] = s;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, clock %.2f GHz (nominal)\n", p.name, cus, ghz);
    float *out; CK(hipMalloc(&out, 4));
    const int iters = 20000; constexpr int U = 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;      // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        for (int kind = 0; kind < 3; ++kind) {
            auto run = [&] {
                if (kind == 0) hipLaunchKernelGGL(k_scalar<U>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f, iters);
                else if (kind == 1) hipLaunchKernelGGL(k_packed<U>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1e-9f, iters);
                else hipLaunchKernelGGL(k_f64<U>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, 1.0000001, iters);
            };
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_wave = (double)iters * U * (kind == 2 ? 3 : 2);
            const double per_simd = instr_per_wave * wps;                      // wave-instructions one SIMD issued
            const double cyc = ms * 1e-3 * ghz * 1e9;
            const double flops = (double)blocks * 256 * iters * U * (kind == 1 ? 4 : 2) / (ms * 1e-3) / 1e12;
            printf("%d waves/SIMD  %-8s %8.3f ms  %.3f wave-instr / SIMD / cycle (%.2f cycles each)%s\n", wps, kind == 0 ? "scalar" : kind == 1 ? "packed" : "f64div",
                   ms, per_simd / cyc, cyc / per_simd, kind == 2 ? "" : (std::string("  ") + std::to_string(flops) + " TFLOP/s").c_str());
        }
    }
    return 0;
}
