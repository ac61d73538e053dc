// marchbw.hip - does a ROW-MARCHING access pattern stream on this chip? (tool, not product)
//
// The multi-stage passes (two red-black iterations, four Jacobi sweeps, K3 + K4) pay for their depth with halo rows: a 4-row register tile
// of a 4-stage pass requests 12 rows.  A wave that marches down a strip of L rows with the stages as a software pipeline requests L + 8.
// Two earlier marching kernels lost against the tile kernels (single-stage Jacobi sweeps: 90-100 against 75 us) - the question here is what
// the memory system makes of the pattern itself, with the traffic of the red-black pair pass: 4 planes in (p.current, p.next, v.x, v.y),
// 2 planes out, every wave a column of NC = 60 * N cells, prefetch ring of PF rows, strips of L rows.
//   hipcc -O3 --offload-arch=gfx950 tools/marchbw.hip -o gpurun_out/marchbw && gpurun_out/marchbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int N> struct V;
template <> struct V<2> { using t = float2; };
template <> struct V<4> { using t = float4; };
template <int N> __device__ __forceinline__ float vsum(const typename V<N>::t &v);
template <> __device__ __forceinline__ float vsum<2>(const float2 &v) { return v.x + v.y; }
template <> __device__ __forceinline__ float vsum<4>(const float4 &v) { return (v.x + v.y) + (v.z + v.w); }

// MAP 0: strip id = blockIdx.x, columns fastest (consecutive workgroups = neighbouring columns of one strip row, dealt round-robin to the XCDs)
// MAP 1: XCD x takes the strip rows s = x, x + 8, ...: blockIdx.x = k * 8 + x, k = (s / 8) * ncol + col
template <int N, int PF, int WAVES, int MAP, int OWN = 60, bool CODE = false>
__global__ __launch_bounds__(64 * WAVES) void k_march(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c, const float *__restrict__ d,
                                                      float *__restrict__ o1, float *__restrict__ o2, int X, int Y, int P, int L, int ncol, int nstrips, const unsigned short *__restrict__ code)
{
    using Q = typename V<N>::t;
    constexpr int R = PF + 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    int col, strip;
    if (MAP == 0) { const int id = blockIdx.x; strip = id / ncol; col = (id - strip * ncol) * WAVES + wave; }
    else { const int x = blockIdx.x & 7, k = blockIdx.x >> 3; const int sg = k / ncol; strip = sg * 8 + x; col = (k - sg * ncol) * WAVES + wave; }
    if (strip >= nstrips) return;
    const int nu = X / N;
    constexpr int HL = (64 - OWN) / 2;          // OWN = 60: 2 halo lanes per side (as the pair pass on lanes of 2 cells); OWN = 64: aligned columns, no overlap
    int q = col * OWN - HL + lane;
    const bool owner = lane >= HL && lane < 64 - HL && q >= 0 && q < nu;
    q = q < 0 ? 0 : (q > nu - 1 ? nu - 1 : q);
    if (col * OWN >= nu) return;
    const int j0 = strip * L, W = L + 8;
    Q ra[R], rb[R], rc[R], rd[R];
    unsigned short rcode[R];
#pragma unroll
    for (int t = 0; t < R; ++t) rcode[t] = 0;
    auto row = [&](int t) { int j = j0 - 4 + t; j = j < 0 ? 0 : (j > Y - 1 ? Y - 1 : j); return (size_t)j * P; };
#pragma unroll
    for (int t = 0; t < PF; ++t) {
        const size_t r = row(t);
        ra[t] = *(const Q *)(a + r + q * N); rb[t] = *(const Q *)(b + r + q * N); rc[t] = *(const Q *)(c + r + q * N); rd[t] = *(const Q *)(d + r + q * N);
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc = 0.f;
    for (int t0 = 0; t0 < W; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int t = t0 + u;
            {   // prefetch row t + PF into the slot that row t - 1 left
                const size_t r = row(t + PF);
                const int s = (u + PF) % R;
                ra[s] = *(const Q *)(a + r + q * N); rb[s] = *(const Q *)(b + r + q * N); rc[s] = *(const Q *)(c + r + q * N); rd[s] = *(const Q *)(d + r + q * N);
                if (CODE) rcode[s] = *(const unsigned short *)((const char *)code + r + q * N);
            }
            __builtin_amdgcn_sched_barrier(0);    // the loads stay HERE: left alone, the scheduler sinks them to their first use and the ring is gone
            const float s1 = vsum<N>(ra[u]) + vsum<N>(rc[u]) + (CODE ? (float)rcode[u] : 0.f), s2 = vsum<N>(rb[u]) + vsum<N>(rd[u]);
            acc = acc * 0.5f + s1;            // (a dependency chain down the strip, like the pipeline stages)
            const int w = t - 6;
            __builtin_amdgcn_sched_barrier(0);
            if (w >= 4 && w < W - 4 && owner) {
                const int j = j0 - 4 + w;
                if (j < Y) {
                    Q x1, x2;
                    if constexpr (N == 2) { x1 = make_float2(acc, s2); x2 = make_float2(s2, acc); }
                    else { x1 = make_float4(acc, s2, s1, acc); x2 = make_float4(s2, acc, s2, s1); }
                    *(Q *)(o1 + (size_t)j * P + q * N) = x1;
                    *(Q *)(o2 + (size_t)j * P + q * N) = x2;
                }
            }
        }
    }
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

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int X = 8192, Y = 4096, P = 8192;
    const size_t n = (size_t)P * Y;
    float *a, *b, *vxy, *o1, *o2;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&vxy, n * 8)); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(vxy, 0, n * 8)); CK(hipMemset(o1, 0, n * 4)); CK(hipMemset(o2, 0, n * 4));
    float *c = vxy, *d = vxy + n;       // (two planes of separate halves: the row-interleaved [row][channel][x] layout changes nothing for the memory system)
    const double alg = 6.0 * X * Y * 4 / 1e9;
    printf("marching stream, 8192 x 4096 cells, 4 planes in + 2 out: %.1f MB algorithmic per pass\n", alg * 1e3);
#define RUN(N, PF, WAVES, MAP, L) RUNX(N, PF, WAVES, MAP, L, 60, false)
#define RUNX(N, PF, WAVES, MAP, L, OWN, CODE) { \
        const int nwcol = (X / N + OWN - 1) / OWN, ncol = (nwcol + WAVES - 1) / WAVES, nstrips = (Y + L - 1) / L; \
        const int blocks = MAP == 0 ? ncol * nstrips : ncol * ((nstrips + 7) / 8) * 8; \
        float ms = time_it([&] { k_march<N, PF, WAVES, MAP, OWN, CODE><<<blocks, 64 * WAVES>>>(a, b, c, d, o1, o2, X, Y, P, L, ncol, nstrips, (const unsigned short *)o2); }, reps); \
        CK(hipGetLastError()); \
        const double act = alg * (4.0 * (L + 8) / L + 2.0) / 6.0; \
        printf("N=%d PF=%d waves/wg=%d map=%d L=%3d own=%d code=%d : %7.1f us  alg %5.0f GB/s  requested %5.0f GB/s  (%d workgroups)\n", N, PF, WAVES, MAP, L, OWN, (int)CODE, ms * 1e3, alg / (ms * 1e-3), act / (ms * 1e-3), blocks); }
    // aligned columns (no halo lanes), the code plane as a fifth input
    RUN(2, 3, 4, 0, 32) RUN(2, 3, 4, 1, 32) RUNX(2, 3, 4, 0, 32, 64, false) RUNX(2, 3, 4, 0, 32, 60, true) RUNX(2, 3, 4, 1, 32, 64, false) RUNX(4, 3, 4, 0, 32, 62, false) RUNX(4, 3, 4, 0, 32, 64, false) RUNX(2, 3, 4, 0, 64, 64, false)
    if (argc > 2) return 0;
    // prefetch depth
    RUN(2, 1, 4, 0, 32) RUN(2, 2, 4, 0, 32) RUN(2, 3, 4, 0, 32) RUN(2, 5, 4, 0, 32) RUN(2, 7, 4, 0, 32)
    // strip height
    RUN(2, 3, 4, 0, 16) RUN(2, 3, 4, 0, 24) RUN(2, 3, 4, 0, 48) RUN(2, 3, 4, 0, 64) RUN(2, 3, 4, 0, 128) RUN(2, 3, 4, 0, 256)
    // workgroup shape and XCD mapping
    RUN(2, 3, 1, 0, 32) RUN(2, 3, 1, 1, 32) RUN(2, 3, 4, 1, 32) RUN(2, 3, 1, 0, 64) RUN(2, 3, 1, 1, 64) RUN(2, 3, 4, 1, 64) RUN(2, 3, 2, 0, 32) RUN(2, 3, 8, 0, 32)
    // 16-byte lanes
    RUN(4, 1, 4, 0, 32) RUN(4, 2, 4, 0, 32) RUN(4, 3, 4, 0, 32) RUN(4, 3, 4, 0, 64) RUN(4, 3, 1, 1, 32) RUN(4, 5, 4, 0, 32)
    RUN(2, 5, 4, 0, 64) RUN(2, 5, 1, 1, 64) RUN(2, 7, 4, 0, 64)
    return 0;
}
