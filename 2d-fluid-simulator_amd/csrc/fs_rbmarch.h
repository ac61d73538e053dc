// fs_rbmarch.h - the two-iteration red-black pass (fs_rbpair.h) as a ROW-MARCHING software pipeline.
//
// Reference: fs/pressure_updater.py:86-114 with n_iter = 2 (fs/fluid_simulator.py:76-78) = K7, odd, even, swap, K7, odd, even, swap
// (fs/boundary_condition.py:41-65 for K7).  fs_rbpair.h runs these four dependent half sweeps on a register tile of RT = 4 rows: the
// window is RT + 8 rows of four planes, i.e. every output row costs three rows of loads, 2.5 rows of source evaluation and 3.5 rows
// of relaxation, and a third of what it fetches is re-fetched by the neighbouring tiles (PMC: 796 MB for 572 algorithmic).
//
// Here a wave walks DOWN a strip of L rows.  Step t loads window row t + PF (window row w <-> local row j0 - 4 + w) and advances every
// stage by one row, each stage one row behind its producer:
//     source(t-1)  view(A)(t-1)  stage 1 = odd pass of iteration 1 (t-2)  stage 2 = even pass (t-3)  view(B')(t-4)
//     stage 3 = odd pass of iteration 2 (t-5)  stage 4 = even pass (t-6)  store row t-6
// so a strip requests L + 8 rows for L rows of output and evaluates every stage (L + 10) / L times per row - the halo cost of the
// tile form divided by L / 4.  The rows in flight live in register rings whose slot numbers are compile-time constants: the loop body
// is 12 steps (the least common multiple of the ring depths), L + 10 is a multiple of 12.
//
// Memory-level parallelism comes from the PREFETCH distance, not from a tile's worth of loads: the loads of row t + PF are issued at
// the top of step t and pinned there (sched_barrier - left alone, the scheduler sinks a load to its first use and the ring is gone);
// 16 waves per CU x PF rows x 5 loads of 512 B keep the memory system as busy as the tile kernels' up-front bursts (tools/marchbw.hip:
// the bare access pattern streams at 4.8-5.0 TB/s of algorithmic bytes on this chip, 0.78 of the same box's float4 copy).
//
// One kernel, no plain / boundary split: the recipe byte of a cell (fs_march.h lazy_value) and "not fluid" travel in ONE byte plane
// (`rbcode`: bits 0-6 the pressure recipe, bit 7 mask != 0), loaded like a field row; whether a row of the wave holds anything but
// fluid is a wave-uniform test of that word, and only such rows pay for the K7 views and the partial stores - per ROW of 120 cells,
// where the tile form classifies workgroups of 16 x 124.  Same arithmetic, same operation order as fs_rbpair.h (its rbp_relax,
// lv_bc_row, source_from): bit-identical results; the validity argument of the shrinking window is the same (rows w < s and
// w > W - 1 - s of stage s hold garbage that no valid row reads), as are the host-checked preconditions (fs_rbsor_pair_ok).
#pragma once
#include "fs_rbpair.h"

namespace fs {

template <int N> __device__ __forceinline__ unsigned rbm_sel_fluid(uint32_t cw)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((cw >> (8 * c + 7)) & 1u) ? 0u : (1u << c);
    return s;
}

// ---- loads the compiler does not schedule ---------------------------------------------------------------------------------------
// The rings only pay if a row is REQUESTED PF steps before it is used and nothing waits for it in between.  hipcc's own s_waitcnt
// insertion cannot be held to that in these loops (dozens of wave-uniform branches per step: after the merges it waits with vmcnt(0)
// right behind the loads it has just issued - measured: the same time per step for PF = 1 and 3, 0.59 of the wave cycles parked).  So the
// ring loads are inline asm (invisible to that pass) and the waits are ours:
//   * every step issues the SAME number of ring loads (rows past the window are clamped onto its last row: an L1 hit) - the counter
//     arithmetic below depends on it;
//   * mwait<K>() = s_waitcnt vmcnt(K) with K = loads issued SINCE the ones that are needed: memory operations retire in order, and the
//     stores in between (compiler-issued, a varying number) only make the wait stricter;
//   * the registers of the arriving row pass through an empty asm ("+v") behind the wait, so no use can move in front of it;
//   * an in-flight value lives in ONE 64-bit register pair from request to arrival (fs_f2, not two floats: a sub-register copy the
//     allocator might place behind the load would read the pair before it has landed), and the loop's back edge waits for everything
//     (vmcnt(0) once per 12 steps), so copies the allocator places there see landed data.
typedef float fs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mload2(fs_f2 &dst, const float *row, unsigned byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(uniform64((uint64_t)row)) : "memory");
#else
    dst = *(const fs_f2 *)((const char *)row + byte_off);
#endif
}
__device__ __forceinline__ void mload_u16(uint32_t &dst, const uint8_t *row, unsigned byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_ushort %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(uniform64((uint64_t)row)) : "memory");
#else
    dst = *(const uint16_t *)(row + byte_off);
#endif
}
template <int K> __device__ __forceinline__ void mwait()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(K) : "memory");
#endif
}
__device__ __forceinline__ LV<float, 2> lv_of(const fs_f2 &v) { LV<float, 2> r; r.a[0] = v.x; r.a[1] = v.y; return r; }
__device__ __forceinline__ fs_f2 f2_of(const LV<float, 2> &v) { fs_f2 r; r.x = v.a[0]; r.y = v.a[1]; return r; }
__device__ __forceinline__ const float *row_ptr(const void *f, const Grid &g, int C, int c, int j) { return (const float *)f + ((size_t)j * C + c) * g.P; }

template <int PF>
struct RbmState {
    static constexpr int DR = 3 + PF;      // raw rows t-2 .. t+PF
    using R = LV<float, 2>;
    fs_f2 PA[DR], VX[DR], VY[DR];          // requested PF steps ahead (mload2)
    fs_f2 PB[DR];                          // rows t-4 .. t-2+PF: raw until stage 1 has run on the row, then the state after iteration 1's odd pass
    R VA[6];                               // view(A), rows t-5 .. t-1
    R S2[6], S3[6];                        // Poisson source, rows t-6 .. t-1
    R P2[3];                               // after iteration 1's even pass, rows t-5 .. t-3
    R VB[3];                               // view(B'), rows t-6 .. t-4
    R P3[3];                               // after iteration 2's odd pass, rows t-7 .. t-5
    uint32_t CW[12];                       // rbcode words, rows t-6 .. t+PF
};

template <typename T, int N> __device__ __forceinline__ LV<T, N> lv_zero()
{
    LV<T, N> r;
#pragma unroll
    for (int c = 0; c < N; ++c) r.a[c] = (T)0;
    return r;
}

struct RbmArgs {
    const uint8_t *rbcode;
    void *C, *D;
    const void *A, *B, *v;
};

// U = t mod 12 (compile time), t = step (wave-uniform).  Ring loads per step: 5.
template <int U, int PF, int PAR0, int DM>
__device__ __forceinline__ void rbm_step(RbmState<PF> &s, const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, unsigned off4, unsigned off1, int j0, int je, int t, int W,
                                         const RbmArgs &a)
{
    constexpr int DR = 3 + PF, N = 2;
    constexpr unsigned ALL = 3u;
    using T = float;
    using R = LV<T, N>;
    {   // request window row t + PF of A, v and the code plane, row t + PF - 2 of B (rows past the window: its last row again - an L1 hit)
        const int wr = t + PF < W ? t + PF : W - 1, wb = t + PF - 2 < W ? t + PF - 2 : W - 1;
        const int j = clampy(g, j0 - 4 + wr), jb = clampy(g, j0 - 4 + wb);
        constexpr int sl = (U + PF) % DR;
        mload2(s.PA[sl], row_ptr(a.A, g, 1, 0, j), off4);
        mload2(s.VX[sl], row_ptr(a.v, g, 2, 0, j), off4);
        mload2(s.VY[sl], row_ptr(a.v, g, 2, 1, j), off4);
        mload_u16(s.CW[(U + PF) % 12], a.rbcode + (size_t)j * g.Pm, off1);
        mload2(s.PB[(U + PF - 2 + 12) % DR], row_ptr(a.B, g, 1, 0, jb), off4);
    }
    constexpr int r0 = U + 12;             // (U - k + 12) % depth: non-negative operands
    // rows t of A / v / code and t - 2 of B were requested PF steps ago: 5 PF loads have been issued since
    mwait<5 * PF>();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(s.PA[r0 % DR]), "+v"(s.VX[r0 % DR]), "+v"(s.VY[r0 % DR]), "+v"(s.CW[r0 % 12]), "+v"(s.PB[(r0 - 2) % DR]));
#endif
    {   // Poisson source of row t-1 from v rows t-2, t-1, t (fs/pressure_updater.py:25-38), once per cell for all four half sweeps
        const R xm = lv_of(s.VX[(r0 - 2) % DR]), xc = lv_of(s.VX[(r0 - 1) % DR]), xp = lv_of(s.VX[r0 % DR]);
        const R ym = lv_of(s.VY[(r0 - 2) % DR]), yc = lv_of(s.VY[(r0 - 1) % DR]), yp = lv_of(s.VY[r0 % DR]);
        const T xl = lv_left<T, N>(lm, xc), xr = lv_right<T, N>(lm, xc), yl = lv_left<T, N>(lm, yc), yr = lv_right<T, N>(lm, yc);
        R &s2 = s.S2[(r0 - 1) % 6], &s3 = s.S3[(r0 - 1) % 6];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const T xE = c == N - 1 ? xr : xc.a[c == N - 1 ? c : c + 1], xW = c == 0 ? xl : xc.a[c == 0 ? 0 : c - 1];
            const T yE = c == N - 1 ? yr : yc.a[c == N - 1 ? c : c + 1], yW = c == 0 ? yl : yc.a[c == 0 ? 0 : c - 1];
            source_from<DM>(k, xE, xW, yE, yW, xp.a[c], xm.a[c], yp.a[c], ym.a[c], s2.a[c], s3.a[c]);
        }
    }
    // view(A) of row t-1 (lv_bc_row returns the raw row unless the wave's row holds a target)
    s.VA[(r0 - 1) % 6] = lv_bc_row<T, N>(lm, lv_of(s.PA[(r0 - 2) % DR]), lv_of(s.PA[(r0 - 1) % DR]), lv_of(s.PA[r0 % DR]), s.CW[(r0 - 1) % 12]);
    // stage 1: odd pass of iteration 1 on row t-2, B[odd] <- view(A); in place in the ring
    {
        constexpr int PAR = (PAR0 + U + 10) & 1;       // parity of window row t-2
        R pb = lv_of(s.PB[(r0 - 2) % DR]);
        rbp_relax<PAR, 1>(k, lm, rbm_sel_fluid<N>(s.CW[(r0 - 2) % 12]), s.VA[(r0 - 3) % 6], s.VA[(r0 - 2) % 6], s.VA[(r0 - 1) % 6],
                          s.S2[(r0 - 2) % 6], s.S3[(r0 - 2) % 6], pb);
        s.PB[(r0 - 2) % DR] = f2_of(pb);
    }
    // stage 2: even pass of iteration 1 on row t-3, from the stage-1 state of rows t-4, t-3, t-2
    {
        constexpr int PAR = (PAR0 + U + 9) & 1;
        R &o = s.P2[(r0 - 3) % 3];
        const R bm = lv_of(s.PB[(r0 - 4) % DR]), bc = lv_of(s.PB[(r0 - 3) % DR]), bp = lv_of(s.PB[(r0 - 2) % DR]);
        o = bc;
        rbp_relax<PAR, 0>(k, lm, rbm_sel_fluid<N>(s.CW[(r0 - 3) % 12]), bm, bc, bp, s.S2[(r0 - 3) % 6], s.S3[(r0 - 3) % 6], o);
    }
    // view(B') of row t-4
    s.VB[(r0 - 4) % 3] = lv_bc_row<T, N>(lm, s.P2[(r0 - 5) % 3], s.P2[(r0 - 4) % 3], s.P2[(r0 - 3) % 3], s.CW[(r0 - 4) % 12]);
    // stage 3: odd pass of iteration 2 on row t-5, A[odd] <- view(B'); the other cells of the row stay view(A)
    {
        constexpr int PAR = (PAR0 + U + 7) & 1;
        R &o = s.P3[(r0 - 5) % 3];
        o = s.VA[(r0 - 5) % 6];
        rbp_relax<PAR, 1>(k, lm, rbm_sel_fluid<N>(s.CW[(r0 - 5) % 12]), s.VB[(r0 - 6) % 3], s.VB[(r0 - 5) % 3], s.VB[(r0 - 4) % 3],
                          s.S2[(r0 - 5) % 6], s.S3[(r0 - 5) % 6], o);
    }
    // stage 4: even pass of iteration 2 on row t-6, and the stores: C <- A after iteration 2, D <- view(B')
    {
        constexpr int PAR = (PAR0 + U + 6) & 1;
        const uint32_t cw = s.CW[(r0 - 6) % 12];
        const unsigned fl = rbm_sel_fluid<N>(cw);
        R o = s.P3[(r0 - 6) % 3];
        rbp_relax<PAR, 0>(k, lm, fl, s.P3[(r0 - 7) % 3], s.P3[(r0 - 6) % 3], s.P3[(r0 - 5) % 3], s.S2[(r0 - 6) % 6], s.S3[(r0 - 6) % 6], o);
        const int w = t - 6, j = j0 - 4 + w;
        if (w >= 4 && w <= W - 5 && j < je) {
            const unsigned sel = fl | lv_sel_target<N>(cw);
            T *pc = (T *)a.C + (size_t)j * g.P + lm.i0, *pd = (T *)a.D + (size_t)j * g.P + lm.i0;
            if (__all(sel == ALL)) {       // the common row: every cell of the wave is stored - one vector store per lane and plane
                if (lm.owner) {
                    lv_store_sel<T, N>(pc, o, ALL);
                    lv_store_sel<T, N>(pd, s.VB[(r0 - 6) % 3], ALL);
                }
            } else if (lm.owner && sel) {
                lv_store_sel<T, N>(pc, o, sel);
                lv_store_sel<T, N>(pd, s.VB[(r0 - 6) % 3], sel);
            }
        }
    }
}

template <int U, int PF, int PAR0, int DM>
struct RbmUnroll {
    static __device__ __forceinline__ void run(RbmState<PF> &s, const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, unsigned off4, unsigned off1, int j0, int je, int t0, int W, const RbmArgs &a)
    {
        rbm_step<U, PF, PAR0, DM>(s, g, k, lm, off4, off1, j0, je, t0 + U, W, a);
        if constexpr (U + 1 < 12) RbmUnroll<U + 1, PF, PAR0, DM>::run(s, g, k, lm, off4, off1, j0, je, t0, W, a);
    }
};

// grid: workgroups of 4 waves = 4 neighbouring wave columns (60 owner lanes of 2 cells) of ONE strip of L rows; dense XCD-band launch or
// compact list (band_coords).  PAR0: parity of (g.ybase + jb - 4), a launch constant because L is even.  f32, lanes of 2 cells.
template <int N, int PF, int PAR0, int DM, typename T>
__global__ __launch_bounds__(256) void k_rbsor_march(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, int L, RbmArgs a)
{
    static_assert(N == 2 && sizeof(T) == 4, "the marching passes are built for f32 on lanes of 2 cells");
    int wx, ty;
    if (!tile_coords_n<N>(g, nbx, nby, jb, je, L, wx, ty)) return;
    const LaneMapN<N> lm = lane_map_n<N>(g, wx);
    const int j0 = jb + ty * L, W = L + 8;
    const unsigned off4 = (unsigned)lm.i0 * 4u, off1 = (unsigned)lm.i0;
    RbmState<PF> s;
    constexpr int DR = 3 + PF;
#pragma unroll
    for (int r = 0; r < DR; ++r) { s.PA[r] = fs_f2{0.f, 0.f}; s.VX[r] = fs_f2{0.f, 0.f}; s.VY[r] = fs_f2{0.f, 0.f}; s.PB[r] = fs_f2{0.f, 0.f}; }
#pragma unroll
    for (int r = 0; r < 6; ++r) { s.VA[r] = lv_zero<T, N>(); s.S2[r] = lv_zero<T, N>(); s.S3[r] = lv_zero<T, N>(); }
#pragma unroll
    for (int r = 0; r < 3; ++r) { s.P2[r] = lv_zero<T, N>(); s.VB[r] = lv_zero<T, N>(); s.P3[r] = lv_zero<T, N>(); }
#pragma unroll
    for (int r = 0; r < 12; ++r) s.CW[r] = 0x80808080u;          // "not fluid, no recipe": rows that were never loaded relax nothing
    // what steps -PF .. -1 would have requested (5 loads each: the counter arithmetic of rbm_step holds from the first step on)
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const int j = clampy(g, j0 - 4 + r), jb2 = clampy(g, j0 - 4 + (r >= 2 ? r - 2 : 0));
        mload2(s.PA[r % DR], row_ptr(a.A, g, 1, 0, j), off4);
        mload2(s.VX[r % DR], row_ptr(a.v, g, 2, 0, j), off4);
        mload2(s.VY[r % DR], row_ptr(a.v, g, 2, 1, j), off4);
        mload_u16(s.CW[r % 12], a.rbcode + (size_t)j * g.Pm, off1);
        mload2(s.PB[(r - 2 + 12) % DR], row_ptr(a.B, g, 1, 0, jb2), off4);
    }
    // the last stored row is window row W - 5 = step W + 1: L + 10 steps, a whole number of 12-step bodies when L = 12 m - 10
    for (int t0 = 0; t0 < W + 2; t0 += 12) {
        RbmUnroll<0, PF, PAR0, DM>::run(s, g, k, lm, off4, off1, j0, je, t0, W, a);
        mwait<0>();          // the back edge: whatever copies the register allocator places there must see landed data
    }
}

}  // namespace fs
