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

// Row loads of the marching loop.  fs_march.h load_row_quad gets the `saddr` form (scalar row base + one 32-bit lane offset) because base and
// offset meet in one basic block; here the lane offset is loop-invariant, gets hoisted - zero-extended to 64 bits - out of the loop, and
// instruction selection, which works block by block, no longer sees a 32-bit offset: every load then costs a 64-bit VALU add and an address
// register pair.  The empty asm makes the offset opaque at each use, so the extension stays next to the load.
template <typename Q, typename T>
__device__ __forceinline__ Q rbm_load_row(const T *row, unsigned byte_off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(byte_off));
    return *reinterpret_cast<__attribute__((address_space(1))) const Q *>((fs_gcptr)uniform64((uint64_t)row) + byte_off);
#else
    return *reinterpret_cast<const Q *>((const char *)row + byte_off);
#endif
}
template <int C, typename T, int N>
__device__ __forceinline__ LV<T, N> rbm_field(const T *f, const Grid &g, int c, int i0, int j)
{
    const typename LVec<T, N>::type q = rbm_load_row<typename LVec<T, N>::type, T>(f + ((size_t)j * C + c) * g.P, (unsigned)i0 * (unsigned)sizeof(T));
    LV<T, N> r;
    if constexpr (N == 4) { r.a[0] = q.x; r.a[1] = q.y; r.a[2] = q.z; r.a[3] = q.w; }
    else { r.a[0] = q.x; r.a[1] = q.y; }
    return r;
}
template <int N>
__device__ __forceinline__ uint32_t rbm_bytes(const uint8_t *plane, const Grid &g, int i0, int j)
{ return (uint32_t)rbm_load_row<typename LMaskWord<N>::type, uint8_t>(plane + (size_t)j * g.Pm, (unsigned)i0); }

template <typename T, int N, int PF>
struct RbmState {
    static constexpr int DR = 3 + PF;      // raw rows t-2 .. t+PF
    using R = LV<T, N>;
    R PA[DR], VX[DR], VY[DR];
    R PB[DR];                              // rows t-4 .. t-2+PF: raw until stage 1 has run on the row, then the state after iteration 1's odd pass
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

// U = t mod 12 (compile time), t = step (wave-uniform)
template <int U, int N, int PF, int PAR0, int DM, typename T>
__device__ __forceinline__ void rbm_step(RbmState<T, N, PF> &s, const Grid &g, const Konst<T> &k, const LaneMapN<N> &lm, int i0, int j0, int je, int t, int W,
                                         const RbmArgs &a)
{
    constexpr int DR = 3 + PF;
    constexpr unsigned ALL = (1u << N) - 1u;
    using R = LV<T, N>;
    {   // prefetch: window row t + PF of A, v and the code plane, row t + PF - 2 of B
        const int j = clampy(g, j0 - 4 + t + PF), jb = clampy(g, j0 - 4 + t + PF - 2);
        constexpr int sl = (U + PF) % DR;
        if (t + PF < W) {                  // (rows past the window: nothing valid reads them)
            s.PA[sl] = rbm_field<1, T, N>((const T *)a.A, g, 0, i0, j);
            s.VX[sl] = rbm_field<2, T, N>((const T *)a.v, g, 0, i0, j);
            s.VY[sl] = rbm_field<2, T, N>((const T *)a.v, g, 1, i0, j);
            s.CW[(U + PF) % 12] = rbm_bytes<N>(a.rbcode, g, i0, j);
        }
        if (t + PF - 2 < W - 1) s.PB[(U + PF - 2 + 12) % DR] = rbm_field<1, T, N>((const T *)a.B, g, 0, i0, jb);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int r0 = U + 12;             // (U - k + 12) % depth: non-negative operands
    {   // Poisson source of row t-1 from v rows t-2, t-1, t (fs/pressure_updater.py:25-38), once per cell for all four half sweeps
        const R &xm = s.VX[(r0 - 2) % DR], &xc = s.VX[(r0 - 1) % DR], &xp = s.VX[r0 % DR];
        const R &ym = s.VY[(r0 - 2) % DR], &yc = s.VY[(r0 - 1) % DR], &yp = s.VY[r0 % DR];
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
    s.VA[(r0 - 1) % 6] = lv_bc_row<T, N>(lm, s.PA[(r0 - 2) % DR], s.PA[(r0 - 1) % DR], s.PA[r0 % DR], s.CW[(r0 - 1) % 12]);
    // stage 1: odd pass of iteration 1 on row t-2, B[odd] <- view(A); in place in the ring
    {
        constexpr int PAR = (PAR0 + U + 10) & 1;       // parity of window row t-2
        rbp_relax<PAR, 1>(k, lm, rbm_sel_fluid<N>(s.CW[(r0 - 2) % 12]), s.VA[(r0 - 3) % 6], s.VA[(r0 - 2) % 6], s.VA[(r0 - 1) % 6],
                          s.S2[(r0 - 2) % 6], s.S3[(r0 - 2) % 6], s.PB[(r0 - 2) % DR]);
    }
    // stage 2: even pass of iteration 1 on row t-3, from the stage-1 state of rows t-4, t-3, t-2
    {
        constexpr int PAR = (PAR0 + U + 9) & 1;
        R &o = s.P2[(r0 - 3) % 3];
        o = s.PB[(r0 - 3) % DR];
        rbp_relax<PAR, 0>(k, lm, rbm_sel_fluid<N>(s.CW[(r0 - 3) % 12]), s.PB[(r0 - 4) % DR], s.PB[(r0 - 3) % DR], s.PB[(r0 - 2) % DR],
                          s.S2[(r0 - 3) % 6], s.S3[(r0 - 3) % 6], o);
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
            T *pc = (T *)a.C + idx<1, T>(g, 0, i0, j), *pd = (T *)a.D + idx<1, T>(g, 0, i0, j);
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

template <int U, int N, int PF, int PAR0, int DM, typename T>
struct RbmUnroll {
    static __device__ __forceinline__ void run(RbmState<T, N, PF> &s, const Grid &g, const Konst<T> &k, const LaneMapN<N> &lm, int i0, int j0, int je, int t0, int W, const RbmArgs &a)
    {
        rbm_step<U, N, PF, PAR0, DM, T>(s, g, k, lm, i0, j0, je, t0 + U, W, a);
        if constexpr (U + 1 < 12) RbmUnroll<U + 1, N, PF, PAR0, DM, T>::run(s, g, k, lm, i0, j0, je, t0, W, a);
    }
};

// grid: workgroups of 4 waves = 4 neighbouring wave columns (60 owner lanes of N cells) of ONE strip of L rows; dense XCD-band launch or
// compact list (band_coords).  PAR0: parity of (g.ybase + jb - 4), a launch constant because L is even.
template <int N, int PF, int PAR0, int DM, typename T>
__global__ __launch_bounds__(256) void k_rbsor_march(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, int L, RbmArgs a)
{
    int wx, ty;
    if (!tile_coords_n<N>(g, nbx, nby, jb, je, L, wx, ty)) return;
    const LaneMapN<N> lm = lane_map_n<N>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * L, W = L + 8;
    RbmState<T, N, PF> s;
    constexpr int DR = 3 + PF;
#pragma unroll
    for (int r = 0; r < DR; ++r) { s.PA[r] = lv_zero<T, N>(); s.VX[r] = lv_zero<T, N>(); s.VY[r] = lv_zero<T, N>(); s.PB[r] = lv_zero<T, N>(); }
#pragma unroll
    for (int r = 0; r < 6; ++r) { s.VA[r] = lv_zero<T, N>(); s.S2[r] = lv_zero<T, N>(); s.S3[r] = lv_zero<T, N>(); }
#pragma unroll
    for (int r = 0; r < 3; ++r) { s.P2[r] = lv_zero<T, N>(); s.VB[r] = lv_zero<T, N>(); s.P3[r] = lv_zero<T, N>(); }
#pragma unroll
    for (int r = 0; r < 12; ++r) s.CW[r] = 0x80808080u;          // "not fluid, no recipe": rows that were never loaded relax nothing
    // rows 0 .. PF-1 of A, v and the code plane, rows 0 .. PF-3 of B: what steps -PF .. -1 would have requested
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const int j = clampy(g, j0 - 4 + r);
        s.PA[r % DR] = rbm_field<1, T, N>((const T *)a.A, g, 0, i0, j);
        s.VX[r % DR] = rbm_field<2, T, N>((const T *)a.v, g, 0, i0, j);
        s.VY[r % DR] = rbm_field<2, T, N>((const T *)a.v, g, 1, i0, j);
        s.CW[r % 12] = rbm_bytes<N>(a.rbcode, g, i0, j);
        if (r < PF - 2) s.PB[r % DR] = rbm_field<1, T, N>((const T *)a.B, g, 0, i0, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the last stored row is window row W - 5 = step W + 1: L + 10 steps, a whole number of 12-step bodies when L = 12 m - 10
    for (int t0 = 0; t0 < W + 2; t0 += 12)
        RbmUnroll<0, N, PF, PAR0, DM, T>::run(s, g, k, lm, i0, j0, je, t0, W, a);
}

}  // namespace fs
