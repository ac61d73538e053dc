// fs_jmarch.h - S lazily-bounded Jacobi sweeps per pass as a ROW-MARCHING software pipeline (S = 4, 6, 8).
//
// Reference (fs/pressure_updater.py:56-66): n x { K7(p.current); p.next[not wall] = predict_p(p.current); swap }, K7 =
// fs/boundary_condition.py:41-65.  fs_jquad.h runs four of these rounds on a register tile of 4 rows: a window of 12 rows, every sweep
// evaluated on 2.5 x the rows it is needed for, 2500 VALU instructions per tile - long Jacobi runs (BASELINE configs[1]: 50 sweeps per
// step on a cache-resident 3200 x 1600 grid) are bound by instruction issue and latency, not by memory (0.29 of the roofline).
//
// Here a wave walks down a strip of L rows (fs_rbmarch.h): step t loads window row t + PF of the iterate, the code plane and (one row
// behind) the source pair, and sweep s advances to window row t - s, reading rows t-s-1 .. t-s+1 of sweep s-1 from a ring of three rows.
// A strip requests L + 2 S rows for L rows of output and runs every sweep (L + 2 S) / L times per row; 8 sweeps cost 8 three-row
// rings, not a 20-row window.  K7 is evaluated where it is consumed: sweep s at row w needs the boundary VIEW of rows w-1, w, w+1 of
// the previous iterate (lv_bc_row, fs_rbpair.h) - a recipe never reads a source on the far side of its target as seen from a live
// reader (host-checked: fs_jacobi_quad_ok, the condition of fs_jquad.h), so the view of row w+1 AS SEEN FROM ROW w needs rows w and
// w+1 only.  Rows without a wall or a target - a wave-uniform test of the code words, kept as bit rings in scalar registers - take a
// path without views and selects.  Cells the sweep does not compute (walls) carry their raw value from sweep to sweep: a wall cell with a
// recipe is recomputed by whoever reads it, one without holds the same value in both buffers (Field.static_id, checked by the host).
// `jcode`: ONE byte per cell, bits 0-6 the pressure recipe (fs_march.h lazy_value), bit 7 "wall" (mask == 1: not computed).
// Same arithmetic and operation order as fs_jquad.h jq_row / predict_from: bit-identical results.
#pragma once
#include "fs_rbmarch.h"

namespace fs {

template <int N> __device__ __forceinline__ unsigned jm_sel_computed(uint32_t cw) { return rbm_sel_fluid<N>(cw); }   // bit 7 clear

// one sweep of one row from finished rows (fs_jquad.h jq_row; cells that are not computed keep `carry`)
template <bool SELECT, typename T, int N>
__device__ __forceinline__ LV<T, N> jm_row(const LaneMapN<N> &lm, unsigned computed, const LV<T, N> &m, const LV<T, N> &ctr, const LV<T, N> &p,
                                           const LV<T, N> &s2, const LV<T, N> &s3, const LV<T, N> &carry)
{
    const T pl = lv_left<T, N>(lm, ctr), pr = lv_right<T, N>(lm, ctr);
    LV<T, N> o;
#pragma unroll
    for (int c = 0; c < N; ++c) {
        const T pE = c == N - 1 ? pr : ctr.a[c == N - 1 ? c : c + 1], pW = c == 0 ? pl : ctr.a[c == 0 ? 0 : c - 1];
        const T val = predict_from(pE, pW, p.a[c], m.a[c], s2.a[c], s3.a[c]);
        o.a[c] = !SELECT || (computed & (1u << c)) ? val : carry.a[c];
    }
    return o;
}

// The boundary views of the three rows a sweep reads, OUT OF LINE: inlined, the recipe decode of 3 x N cells made the 12-step loop body
// 88 KB at 4 sweeps per pass (170 KB at 8) - beyond the instruction cache, and the kernel ran at the speed of its instruction fetch
// (3.7 us per step and wave, whatever the prefetch distance).  Rows with a target nearby are rare; they pay a call.
template <typename T, int N>
__device__ __attribute__((noinline)) void jm_views(int edge, LV<T, N> Am, LV<T, N> Ac, LV<T, N> Ap, uint32_t cm, uint32_t cc, uint32_t cp,
                                                   LV<T, N> &Vm, LV<T, N> &Vc, LV<T, N> &Vp)
{
    LaneMapN<N> lm;
    lm.i0 = 0; lm.owner = true; lm.at_lo = (edge & 1) != 0; lm.at_hi = (edge & 2) != 0;
    Vm = lv_bc_row<T, N>(lm, Am, Am, Ac, cm);
    Vc = lv_bc_row<T, N>(lm, Am, Ac, Ap, cc);
    Vp = lv_bc_row<T, N>(lm, Ac, Ap, Ap, cp);
}

template <int S, int PF>
struct JmState {
    static constexpr int DR = 3 + PF;
    using R = LV<float, 2>;
    fs_f2 P0[DR];            // raw iterate, rows t-2 .. t+PF (requested PF steps ahead)
    R P[S][3];               // P[s-1]: after sweep s, rows t-s-2 .. t-s   (the last one is only a temporary)
    fs_f2 S2[12], S3[12];    // source pair, rows t-S .. t-1+PF
    uint32_t CW[12];         // code words, rows t-S-1 .. t+PF
};

struct JmArgs {
    const uint8_t *jcode;
    void *pn;
    const void *pc, *src;
};

// U = t mod 12 (compile time), t = step (wave-uniform).  Ring loads per step: 4 (fs_rbmarch.h mload2 / mwait).
template <int U, int S, int PF>
__device__ __forceinline__ void jm_step(JmState<S, PF> &st, uint32_t &tf, uint32_t &nf, const Grid &g, const LaneMapN<2> &lm, unsigned off4, unsigned off1, int j0, int je, int t, int W,
                                        const JmArgs &a)
{
    constexpr int DR = 3 + PF, N = 2;
    constexpr unsigned ALL = 3u;
    using T = float;
    using R = LV<T, N>;
    {   // request window row t + PF of the iterate and the code plane, row t + PF - 1 of the source pair (past the window: its last row again)
        const int wr = t + PF < W ? t + PF : W - 1, ws = t + PF - 1 < W ? t + PF - 1 : W - 1;
        const int j = clampy(g, j0 - S + wr), js = clampy(g, j0 - S + ws);
        mload2(st.P0[(U + PF) % DR], row_ptr(a.pc, g, 1, 0, j), off4);
        mload_u16(st.CW[(U + PF) % 12], a.jcode + (size_t)j * g.Pm, off1);
        mload2(st.S2[(U + PF + 11) % 12], row_ptr(a.src, g, 2, 0, js), off4);
        mload2(st.S3[(U + PF + 11) % 12], row_ptr(a.src, g, 2, 1, js), off4);
    }
    constexpr int r0 = U + 24;
    // rows t (iterate, code) and t - 1 (source) were requested PF steps ago: 4 PF loads have been issued since
    mwait<4 * PF>();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(st.P0[r0 % DR]), "+v"(st.CW[r0 % 12]), "+v"(st.S2[(r0 - 1) % 12]), "+v"(st.S3[(r0 - 1) % 12]));
#endif
    {   // row t has arrived: does the wave's row hold a recipe target / anything but plain fluid?  (bit k of the rings <-> row t - k)
        const uint32_t cw = st.CW[r0 % 12];
        tf = (tf << 1) | (__any((cw & 0x01010101u) != 0u) ? 1u : 0u);
        nf = (nf << 1) | (__any(cw != 0u) ? 1u : 0u);
    }
#pragma unroll
    for (int s = 1; s <= S; ++s) {
        // sweep s on window row w = t - s from rows w-1, w, w+1 of sweep s-1
        const R Am = s == 1 ? lv_of(st.P0[(r0 - 2) % DR]) : st.P[s == 1 ? 0 : s - 2][(r0 - s - 1) % 3];
        const R Ac = s == 1 ? lv_of(st.P0[(r0 - 1) % DR]) : st.P[s == 1 ? 0 : s - 2][(r0 - s) % 3];
        const R Ap = s == 1 ? lv_of(st.P0[r0 % DR]) : st.P[s == 1 ? 0 : s - 2][(r0 - s + 1) % 3];
        const R s2 = lv_of(st.S2[(r0 - s) % 12]), s3 = lv_of(st.S3[(r0 - s) % 12]);
        R &out = st.P[s - 1][(r0 - s) % 3];
        const bool views = ((tf >> (s - 1)) & 7u) != 0u, walls = ((nf >> s) & 1u) != 0u;
        if (!views && !walls) {
            out = jm_row<false, T, N>(lm, ALL, Am, Ac, Ap, s2, s3, Ac);
        } else {
            const uint32_t cm = st.CW[(r0 - s - 1) % 12], cc = st.CW[(r0 - s) % 12], cp = st.CW[(r0 - s + 1) % 12];
            R Vm, Vc, Vp;
            jm_views<T, N>((lm.at_lo ? 1 : 0) | (lm.at_hi ? 2 : 0), Am, Ac, Ap, cm, cc, cp, Vm, Vc, Vp);
            out = jm_row<true, T, N>(lm, jm_sel_computed<N>(cc), Vm, Vc, Vp, s2, s3, Ac);
        }
    }
    {
        const int w = t - S, j = j0 - S + w;
        if (w >= S && w <= W - 1 - S && j < je) {
            const unsigned sel = jm_sel_computed<N>(st.CW[(r0 - S) % 12]);
            const R &o = st.P[S - 1][(r0 - S) % 3];
            T *dst = (T *)a.pn + (size_t)j * g.P + lm.i0;
            if (__all(sel == ALL)) { if (lm.owner) lv_store_sel<T, N>(dst, o, ALL); }
            else if (lm.owner && sel) lv_store_sel<T, N>(dst, o, sel);
        }
    }
}

template <int U, int S, int PF>
struct JmUnroll {
    static __device__ __forceinline__ void run(JmState<S, PF> &st, uint32_t &tf, uint32_t &nf, const Grid &g, const LaneMapN<2> &lm, unsigned off4, unsigned off1, int j0, int je, int t0, int W, const JmArgs &a)
    {
        jm_step<U, S, PF>(st, tf, nf, g, lm, off4, off1, j0, je, t0 + U, W, a);
        if constexpr (U + 1 < 12) JmUnroll<U + 1, S, PF>::run(st, tf, nf, g, lm, off4, off1, j0, je, t0, W, a);
    }
};

// grid: workgroups of 4 waves = neighbouring wave columns of ONE strip of L rows (fs_rbmarch.h k_rbsor_march).  S sweeps reach S cells in
// x: S / 2 halo lanes per side (S = 4: 60 owner lanes = 120 cells per wave, 6: 58 = 116, 8: 56 = 112).  f32, lanes of 2 cells.
template <int N, int S, int PF, typename T>
__global__ __launch_bounds__(256) void k_jacobi_march(Grid g, int nbx, int nby, int jb, int je, int L, JmArgs a)
{
    static_assert(N == 2 && sizeof(T) == 4, "the marching passes are built for f32 on lanes of 2 cells");
    static_assert(S + PF <= 12 && S + 2 + PF <= 12, "ring depths");
    static_assert(S % N == 0, "whole halo lanes");
    constexpr int HL = S / N, OW = 64 - 2 * HL;
    int wx, ty;
    {
        int bx, by, cg;
        if (!band_coords<1>(g, nbx, nby, bx, by, cg)) return;
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
        wx = bx * nw + w;
        ty = by;
        if (!(wx * OW < g.X / N && jb + ty * L < je)) return;
    }
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int j0 = jb + ty * L, W = L + 2 * S;           // window row w <-> local row j0 - S + w
    const unsigned off4 = (unsigned)lm.i0 * 4u, off1 = (unsigned)lm.i0;
    JmState<S, PF> st;
    constexpr int DR = 3 + PF;
#pragma unroll
    for (int r = 0; r < DR; ++r) st.P0[r] = fs_f2{0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int r = 0; r < 3; ++r) st.P[s][r] = lv_zero<T, N>();
#pragma unroll
    for (int r = 0; r < 12; ++r) { st.S2[r] = fs_f2{0.f, 0.f}; st.S3[r] = fs_f2{0.f, 0.f}; st.CW[r] = 0x80808080u; }
    uint32_t tf = 0u, nf = ~0u;
    // what steps -PF .. -1 would have requested: 4 loads each, so that the counter arithmetic of jm_step holds from the first step on
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const int j = clampy(g, j0 - S + r), js = clampy(g, j0 - S + (r > 0 ? r - 1 : 0));
        mload2(st.P0[r % DR], row_ptr(a.pc, g, 1, 0, j), off4);
        mload_u16(st.CW[r % 12], a.jcode + (size_t)j * g.Pm, off1);
        mload2(st.S2[(r + 11) % 12], row_ptr(a.src, g, 2, 0, js), off4);
        mload2(st.S3[(r + 11) % 12], row_ptr(a.src, g, 2, 1, js), off4);
    }
    for (int t0 = 0; t0 < W; t0 += 12) {
        JmUnroll<0, S, PF>::run(st, tf, nf, g, lm, off4, off1, j0, je, t0, W, a);
        mwait<0>();          // the back edge: whatever copies the register allocator places there must see landed data
    }
}

}  // namespace fs
