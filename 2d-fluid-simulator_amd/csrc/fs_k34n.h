// fs_k34n.h - K3 + K4 of the CIP solvers in ONE pass, one body for the velocity (C = 2) and the dye (C = 3), on lanes of N cells.
//
// Reference: fs/solver.py:242-261 (_non_advection_phase_grad) and :267-332 (_advection_phase / _cip_advect); the dye :378-401.
//
// In the reference K3 writes the intermediate gradients into vx.next / vy.next, the buffers swap, and K4 reads them back on a 3x3
// neighbourhood: 16 B/cell written and 16 B/cell read again, plus the intermediate field read twice.  Here a lane evaluates K3 for rows
// j0-1 .. j0+RT of its cells in registers (halo recompute) and feeds K4 for rows j0 .. j0+RT-1 directly.  The intermediate gradients are never
// observable (the buffer that would hold them is overwritten by the next step's K3 before anything reads it), so they are not stored.
//
// Buffer choreography (SURVEY.md H5 - contents, not addresses, are what later kernels see): K4's output goes to a THIRD buffer `out`
// (the unfused code overwrites the pre-K2 buffer `fc` in place, which a fused kernel still has to read at radius 2 in other tiles),
// carrying fc's values on non-fluid cells exactly as the in-place update would leave them.  The new gradients go to the buffers the
// reference would use for the intermediates (gxo / gyo), carrying gxc / gyc on inflow / outflow cells; wall cells of the gradient
// buffers are never written by any kernel.  The caller rotates (f.cur, f.next, spare) and swaps the gradient buffers once.
//   fn = field after K2 (f.next), fc = field before K2 (f.cur), gxc / gyc = gradients before K3;
//   C = 2: the field advects itself (the other component is read from fn); C = 3: advected by `v`, the final velocity of the flow step.
// Every row is loaded CLAMPED (sample() clamps coordinates, fs/differentiation.py:4-9) and a register slot that stands for a row outside
// the domain takes the K3 result of the edge row it clamps onto, so one launch covers every row.  One channel per wave: the C passes over
// a tile are adjacent in dispatch order on the SAME XCD and share their input rows through that XCD's L2.
// PLAIN: the host listed this workgroup as seeing nothing but fluid within its reach (fs_api.hip tile_list, compact launch): no mask
// loads, constant selectors, unconditional stores - as its own kernel.  CLAMP: clamp_field(dye, 0, 1) (fs/solver.py:46-49) folded into
// the store of the advected value.
//
// Lane width: the round-2 form held 4 cells per lane and RT = 2 rows per tile (N = 4 here, FS_K34_N=4): of the 12
// 16-byte rows a lane requests per output row 7 are halo rows of the tile (re-read by the tiles above / below), and the gradient update
// runs on RT + 2 = 2 x the rows it is needed for.  With 2 cells per lane the same registers hold a tile of RT = 4 rows: 8.5 8-byte rows
// per output row, the gradient update on 1.5 x the rows, and a stacked workgroup covers 16 rows instead of 8 (halo rows shared with
// OTHER workgroups, i.e. through the L2 at best: 4 of 20 instead of 4 of 12).
// Lane map of fs_rbpair.h with ONE halo lane per side (the pass reaches 2 cells in x): 62 owner lanes = 124 cells per wave at N = 2.
#pragma once
#include "fs_rbpair.h"

namespace fs {

template <int N> __device__ __forceinline__ unsigned lv_sel_nw(uint32_t m)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((m >> (8 * c)) & 0xffu) != 1u ? (1u << c) : 0u;
    return s;
}
template <int N> __device__ __forceinline__ unsigned lv_sel_bit7(uint32_t code)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((code >> (8 * c + 7)) & 1u) << c;
    return s;
}
template <typename T, int N> __device__ __forceinline__ bool lv_hot1(const LV<T, N> &v)
{
    bool h = false;
#pragma unroll
    for (int c = 0; c < N; ++c) h = h || hot1(v.a[c]);
    return h;
}
template <typename T, int N> __device__ __forceinline__ void lv_store(T *dst, const LV<T, N> &v) { lv_store_sel<T, N>(dst, v, (1u << N) - 1u); }

// (tile_coords_nz: fs_rbpair.h)

template <int C, int c, int N, int RT, int DM, bool PLAIN, bool CLAMP, typename T, int HL = 1>
__device__ __forceinline__ void cip_grad_advect_n_body(const Grid &g, const Konst<T> &k, int nbx, int nby, int jb, int je,
                                                       T *out, T *gxo, T *gyo, const T *fn, const T *fc,
                                                       const T *gxc, const T *gyc, const T *v, unsigned *hot, const uint8_t *bcmap, int full)
{
    using R = LV<T, N>;
    constexpr unsigned ALL = (1u << N) - 1u;
    // HL: halo lanes per side.  The pass reaches 2 cells in x: ONE suffices (62 owner lanes); 2 (60 owners) where the launch shares its wave
    // columns with the pass that evaluates K2 on the way (fs_k234.h: reach 3 cells)
    constexpr bool SELF = C == 2;                // the field advects itself
    int wx, ty, cg;
    if (!tile_coords_nz<N, C, HL>(g, nbx, nby, jb, je, RT, wx, ty, cg)) return;
    const LaneMapN<N> lm_in = lane_map_n<N, HL>(g, wx);
    // (PLAIN: no lane at the domain's first / last column - the clamp of the x-neighbours folds away, fs_rbpair.h rbsor_pair_tile)
    const LaneMapN<N> lm = PLAIN ? LaneMapN<N>{lm_in.i0, lm_in.owner, false, false} : lm_in;
    const int i0 = lm.i0, j0 = jb + ty * RT;

    unsigned nw[RT + 2], fl[RT];                 // not-wall selectors of rows j0-1 .. j0+RT, fluid selectors of rows j0 .. j0+RT-1
    bool any_fl = PLAIN;
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        if (PLAIN) { nw[s] = ALL; if (s >= 1 && s <= RT) fl[s - 1] = j0 + s - 1 < je ? ALL : 0u; continue; }
        const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 1 + s));
        nw[s] = lv_sel_nw<N>(m);
        if (s >= 1 && s <= RT) { fl[s - 1] = j0 + s - 1 < je ? lv_sel_fluid<N>(m) : 0u; any_fl |= fl[s - 1] != 0u; }
    }
    if (!PLAIN && !__any(any_fl)) {
        // no fluid cell in this wave's tile: every output is a carried value (out = fc, old gradients on inflow / outflow cells) - and only
        // cells that SOME kernel writes can differ between fc and out: not-wall cells and, for the velocity, the targets of the velocity
        // boundary kernel (bit 7 of the recipe byte, fs_api.hip build_bc_ops).  Deep wall rows move nothing (a third of scene 5); `full`:
        // after an upload the two buffers may differ anywhere - carry every cell once (fs/solver.py, Field.static_id).
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int j = j0 + t;
            if (j >= je) break;
            unsigned touch = full ? ALL : nw[t + 1];
            if (SELF && !full) touch |= lv_sel_bit7<N>(lv_bytes<N>(bcmap, g, i0, clampy(g, j)));
            if (!__any(lm.owner && touch != 0u)) continue;
            const R f = lv_field<C, T, N>(fc, g, c, i0, j);
            if (lm.owner && touch) {
                if (SELF) raise_hot(hot, lv_hot1<T, N>(f));
                lv_store<T, N>(out + idx<C, T>(g, c, i0, j), f);
                if (nw[t + 1]) {
                    lv_store_sel<T, N>(gxo + idx<C, T>(g, c, i0, j), lv_field<C, T, N>(gxc, g, c, i0, j), nw[t + 1]);
                    lv_store_sel<T, N>(gyo + idx<C, T>(g, c, i0, j), lv_field<C, T, N>(gyc, g, c, i0, j), nw[t + 1]);
                }
            }
        }
        return;
    }

    // slot u of Nn / Fc <-> row j0 - 2 + u;   slot s of GX / GY / NX / NY and of the advecting rows <-> row j0 - 1 + s
    // advecting velocity: C = 2: the component at hand is Nn itself, the other one AX (c == 1) / AY (c == 0); C = 3: AX, AY from v
    R Nn[RT + 4], Fc[RT + 4], GX[RT + 2], GY[RT + 2], AX[RT + 2], AY[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        Nn[u] = lv_field<C, T, N>(fn, g, c, i0, clampy(g, j0 - 2 + u));
        Fc[u] = lv_field<C, T, N>(fc, g, c, i0, clampy(g, j0 - 2 + u));
    }
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        GX[s] = lv_field<C, T, N>(gxc, g, c, i0, row);
        GY[s] = lv_field<C, T, N>(gyc, g, c, i0, row);
        if (SELF) {
            if (c == 0) AY[s] = lv_field<2, T, N>(fn, g, 1, i0, row); else AX[s] = lv_field<2, T, N>(fn, g, 0, i0, row);
        } else {
            AX[s] = lv_field<2, T, N>(v, g, 0, i0, row);
            AY[s] = lv_field<2, T, N>(v, g, 1, i0, row);
        }
    }
    // ---- K3 on rows j0-1 .. j0+RT: wall cells keep the stored gradient ----
    R NX[RT + 2], NY[RT + 2];
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const R &n1 = Nn[s + 1], &c1 = Fc[s + 1];
        const T nl = lv_left<T, N>(lm, n1), nr = lv_right<T, N>(lm, n1);
        const T cl = lv_left<T, N>(lm, c1), cr = lv_right<T, N>(lm, c1);
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const T nE = q == N - 1 ? nr : n1.a[q == N - 1 ? q : q + 1], nW = q == 0 ? nl : n1.a[q == 0 ? 0 : q - 1];
            const T cE = q == N - 1 ? cr : c1.a[q == N - 1 ? q : q + 1], cW = q == 0 ? cl : c1.a[q == 0 ? 0 : q - 1];
            const T sx = ((nE - cE) - nW) + cW;
            const T sy = ((Nn[s + 2].a[q] - Fc[s + 2].a[q]) - Nn[s].a[q]) + Fc[s].a[q];
            const bool live = (nw[s] >> q) & 1u;
            NX[s].a[q] = live ? GX[s].a[q] + xdiv<DM>(sx, k.two_dx, k.inv_two_dx, k.r_two_dx) : GX[s].a[q];
            NY[s].a[q] = live ? GY[s].a[q] + xdiv<DM>(sy, k.two_dx, k.inv_two_dx, k.r_two_dx) : GY[s].a[q];
        }
    }
    // a slot that stands for a row outside the domain takes the K3 result of the edge row it clamps onto (wave-uniform)
    if (j0 - 1 < g.jlo) { NX[0] = NX[1]; NY[0] = NY[1]; }
#pragma unroll
    for (int s = 1; s < RT + 2; ++s)
        if (j0 - 1 + s > g.jhi) { NX[s] = NX[s - 1]; NY[s] = NY[s - 1]; }
    // ---- K4 on rows j0 .. j0+RT-1 ----
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const R &Nm = Nn[t + 1], &Nc = Nn[t + 2], &Np = Nn[t + 3];                                 // value field rows j-1, j, j+1
        const R &VXm = SELF && c == 0 ? Nm : AX[t], &VXr = SELF && c == 0 ? Nc : AX[t + 1], &VXp = SELF && c == 0 ? Np : AX[t + 2];   // advecting velocity
        const R &VYm = SELF && c == 1 ? Nm : AY[t], &VYr = SELF && c == 1 ? Nc : AY[t + 1], &VYp = SELF && c == 1 ? Np : AY[t + 2];
        const T vxl = lv_left<T, N>(lm, VXr), vxr = lv_right<T, N>(lm, VXr);
        const T vyl = lv_left<T, N>(lm, VYr), vyr = lv_right<T, N>(lm, VYr);
        const T fl0 = lv_left<T, N>(lm, Nm), fr0 = lv_right<T, N>(lm, Nm);
        const T fl1 = lv_left<T, N>(lm, Nc), fr1 = lv_right<T, N>(lm, Nc);
        const T fl2 = lv_left<T, N>(lm, Np), fr2 = lv_right<T, N>(lm, Np);
        const T fxl = lv_left<T, N>(lm, NX[t + 1]), fxr = lv_right<T, N>(lm, NX[t + 1]);
        const T fyl = lv_left<T, N>(lm, NY[t + 1]), fyr = lv_right<T, N>(lm, NY[t + 1]);
        R OV = Fc[t + 2], OX = GX[t + 1], OY = GY[t + 1];           // carry values; fluid cells are replaced below
#pragma unroll
        for (int q = 0; q < N; ++q) {
            constexpr int L = N - 1;
            const T vx = VXr.a[q], vy = VYr.a[q];
            const bool nx = vx < (T)0.0, ny = vy < (T)0.0;
            const T vxE = q == L ? vxr : VXr.a[q == L ? q : q + 1], vxW = q == 0 ? vxl : VXr.a[q == 0 ? 0 : q - 1];
            const T vyE = q == L ? vyr : VYr.a[q == L ? q : q + 1], vyW = q == 0 ? vyl : VYr.a[q == 0 ? 0 : q - 1];
            const T dxx = xdiv<DM>((T)0.5 * (vxE - vxW), k.dx, k.inv_dx, k.r_dx), dxy = xdiv<DM>((T)0.5 * (vyE - vyW), k.dx, k.inv_dx, k.r_dx);
            const T dyx = xdiv<DM>((T)0.5 * (VXp.a[q] - VXm.a[q]), k.dx, k.inv_dx, k.r_dx), dyy = xdiv<DM>((T)0.5 * (VYp.a[q] - VYm.a[q]), k.dx, k.inv_dx, k.r_dx);
            const T fE1 = q == L ? fr1 : Nc.a[q == L ? q : q + 1], fW1 = q == 0 ? fl1 : Nc.a[q == 0 ? 0 : q - 1];
            const T fE0 = q == L ? fr0 : Nm.a[q == L ? q : q + 1], fW0 = q == 0 ? fl0 : Nm.a[q == 0 ? 0 : q - 1];
            const T fE2 = q == L ? fr2 : Np.a[q == L ? q : q + 1], fW2 = q == 0 ? fl2 : Np.a[q == 0 ? 0 : q - 1];
            const T fxE = q == L ? fxr : NX[t + 1].a[q == L ? q : q + 1], fxW = q == 0 ? fxl : NX[t + 1].a[q == 0 ? 0 : q - 1];
            const T fyE = q == L ? fyr : NY[t + 1].a[q == L ? q : q + 1], fyW = q == 0 ? fyl : NY[t + 1].a[q == 0 ? 0 : q - 1];
            const T f00 = Nc.a[q];
            const T f0m = ny ? Np.a[q] : Nm.a[q];
            const T fm0 = nx ? fE1 : fW1;
            const T fmm = ny ? (nx ? fE2 : fW2) : (nx ? fE0 : fW0);
            const T fx00 = NX[t + 1].a[q], fxm0 = nx ? fxE : fxW, fx0m = ny ? NX[t + 2].a[q] : NX[t].a[q];
            const T fy00 = NY[t + 1].a[q], fy0m = ny ? NY[t + 2].a[q] : NY[t].a[q], fym0 = nx ? fyE : fyW;
            T of, ofx, ofy;
            cip_point<DM>(k, vx, vy, dxx, dxy, dyx, dyy, f00, f0m, fm0, fmm, fx00, fxm0, fx0m, fy00, fy0m, fym0, of, ofx, ofy);
            if (CLAMP) of = tmin(tmax(of, (T)0.0), (T)1.0);
            if (PLAIN || ((fl[t] >> q) & 1u)) { OV.a[q] = of; OX.a[q] = ofx; OY.a[q] = ofy; }      // (PLAIN: rows past je have left the loop above)
        }
        if (lm.owner) {
            if (SELF && lv_hot1<T, N>(OV)) {                                               // one component per pass: conservative - and rare
                // FLUID cells raise word [3] (cleared by the next kernel that rewrites every fluid cell of this buffer with both components in
                // sight: K2, K5+K6), carried non-fluid values the sticky word [0] (fs_device.h)
                bool hf = false, hn = false;
#pragma unroll
                for (int q = 0; q < N; ++q)
                    if (hot1(OV.a[q])) { if (PLAIN || ((fl[t] >> q) & 1u)) hf = true; else hn = true; }
                raise_hot(hot + 3, hf);
                raise_hot(hot, hn);
            }
            lv_store<T, N>(out + idx<C, T>(g, c, i0, j), OV);                              // every cell: result or carried value
            if (nw[t + 1]) {
                lv_store_sel<T, N>(gxo + idx<C, T>(g, c, i0, j), OX, nw[t + 1]);           // fluid: result, inflow/outflow: carried
                lv_store_sel<T, N>(gyo + idx<C, T>(g, c, i0, j), OY, nw[t + 1]);
            }
        }
    }
}

// ---- the same pass with the lane's two cells as ONE packed operand (f32, N = 2; fs_device.h v2f) ----------------------------------
// Every mul / add / sub of K3 and of the CIP polynomial is the same expression for both cells of the lane: on v2f they issue as
// v_pk_mul_f32 / v_pk_add_f32 (one slot for two IEEE operations); the upwind selects, the DPP shifts and the divisions that are not
// multiplications stay per half.  Same expression tree, same rounding per element: bit-identical to the scalar form above (which
// remains the form of N = 4 and of f64).
// (pk / unpk / east / west / sel2 / ew_diff: fs_rbpair.h, next to LV)
// K3 on rows j0-1 .. j0+RT and K4 on rows j0 .. j0+RT-1 of one wave's tile from its register window (packed): Nn / Fc rows j0-2 .. j0+RT+1 of
// the field after / before K2, GX / GY rows j0-1 .. j0+RT of the old gradients, AX / AY the advecting velocity on those rows (C = 2: only
// the component that is not the field's own is read).  Shared by the two-kernel form below and the pass that evaluates K2 on the way
// (fs_k234.h).
// selectors of a tile's window for the packed core: nw(s) = not-wall bits of row j0-1+s (s = 0 .. RT+1), fl(t) = fluid bits of row j0+t, both
// 2 bits per lane.  Packed: all rows in one word each (one register per row cost the general bodies 8 - 12 VGPRs: the dye's spilled 28 - 52 bytes);
// Plain: constants.
struct MaskPacked {
    unsigned nwbits, flbits;        // bits 2s, 2s+1: nw(s);  bits 2t, 2t+1: fl(t)
    __device__ __forceinline__ unsigned nw(int s) const { return (nwbits >> (2 * s)) & 3u; }
    __device__ __forceinline__ unsigned fl(int t) const { return (flbits >> (2 * t)) & 3u; }
};
struct MaskPlain {
    int rows;                       // own rows below je
    __device__ __forceinline__ unsigned nw(int) const { return 3u; }
    __device__ __forceinline__ unsigned fl(int t) const { return t < rows ? 3u : 0u; }
};
// the advecting velocity of the core, rows j0-1 .. j0+RT (slot s): register rows (the two-kernel form and the dye), or the sibling wave's rows in LDS,
// read where they are used (fs_k234.h: six rows held from the barrier on were 12 VGPRs at the register peak of the boundary body)
template <int RT> struct AdvRows {
    const v2f (&ax)[RT + 2]; const v2f (&ay)[RT + 2];
    __device__ __forceinline__ v2f x(int s) const { return ax[s]; }
    __device__ __forceinline__ v2f y(int s) const { return ay[s]; }
};
template <int RT> struct AdvLds {
    const v2f (*rows)[64]; int lane;       // rows[s][lane]: the sibling component's row s
    __device__ __forceinline__ v2f x(int s) const { return rows[s][lane]; }
    __device__ __forceinline__ v2f y(int s) const { return rows[s][lane]; }
};
template <int C, int c, int RT, int DM, bool PLAIN, bool CLAMP, typename MK, typename AV>
__device__ __forceinline__ void cip_k34_pk_core(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je, const MK &mk,
                                                const v2f (&Nn)[RT + 4], const v2f (&Fc)[RT + 4], const v2f (&GX)[RT + 2], const v2f (&GY)[RT + 2],
                                                const AV &adv, float *out, float *gxo, float *gyo, unsigned *hot,
                                                const float *gxc = nullptr, const float *gyc = nullptr)      // (not PLAIN: the old gradients in memory, for the cells that carry them)
{
    using T = float;
    constexpr int N = 2;
    using R = LV<T, N>;
    constexpr bool SELF = C == 2;
    // K3 on rows j0-1 .. j0+RT (slot s) and K4 on rows j0 .. j0+RT-1 (t), interleaved: K4 of row t follows K3 of row t+2, the last gradient
    // row it reads - a row of the old gradients and of fc is dead one step after its K3 (the register peak of this pass is K4's: ~30 live pairs).
    // D = fn - fc per row, once: it is the first difference of sy two rows below AND - shifted one cell - of sx one row below
    // ((nE - cE) of a lane's second cell is the next lane's first D: the same operation on the same operands)
    v2f NX[RT + 2], NY[RT + 2], D[RT + 4];
    D[0] = Nn[0] - Fc[0];
    D[1] = Nn[1] - Fc[1];
#pragma unroll
    for (int t = -2; t < RT; ++t) {
        const int s = t + 2;
        {
            D[s + 2] = Nn[s + 2] - Fc[s + 2];
            const v2f n1 = Nn[s + 1], c1 = Fc[s + 1], d1 = D[s + 1];
            const T nl = lv_left<T, N>(lm, unpk(n1)), cl = lv_left<T, N>(lm, unpk(c1)), dr = lv_right<T, N>(lm, unpk(d1));
            v2f sx;
            sx.x = (d1.y - nl) + cl;
            sx.y = (dr - n1.x) + c1.x;
            const v2f sy = (D[s + 2] - Nn[s]) + Fc[s];
            const v2f ux = GX[s] + xdiv<DM>(sx, k.two_dx, k.inv_two_dx, k.r_two_dx), uy = GY[s] + xdiv<DM>(sy, k.two_dx, k.inv_two_dx, k.r_two_dx);
            NX[s] = PLAIN ? ux : sel2(mk.nw(s), ux, GX[s]);
            NY[s] = PLAIN ? uy : sel2(mk.nw(s), uy, GY[s]);
            if (!PLAIN) {        // a slot that stands for a row outside the domain takes the K3 result of the edge row it clamps onto (wave-uniform;
                                 // a plain tile lies inside the fluid: no row of its window is outside the domain)
                if (s == 1 && j0 - 1 < g.jlo) { NX[0] = NX[1]; NY[0] = NY[1]; }
                if (s >= 1 && j0 - 1 + s > g.jhi) { NX[s] = NX[s - 1]; NY[s] = NY[s - 1]; }
            }
        }
        const int j = j0 + t;
        if (t < 0 || j >= je) continue;
        const v2f Nm = Nn[t + 1], Nc = Nn[t + 2], Np = Nn[t + 3];
        v2f VXm, VXr, VXp, VYm, VYr, VYp;
        if constexpr (SELF && c == 0) { VXm = Nm; VXr = Nc; VXp = Np; } else { VXm = adv.x(t); VXr = adv.x(t + 1); VXp = adv.x(t + 2); }
        if constexpr (SELF && c == 1) { VYm = Nm; VYr = Nc; VYp = Np; } else { VYm = adv.y(t); VYr = adv.y(t + 1); VYp = adv.y(t + 2); }
        const T vxl = lv_left<T, N>(lm, unpk(VXr)), vxr = lv_right<T, N>(lm, unpk(VXr));
        const T vyl = lv_left<T, N>(lm, unpk(VYr)), vyr = lv_right<T, N>(lm, unpk(VYr));
        const T fl0 = lv_left<T, N>(lm, unpk(Nm)), fr0 = lv_right<T, N>(lm, unpk(Nm));
        const T fl1 = lv_left<T, N>(lm, unpk(Nc)), fr1 = lv_right<T, N>(lm, unpk(Nc));
        const T fl2 = lv_left<T, N>(lm, unpk(Np)), fr2 = lv_right<T, N>(lm, unpk(Np));
        const T fxl = lv_left<T, N>(lm, unpk(NX[t + 1])), fxr = lv_right<T, N>(lm, unpk(NX[t + 1]));
        const T fyl = lv_left<T, N>(lm, unpk(NY[t + 1])), fyr = lv_right<T, N>(lm, unpk(NY[t + 1]));
        const v2f vx = VXr, vy = VYr;
        const v2f dxx = xdiv<DM>(0.5f * ew_diff(VXr, vxl, vxr), k.dx, k.inv_dx, k.r_dx), dxy = xdiv<DM>(0.5f * ew_diff(VYr, vyl, vyr), k.dx, k.inv_dx, k.r_dx);
        const v2f dyx = xdiv<DM>(0.5f * (VXp - VXm), k.dx, k.inv_dx, k.r_dx), dyy = xdiv<DM>(0.5f * (VYp - VYm), k.dx, k.inv_dx, k.r_dx);
        const v2f f00 = Nc;
        const v2f f0m = sel_neg(vy, Np, Nm);
        const v2f fm0 = sel_neg(vx, east(Nc, fr1), west(fl1, Nc));
        const v2f fmm = sel_neg(vy, sel_neg(vx, east(Np, fr2), west(fl2, Np)), sel_neg(vx, east(Nm, fr0), west(fl0, Nm)));
        const v2f fx00 = NX[t + 1], fxm0 = sel_neg(vx, east(NX[t + 1], fxr), west(fxl, NX[t + 1])), fx0m = sel_neg(vy, NX[t + 2], NX[t]);
        const v2f fy00 = NY[t + 1], fy0m = sel_neg(vy, NY[t + 2], NY[t]), fym0 = sel_neg(vx, east(NY[t + 1], fyr), west(fyl, NY[t + 1]));
        v2f of, ofx, ofy;
        cip_point<DM>(k, vx, vy, dxx, dxy, dyx, dyy, f00, f0m, fm0, fmm, fx00, fxm0, fx0m, fy00, fy0m, fym0, of, ofx, ofy);
        if (CLAMP) { of.x = tmin(tmax(of.x, 0.0f), 1.0f); of.y = tmin(tmax(of.y, 0.0f), 1.0f); }
        // (not PLAIN: the gradients of a not-wall cell that is not fluid - inflow / outflow - are carried from the old buffers; those cells are re-read below
        //  instead of holding rows t+1 of GX / GY across K3 of the two rows above: 8 VGPRs of the boundary bodies)
        const R OV = unpk(PLAIN ? of : sel2(mk.fl(t), of, Fc[t + 2])), OX = unpk(ofx), OY = unpk(ofy);
        if (lm.owner) {
            if (SELF && lv_hot1<T, N>(OV)) {
                bool hf = false, hn = false;
#pragma unroll
                for (int q = 0; q < N; ++q)
                    if (hot1(OV.a[q])) { if (PLAIN || ((mk.fl(t) >> q) & 1u)) hf = true; else hn = true; }
                raise_hot(hot + 3, hf);
                raise_hot(hot, hn);
            }
            if (PLAIN) {         // whole lanes: scalar row base + the lane offset of the loads
                lv_store_row<C, T, N>(out, g, c, i0, j, OV);
                lv_store_row<C, T, N>(gxo, g, c, i0, j, OX);
                lv_store_row<C, T, N>(gyo, g, c, i0, j, OY);
            } else {
                lv_store_row<C, T, N>(out, g, c, i0, j, OV);
                if (mk.fl(t)) {
                    lv_store_row_sel<C, T, N>(gxo, g, c, i0, j, OX, mk.fl(t));
                    lv_store_row_sel<C, T, N>(gyo, g, c, i0, j, OY, mk.fl(t));
                }
                const unsigned carry = mk.nw(t + 1) & ~mk.fl(t) & 3u;
                if (carry) {
                    lv_store_row_sel<C, T, N>(gxo, g, c, i0, j, lv_field<C, T, N>(gxc, g, c, i0, j), carry);
                    lv_store_row_sel<C, T, N>(gyo, g, c, i0, j, lv_field<C, T, N>(gyc, g, c, i0, j), carry);
                }
            }
        }
    }
}

template <int C, int c, int RT, int DM, bool PLAIN, bool CLAMP, int HL = 1>
__device__ __forceinline__ void cip_grad_advect_pk_body(const Grid &g, const Konst<float> &k, int nbx, int nby, int jb, int je,
                                                        float *out, float *gxo, float *gyo, const float *fn, const float *fc,
                                                        const float *gxc, const float *gyc, const float *v, unsigned *hot, const uint8_t *bcmap, int full)
{
    using T = float;
    constexpr int N = 2;
    using R = LV<T, N>;
    constexpr unsigned ALL = 3u;
    constexpr bool SELF = C == 2;
    int wx, ty, cg;
    unsigned cls = 0u;
    if (!tile_coords_nz<N, C, HL>(g, nbx, nby, jb, je, RT, wx, ty, cg, PLAIN ? nullptr : &cls)) return;
    // boundary list of a multi-part launch (one-wave workgroups): bit 1 of the entry's hint = the host saw a fluid cell in the tile's own rows
    // (fs_core.hip tile_list) - the window is then requested WITH the masks instead of behind the test that needs them
    const bool bnd_fluid = !PLAIN && !full && (blockDim.x >> 6) == 1 && (cls & 2u) != 0u;
    const LaneMapN<N> lm_in = lane_map_n<N, HL>(g, wx);
    const LaneMapN<N> lm = PLAIN ? LaneMapN<N>{lm_in.i0, lm_in.owner, false, false} : lm_in;
    const int i0 = lm.i0, j0 = jb + ty * RT;

    // the masks of the window in one word each (MaskPacked: bits 2s, 2s+1 of nwbits = not-wall bits of row j0-1+s, of flbits = fluid bits of row j0+t)
    unsigned nwbits = 0u, flbits = 0u;
    if (!PLAIN) {
#pragma unroll
        for (int s = 0; s < RT + 2; ++s) {
            const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 1 + s));
            nwbits |= lv_sel_nw<N>(m) << (2 * s);
            if (s >= 1 && s <= RT && j0 + s - 1 < je) flbits |= lv_sel_fluid<N>(m) << (2 * (s - 1));
        }
    }
    const MaskPacked mk{nwbits, flbits};
    if (!PLAIN && !bnd_fluid && !__any(flbits != 0u)) {
        // (no fluid cell in this wave's tile: carried values only - see cip_grad_advect_n_body)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int j = j0 + t;
            if (j >= je) break;
            unsigned touch = full ? ALL : mk.nw(t + 1);
            if (SELF && !full) touch |= lv_sel_bit7<N>(lv_bytes<N>(bcmap, g, i0, clampy(g, j)));
            if (!__any(lm.owner && touch != 0u)) continue;
            const R f = lv_field<C, T, N>(fc, g, c, i0, j);
            if (lm.owner && touch) {
                if (SELF) raise_hot(hot, lv_hot1<T, N>(f));
                lv_store<T, N>(out + idx<C, T>(g, c, i0, j), f);
                if (mk.nw(t + 1)) {
                    lv_store_sel<T, N>(gxo + idx<C, T>(g, c, i0, j), lv_field<C, T, N>(gxc, g, c, i0, j), mk.nw(t + 1));
                    lv_store_sel<T, N>(gyo + idx<C, T>(g, c, i0, j), lv_field<C, T, N>(gyc, g, c, i0, j), mk.nw(t + 1));
                }
            }
        }
        return;
    }

    v2f Nn[RT + 4], Fc[RT + 4], GX[RT + 2], GY[RT + 2], AX[RT + 2], AY[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        Nn[u] = pk(lv_field<C, T, N>(fn, g, c, i0, clampy(g, j0 - 2 + u)));
        Fc[u] = pk(lv_field<C, T, N>(fc, g, c, i0, clampy(g, j0 - 2 + u)));
    }
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        GX[s] = pk(lv_field<C, T, N>(gxc, g, c, i0, row));
        GY[s] = pk(lv_field<C, T, N>(gyc, g, c, i0, row));
        if (SELF) {
            if (c == 0) AY[s] = pk(lv_field<2, T, N>(fn, g, 1, i0, row)); else AX[s] = pk(lv_field<2, T, N>(fn, g, 0, i0, row));
        } else {
            AX[s] = pk(lv_field<2, T, N>(v, g, 0, i0, row));
            AY[s] = pk(lv_field<2, T, N>(v, g, 1, i0, row));
        }
    }
    if constexpr (PLAIN) cip_k34_pk_core<C, c, RT, DM, true, CLAMP>(g, k, lm, i0, j0, je, MaskPlain{je - j0}, Nn, Fc, GX, GY, AdvRows<RT>{AX, AY}, out, gxo, gyo, hot);
    else cip_k34_pk_core<C, c, RT, DM, false, CLAMP>(g, k, lm, i0, j0, je, mk, Nn, Fc, GX, GY, AdvRows<RT>{AX, AY}, out, gxo, gyo, hot, gxc, gyc);
}

#ifndef FS_K34_PK
#define FS_K34_PK 1        // f32 lanes of 2 cells: the packed body (0: the scalar body; A/B)
#endif
template <int C, int c, int N, int RT, int DM, bool PLAIN, bool CLAMP, typename T, int HL = 1>
__device__ __forceinline__ void cip_grad_advect_dispatch(const Grid &g, const Konst<T> &k, int nbx, int nby, int jb, int je,
                                                         T *out, T *gxo, T *gyo, const T *fn, const T *fc,
                                                         const T *gxc, const T *gyc, const T *v, unsigned *hot, const uint8_t *bcmap, int full)
{
    if constexpr (FS_K34_PK && N == 2 && sizeof(T) == 4) cip_grad_advect_pk_body<C, c, RT, DM, PLAIN, CLAMP, HL>(g, k, nbx, nby, jb, je, out, gxo, gyo, fn, fc, gxc, gyc, v, hot, bcmap, full);
    else cip_grad_advect_n_body<C, c, N, RT, DM, PLAIN, CLAMP, T, HL>(g, k, nbx, nby, jb, je, out, gxo, gyo, fn, fc, gxc, gyc, v, hot, bcmap, full);
}

// blockIdx.y (or, channel groups innermost / compact lists, the block index >> 3) % C = the channel of this workgroup
template <int C, int N, int RT, int DM, bool PLAIN, bool CLAMP, typename T, int HL = 1>
// (4 waves per SIMD: the packed general body of 2 x 4 tiles comes out at 131 VGPRs unbounded - 3 waves; held to 128 it spills 12 bytes)
#ifndef FS_K34_DYE_WAVES
#define FS_K34_DYE_WAVES 4
#endif
__global__ __launch_bounds__(256, C == 3 ? FS_K34_DYE_WAVES : 4) void k_cip_grad_advect_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je,
                                                           T *out, T *gxo, T *gyo, const T *fn, const T *fc,
                                                           const T *gxc, const T *gyc, const T *v, unsigned *hot, const uint8_t *bcmap, int full)
{
    const int yy = (nby & FS_CG_INNER) ? ((int)blockIdx.x >> 3) : (int)blockIdx.y;
    const int ch = yy % C;
    if (ch == 0) cip_grad_advect_dispatch<C, 0, N, RT, DM, PLAIN, CLAMP, T, HL>(g, k, nbx, nby, jb, je, out, gxo, gyo, fn, fc, gxc, gyc, v, hot, bcmap, full);
    else if (ch == 1) cip_grad_advect_dispatch<C, 1, N, RT, DM, PLAIN, CLAMP, T, HL>(g, k, nbx, nby, jb, je, out, gxo, gyo, fn, fc, gxc, gyc, v, hot, bcmap, full);
    else if (C == 3) cip_grad_advect_dispatch<C, C == 3 ? 2 : 0, N, RT, DM, PLAIN, CLAMP, T, HL>(g, k, nbx, nby, jb, je, out, gxo, gyo, fn, fc, gxc, gyc, v, hot, bcmap, full);
}

// K2 for one row of one component on packed operands (fs/solver.py:234-239, 263-265): f1 the row, fm / fp the rows below / above, and for the
// pressure gradient the row of p (c == 0: x-difference) or the rows of p below / above (c == 1).  The division by Re stays per half.
template <int c, int DM>
__device__ __forceinline__ v2f nonadv_pk_row(const Konst<float> &k, const LaneMapN<2> &lm, v2f fm, v2f f1, v2f fp, v2f p1, v2f pm, v2f pp)
{
    const float l = lv_left<float, 2>(lm, unpk(f1)), r = lv_right<float, 2>(lm, unpk(f1));
    const v2f two_f = 2.0f * f1;
    const v2f d2x = xdiv<DM>((east(f1, r) - two_f) + west(l, f1), k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
    const v2f d2y = xdiv<DM>((fp - two_f) + fm, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
    const v2f lap = d2x + d2y;
    v2f dif;
    dif.x = rdiv<DM>(lap.x, k.re, k.r_re);
    dif.y = rdiv<DM>(lap.y, k.re, k.r_re);
    v2f gp;
    if (c == 0) {
        const float pl = lv_left<float, 2>(lm, unpk(p1)), pr = lv_right<float, 2>(lm, unpk(p1));
        gp = xdiv<DM>(0.5f * ew_diff(p1, pl, pr), k.dx, k.inv_dx, k.r_dx);
    } else {
        gp = xdiv<DM>(0.5f * (pp - pm), k.dx, k.inv_dx, k.r_dx);
    }
    const v2f gg = (-gp) + dif;
    return f1 + gg * k.dt;
}

// ------------------------------------------------------------------------------------------------
// K2  CipMacSolver._non_advection_phase (fs/solver.py:229-240, 263-265) on lanes of N cells, tiles of RT rows: rows j0-1 .. j0+RT of v
// (2 planes) and p are requested up front; not-wall cells get  fn = fc + ((-grad p) + lap(fc)/re) * dt.   One halo lane per side.
// (The quad form with one row per tile, fs_march.h cip_nonadv_quad_tile, requests 9 16-byte rows per output row: 36 B per cell for 12 B
// of input; 2 cells x 4 rows: 18 B per cell.)
// ------------------------------------------------------------------------------------------------
template <int N, int RT, int DM, typename T, int HL = 1>
__global__ __launch_bounds__(256) void k_cip_nonadv_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fn, const T *fc, const T *pc, unsigned *hot, int clear3)
{
    using R = LV<T, N>;
    constexpr int L = N - 1;
    if (clear3 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) hot[3] = 0u;      // every not-wall cell of fn is rewritten (fs_device.h "hot" word [3])
    int wx, ty;
    bool plain;
    if (!tile_coords_hint<N, HL>(g, nbx, nby, jb, je, RT, wx, ty, plain)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned nw[RT];
    if (plain) {                                    // (the list's hint: all fluid - the window is requested without waiting for the masks)
#pragma unroll
        for (int t = 0; t < RT; ++t) nw[t] = j0 + t < je ? (1u << N) - 1u : 0u;
    } else {
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            nw[t] = j0 + t < je ? lv_sel_nw<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 + t))) : 0u;
            any = any || (lm.owner && nw[t] != 0u);
        }
        if (!__any(any)) return;
    }
    R F[2][RT + 2], P[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 2; ++u) {
        const int row = clampy(g, j0 - 1 + u);
        F[0][u] = lv_field<2, T, N>(fc, g, 0, i0, row);
        F[1][u] = lv_field<2, T, N>(fc, g, 1, i0, row);
        P[u] = lv_field<1, T, N>(pc, g, 0, i0, row);
    }
    if constexpr (FS_K34_PK && N == 2 && sizeof(T) == 4) {      // the lane's two cells as one packed operand (fs_device.h v2f): same expression, same bits
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int j = j0 + t;
            if (j >= je) break;
            const v2f p1 = pk(P[t + 1]), pm = pk(P[t]), pp = pk(P[t + 2]);
            const R O0 = unpk(nonadv_pk_row<0, DM>(k, lm, pk(F[0][t]), pk(F[0][t + 1]), pk(F[0][t + 2]), p1, pm, pp));
            const R O1 = unpk(nonadv_pk_row<1, DM>(k, lm, pk(F[1][t]), pk(F[1][t + 1]), pk(F[1][t + 2]), p1, pm, pp));
            if (lm.owner && nw[t]) {
#pragma unroll
                for (int q = 0; q < N; ++q) raise_hot(hot, ((nw[t] >> q) & 1u) && hot2(O0.a[q], O1.a[q]));
                lv_store_sel<T, N>(fn + idx<2, T>(g, 0, i0, j), O0, nw[t]);
                lv_store_sel<T, N>(fn + idx<2, T>(g, 1, i0, j), O1, nw[t]);
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const T pl = lv_left<T, N>(lm, P[t + 1]), pr = lv_right<T, N>(lm, P[t + 1]);
        R O[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const R &fm = F[c][t], &f1 = F[c][t + 1], &fp = F[c][t + 2];
            const T l = lv_left<T, N>(lm, f1), r = lv_right<T, N>(lm, f1);
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const T fE = q == L ? r : f1.a[q == L ? q : q + 1], fW = q == 0 ? l : f1.a[q == 0 ? 0 : q - 1];
                const T f0 = f1.a[q];
                const T d2x = xdiv<DM>((fE - (T)2.0 * f0) + fW, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T d2y = xdiv<DM>((fp.a[q] - (T)2.0 * f0) + fm.a[q], k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T dif = rdiv<DM>(d2x + d2y, k.re, k.r_re);
                T gp;
                if (c == 0) {
                    const T pE = q == L ? pr : P[t + 1].a[q == L ? q : q + 1], pW = q == 0 ? pl : P[t + 1].a[q == 0 ? 0 : q - 1];
                    gp = xdiv<DM>((T)0.5 * (pE - pW), k.dx, k.inv_dx, k.r_dx);
                } else {
                    gp = xdiv<DM>((T)0.5 * (P[t + 2].a[q] - P[t].a[q]), k.dx, k.inv_dx, k.r_dx);
                }
                const T gg = (-gp) + dif;
                O[c].a[q] = f0 + gg * k.dt;
            }
        }
        if (lm.owner && nw[t]) {
#pragma unroll
            for (int q = 0; q < N; ++q) raise_hot(hot, ((nw[t] >> q) & 1u) && hot2(O[0].a[q], O[1].a[q]));
            lv_store_sel<T, N>(fn + idx<2, T>(g, 0, i0, j), O[0], nw[t]);
            lv_store_sel<T, N>(fn + idx<2, T>(g, 1, i0, j), O[1], nw[t]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5 + K6 fused: vorticity confinement in one pass on lanes of N cells (fs/vorticity_confinement.py:27-55; the unfused pair of kernels
// remains available and is what the literal-mode parity tests compare with):
//   w(i,j)  = fluid ? diff_x(v).y - diff_y(v).x : 0
//   vn(i,j) = v + dt*weight * clamp((n.y, -n.x) * w, +-0.1),  n = grad|w| / |grad|w||       on fluid cells
// A lane computes w for RT+2 rows of its cells from RT+4 rows of v, takes the x-neighbours of |w| from the adjacent lanes (DPP) and
// writes RT rows of vn; w / |w| never touch HBM unless STORE_W.  One halo lane per side (the pass reaches 2 cells in x).
// ------------------------------------------------------------------------------------------------
// PLAIN: the launch list says this wave sees nothing but fluid within reach (per-wave hint, fs_core.hip tile_list): no mask loads before the
// window is requested, constant selectors, no lane at the domain's first / last column (round 5: the mask round trip in front of the loads was
// what the literal Jacobi sweep lost 8 % to)
template <int N, int RT, int DM, bool STORE_W, bool PLAIN, typename T>
__device__ __forceinline__ void vort_n_tile(const Grid &g, const Konst<T> &k, const LaneMapN<N> &lm_in, int i0, int j0, int je, const unsigned (&fl_in)[RT + 2],
                                            T *vn, const T *vc, T *vort, T *vort_abs, unsigned *hot)
{
    using R = LV<T, N>;
    constexpr int L = N - 1;
    constexpr unsigned ALL = (1u << N) - 1u;
    const LaneMapN<N> lm = PLAIN ? LaneMapN<N>{lm_in.i0, lm_in.owner, false, false} : lm_in;
    unsigned fl[RT + 2];
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) fl[r] = PLAIN ? ALL : fl_in[r];
    R VX[RT + 4], VY[RT + 4];   // rows j0-2 .. j0+RT+1
#pragma unroll
    for (int r = 0; r < RT + 4; ++r) {
        const int j = clampy(g, j0 - 2 + r);
        VX[r] = lv_field<2, T, N>(vc, g, 0, i0, j);
        VY[r] = lv_field<2, T, N>(vc, g, 1, i0, j);
    }
    // vorticity of rows j0-1 .. j0+RT  (index r <-> v slot r+1): rows were loaded with clamped indices, so for an in-domain row the slots
    // r, r+1, r+2 hold exactly sample()'s rows clamp(j-1), j, clamp(j+1); out-of-domain virtual rows are never consumed (see wm / wp)
    R W[RT + 2];
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) {
        const R &yc = VY[r + 1], &xm = VX[r], &xp = VX[r + 2];
        const T yl = lv_left<T, N>(lm, yc), yr = lv_right<T, N>(lm, yc);
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const T yE = q == L ? yr : yc.a[q == L ? q : q + 1], yW = q == 0 ? yl : yc.a[q == 0 ? 0 : q - 1];
            const T w = div_dx<DM>((T)0.5 * (yE - yW), k) - div_dx<DM>((T)0.5 * (xp.a[q] - xm.a[q]), k);
            W[r].a[q] = (fl[r] >> q) & 1u ? w : (T)0;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int j = j0 + r;
        if (j >= je) break;
        const unsigned f = fl[r + 1];
        const R &wc = W[r + 1];
        // |w| of the clamped neighbour rows: for the first / last domain row the neighbour is the row itself
        const R &wm = (j - 1 < g.jlo) ? W[r + 1] : W[r];
        const R &wp = (j + 1 > g.jhi) ? W[r + 1] : W[r + 2];
        R a;
#pragma unroll
        for (int q = 0; q < N; ++q) a.a[q] = tabs(wc.a[q]);
        const T al = lv_left<T, N>(lm, a), ar = lv_right<T, N>(lm, a);
        if (STORE_W && lm.owner && f) {
            lv_store_sel<T, N>(vort + idx<1, T>(g, 0, i0, j), wc, f);
            lv_store_sel<T, N>(vort_abs + idx<1, T>(g, 0, i0, j), a, f);
        }
        R ox, oy;
        bool h = false;
#pragma unroll
        for (int q = 0; q < N; ++q) {
            const T aE = q == L ? ar : a.a[q == L ? q : q + 1], aW = q == 0 ? al : a.a[q == 0 ? 0 : q - 1];
            T gx = div_dx<DM>((T)0.5 * (aE - aW), k);
            T gy = div_dx<DM>((T)0.5 * (tabs(wp.a[q]) - tabs(wm.a[q])), k);
            const T nrm = tsqrt(gx * gx + gy * gy);
            gx = gx / nrm; gy = gy / nrm;
            T f0 = gy * wc.a[q], f1 = (-gx) * wc.a[q];
            f0 = tmax(tmin(f0, (T)0.1), (T)-0.1);
            f1 = tmax(tmin(f1, (T)0.1), (T)-0.1);
            ox.a[q] = VX[r + 2].a[q] + k.dtw * f0;
            oy.a[q] = VY[r + 2].a[q] + k.dtw * f1;
            h = h || (((f >> q) & 1u) && hot2(ox.a[q], oy.a[q]));
        }
        if (lm.owner && f) {
            raise_hot(hot, h);
            lv_store_sel<T, N>(vn + idx<2, T>(g, 0, i0, j), ox, f);
            lv_store_sel<T, N>(vn + idx<2, T>(g, 1, i0, j), oy, f);
        }
    }
}

template <int N, int RT, int DM, bool STORE_W, typename T>
__global__ __launch_bounds__(256) void k_vort_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *vn, const T *vc, T *vort, T *vort_abs, unsigned *hot, int clear3)
{
    constexpr int HL = 1, OW = 64 - 2 * HL;
    if (clear3 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) hot[3] = 0u;      // every fluid cell of vn is rewritten (fs_device.h "hot" word [3])
    int bx, by, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby, bx, by, cg, 0, &cls)) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = blockDim.x >> 6;
    const int wx = (nby & FS_STACKED) ? bx : bx * nwv + w, ty = (nby & FS_STACKED) ? by * nwv + w : by;
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned fl[RT + 2];   // fluid selectors of rows j0-1 .. j0+RT (clamped rows repeat)
    if ((cls >> w) & 1u) {
#pragma unroll
        for (int r = 0; r < RT + 2; ++r) fl[r] = (1u << N) - 1u;
        vort_n_tile<N, RT, DM, STORE_W, true, T>(g, k, lm, i0, j0, je, fl, vn, vc, vort, vort_abs, hot);
        return;
    }
    bool any = false;
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) {
        fl[r] = lv_sel_fluid<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 1 + r)));
        if (r >= 1 && r <= RT && j0 - 1 + r < je) any |= fl[r] != 0u;
    }
    if (!__any(any)) return;
    vort_n_tile<N, RT, DM, STORE_W, false, T>(g, k, lm, i0, j0, je, fl, vn, vc, vort, vort_abs, hot);
}

// ------------------------------------------------------------------------------------------------
// K2'  MacSolver._update_velocities (fs/solver.py:94-107) on lanes of N cells, tiles of RT rows: upwind (fs/advection.py:12-24, +-1
// stencil) or Kawamura-Kuwahara (fs/advection.py:27-60, +-2 stencil: two DPP hops give the two cells left / right of the lane's).
// ------------------------------------------------------------------------------------------------
#ifndef FS_MAC_PK
#define FS_MAC_PK 1        // f32 lanes of 2 cells: the packed body below (0: the scalar body; A/B)
#endif
// The same update with the lane's two cells as ONE packed operand (round 6; f32, N = 2): every mul / add / sub of the advection sums, the Laplacian
// and the pressure gradient is one v_pk_* for both cells; the Kawamura-Kuwahara weights are chosen per cell by the sign of the advecting component
// ONCE per row for both velocity components (five selects per direction and cell instead of per component); the division by Re stays per half.
// Same expression tree and operation order per cell as the scalar body (fs/advection.py:12-24, 27-60; fs/solver.py:94-107): the same bits.
template <int SCHEME, int RT, int DM>
__device__ __forceinline__ void mac_update_pk_tile(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je, const unsigned (&fl)[RT],
                                                   float *vn, const float *vc, const float *pc, unsigned *hot)
{
    using T = float;
    constexpr int N = 2, R = SCHEME == 0 ? 1 : 2;
    v2f V[2][RT + 2 * R], P[RT + 2];                // slot u of V <-> row j0 - R + u, slot u of P <-> row j0 - 1 + u
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int u = 0; u < RT + 2 * R; ++u) V[c][u] = pk(lv_field<2, T, N>(vc, g, c, i0, clampy(g, j0 - R + u)));
#pragma unroll
    for (int u = 0; u < RT + 2; ++u) P[u] = pk(lv_field<1, T, N>(pc, g, 0, i0, clampy(g, j0 - 1 + u)));
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const v2f ux = V[0][t + R], uy = V[1][t + R];
        const T pl = lv_left<T, N>(lm, unpk(P[t + 1])), pr = lv_right<T, N>(lm, unpk(P[t + 1]));
        v2f fE[2], fW[2], adv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // the cells left / right of the lane's (sample()-clamped at the domain edge)
            const LV<T, N> row = unpk(V[c][t + R]);
            fE[c] = east(V[c][t + R], lv_right<T, N>(lm, row));
            fW[c] = west(lv_left<T, N>(lm, row), V[c][t + R]);
        }
        if (SCHEME == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v2f f0 = V[c][t + R];
                const v2f ax = ux * xdiv<DM>(sel_neg(ux, fE[c] - f0, f0 - fW[c]), k.dx, k.inv_dx, k.r_dx);
                const v2f ay = uy * xdiv<DM>(sel_neg(uy, V[c][t + R + 1] - f0, f0 - V[c][t + R - 1]), k.dx, k.inv_dx, k.r_dx);
                adv[c] = ax + ay;
            }
        } else {
            // weights by the sign of the advecting component (fs/advection.py:36-44, 50-58): the x-sums of both components, then the y-sums - ten weights live at a time
            const v2f wn[5] = {v2f{-2.f, -2.f}, v2f{10.f, 10.f}, v2f{-9.f, -9.f}, v2f{2.f, 2.f}, v2f{-1.f, -1.f}};
            const v2f wp[5] = {v2f{1.f, 1.f}, v2f{-2.f, -2.f}, v2f{9.f, 9.f}, v2f{-10.f, -10.f}, v2f{2.f, 2.f}};
            v2f a[2], b[2];
            {
                v2f w[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) w[i] = sel_neg(ux, wn[i], wp[i]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const LV<T, N> row = unpk(V[c][t + R]);
                    // two cells out: BOTH clamped onto the edge cell at the domain's first / last column
                    T l2 = lane_prev(row.a[0]); if (lm.at_lo) l2 = row.a[0];
                    T r2 = lane_next(row.a[1]); if (lm.at_hi) r2 = row.a[1];
                    const v2f fEE = v2f{fE[c].y, r2}, fWW = v2f{l2, fW[c].x};
                    v2f acc = fEE * w[0];
                    acc = acc + fE[c] * w[1]; acc = acc + V[c][t + R] * w[2]; acc = acc + fW[c] * w[3]; acc = acc + fWW * w[4];
                    a[c] = cdiv<DM>(acc, k.six_dx, k.r_six_dx);
                }
            }
            {
                v2f w[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) w[i] = sel_neg(uy, wn[i], wp[i]);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    v2f acc = V[c][t + 2 * R] * w[0];
                    acc = acc + V[c][t + R + 1] * w[1]; acc = acc + V[c][t + R] * w[2]; acc = acc + V[c][t + R - 1] * w[3]; acc = acc + V[c][t] * w[4];
                    b[c] = cdiv<DM>(acc, k.six_dx, k.r_six_dx);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) adv[c] = ux * a[c] + uy * b[c];
        }
        v2f O[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const v2f f0 = V[c][t + R], fN = V[c][t + R + 1], fS = V[c][t + R - 1];
            v2f gp;
            if (c == 0) gp = xdiv<DM>(0.5f * ew_diff(P[t + 1], pl, pr), k.dx, k.inv_dx, k.r_dx);
            else gp = xdiv<DM>(0.5f * (P[t + 2] - P[t]), k.dx, k.inv_dx, k.r_dx);
            const v2f two_f = 2.0f * f0;
            const v2f d2x = xdiv<DM>((fE[c] - two_f) + fW[c], k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
            const v2f d2y = xdiv<DM>((fN - two_f) + fS, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
            const v2f sum = d2x + d2y;
            v2f lap;
            lap.x = rdiv<DM>(sum.x, k.re, k.r_re);
            lap.y = rdiv<DM>(sum.y, k.re, k.r_re);
            O[c] = f0 + k.dt * (((-adv[c]) - gp) + lap);
        }
        if (lm.owner && fl[t]) {
            const LV<T, N> O0 = unpk(O[0]), O1 = unpk(O[1]);
#pragma unroll
            for (int q = 0; q < N; ++q) raise_hot(hot, ((fl[t] >> q) & 1u) && hot2(O0.a[q], O1.a[q]));
            lv_store_row_sel<2, T, N>(vn, g, 0, i0, j, O0, fl[t]);
            lv_store_row_sel<2, T, N>(vn, g, 1, i0, j, O1, fl[t]);
        }
    }
}

template <int SCHEME, int N, int RT, int DM, typename T>
__global__ __launch_bounds__(256) void k_mac_update_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *vn, const T *vc, const T *pc, unsigned *hot)
{
    using Rw = LV<T, N>;
    constexpr int R = SCHEME == 0 ? 1 : 2;          // stencil radius
    constexpr int HL = 1, L = N - 1;
    int wx, ty;
    bool plain;
    if (!tile_coords_hint<N, HL>(g, nbx, nby, jb, je, RT, wx, ty, plain)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned fl[RT];
    if (plain) {                                    // (the list's hint: all fluid - the window is requested without waiting for the masks)
#pragma unroll
        for (int t = 0; t < RT; ++t) fl[t] = j0 + t < je ? (1u << N) - 1u : 0u;
    } else {
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            fl[t] = j0 + t < je ? lv_sel_fluid<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 + t))) : 0u;
            any = any || (lm.owner && fl[t] != 0u);
        }
        if (!__any(any)) return;
    }
    if constexpr (FS_MAC_PK && N == 2 && sizeof(T) == 4) {
        mac_update_pk_tile<SCHEME, RT, DM>(g, k, lm, i0, j0, je, fl, vn, vc, pc, hot);
        return;
    }
    Rw V[2][RT + 2 * R], P[RT + 2];                 // slot u of V <-> row j0 - R + u, slot u of P <-> row j0 - 1 + u
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int u = 0; u < RT + 2 * R; ++u) V[c][u] = lv_field<2, T, N>(vc, g, c, i0, clampy(g, j0 - R + u));
#pragma unroll
    for (int u = 0; u < RT + 2; ++u) P[u] = lv_field<1, T, N>(pc, g, 0, i0, clampy(g, j0 - 1 + u));
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const T pl = lv_left<T, N>(lm, P[t + 1]), pr = lv_right<T, N>(lm, P[t + 1]);
        Rw O[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const Rw &row = V[c][t + R];
            // the two cells left / right of the lane's (sample()-clamped at the domain edge: BOTH onto the edge cell)
            const T l1 = lv_left<T, N>(lm, row), r1 = lv_right<T, N>(lm, row);
            T l2 = l1, r2 = r1;
            if (SCHEME == 1) {
                l2 = lane_prev(row.a[N - 2]); if (lm.at_lo) l2 = row.a[0];
                r2 = lane_next(row.a[1]); if (lm.at_hi) r2 = row.a[L];
            }
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const T ux = V[0][t + R].a[q], uy = V[1][t + R].a[q];
                const T f0 = row.a[q];
                const T fE = q == L ? r1 : row.a[q == L ? q : q + 1], fW = q == 0 ? l1 : row.a[q == 0 ? 0 : q - 1];
                const T fN = V[c][t + R + 1].a[q], fS = V[c][t + R - 1].a[q];
                T adv;
                if (SCHEME == 0) {
                    const T ax = ux * xdiv<DM>(ux < (T)0.0 ? (fE - f0) : (f0 - fW), k.dx, k.inv_dx, k.r_dx);
                    const T ay = uy * xdiv<DM>(uy < (T)0.0 ? (fN - f0) : (f0 - fS), k.dx, k.inv_dx, k.r_dx);
                    adv = ax + ay;
                } else {
                    const T fEE = q + 2 <= L ? row.a[q + 2 <= L ? q + 2 : L] : (q + 2 == N ? r1 : r2);
                    const T fWW = q >= 2 ? row.a[q >= 2 ? q - 2 : 0] : (q == 1 ? l1 : l2);
                    const bool nx = ux < (T)0;
                    T w0 = nx ? (T)-2 : (T)1, w1 = nx ? (T)10 : (T)-2, w2 = nx ? (T)-9 : (T)9, w3 = nx ? (T)2 : (T)-10, w4 = nx ? (T)-1 : (T)2;
                    T acc = fEE * w0;
                    acc = acc + fE * w1; acc = acc + f0 * w2; acc = acc + fW * w3; acc = acc + fWW * w4;
                    const T a = cdiv<DM>(acc, k.six_dx, k.r_six_dx);
                    const bool ny = uy < (T)0;
                    w0 = ny ? (T)-2 : (T)1; w1 = ny ? (T)10 : (T)-2; w2 = ny ? (T)-9 : (T)9; w3 = ny ? (T)2 : (T)-10; w4 = ny ? (T)-1 : (T)2;
                    acc = V[c][t + 2 * R].a[q] * w0;
                    acc = acc + fN * w1; acc = acc + f0 * w2; acc = acc + fS * w3; acc = acc + V[c][t].a[q] * w4;
                    const T b = cdiv<DM>(acc, k.six_dx, k.r_six_dx);
                    adv = ux * a + uy * b;
                }
                T gp;
                if (c == 0) {
                    const T pE = q == L ? pr : P[t + 1].a[q == L ? q : q + 1], pW = q == 0 ? pl : P[t + 1].a[q == 0 ? 0 : q - 1];
                    gp = xdiv<DM>((T)0.5 * (pE - pW), k.dx, k.inv_dx, k.r_dx);
                } else {
                    gp = xdiv<DM>((T)0.5 * (P[t + 2].a[q] - P[t].a[q]), k.dx, k.inv_dx, k.r_dx);
                }
                const T d2x = xdiv<DM>((fE - (T)2.0 * f0) + fW, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T d2y = xdiv<DM>((fN - (T)2.0 * f0) + fS, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T lap = rdiv<DM>(d2x + d2y, k.re, k.r_re);
                O[c].a[q] = f0 + k.dt * (((-adv) - gp) + lap);
            }
        }
        if (lm.owner && fl[t]) {
#pragma unroll
            for (int q = 0; q < N; ++q) raise_hot(hot, ((fl[t] >> q) & 1u) && hot2(O[0].a[q], O[1].a[q]));
            lv_store_sel<T, N>(vn + idx<2, T>(g, 0, i0, j), O[0], fl[t]);
            lv_store_sel<T, N>(vn + idx<2, T>(g, 1, i0, j), O[1], fl[t]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8R fused on lanes of N cells: one red-black SOR iteration (odd pass p.cur -> p.next, then even pass in place on p.next;
// fs/pressure_updater.py:92-114) as ONE kernel.
//
// Even cell (i, j) reads its four (odd) neighbours of p.next AFTER the odd pass.  A lane therefore first
// forms "pnO" = p.next as it stands after the odd pass for rows j0-1 .. j0+RT of its cells:
//     odd fluid cell : (1-w) p.cur + w predict_p(p.cur)       (recomputed redundantly in the halo rows / lanes)
//     any other cell : the stored p.next value                 (stale data the reference also reads, H5)
// and then relaxes the even fluid cells of rows j0 .. j0+RT-1 from pnO, taking x-neighbours from the adjacent
// lanes (DPP).  Only cells this lane owns are stored.  Concurrent tiles never consume a value another tile
// stores: halo pnO values are either recomputed from p.cur (read-only here) or belong to non-fluid cells
// (never written).  Traffic: mask 1 + p.cur 4 + p.next 4 + v 8 read, p.next 4 written = 21 B/cell per
// iteration instead of 2 x 17 for the two half-sweep launches.  One halo lane per side.
// ------------------------------------------------------------------------------------------------
// one colour of one row: cells c with ((c + PAR) & 1) == COLOR and a fluid bit are relaxed from centre row pc, neighbours pm (j-1), pp (j+1)
template <int PAR, int COLOR, int DM, typename T, int N>
__device__ __forceinline__ void lv_rb_relax_row(const Konst<T> &k, const LaneMapN<N> &lm, unsigned fluid,
                                                const LV<T, N> &pm, const LV<T, N> &pc, const LV<T, N> &pp,
                                                const LV<T, N> &xm, const LV<T, N> &xc, const LV<T, N> &xp,
                                                const LV<T, N> &ym, const LV<T, N> &yc, const LV<T, N> &yp, LV<T, N> &out)
{
    constexpr int L = N - 1;
    const T pl = lv_left<T, N>(lm, pc), pr = lv_right<T, N>(lm, pc);
    const T xl = lv_left<T, N>(lm, xc), xr = lv_right<T, N>(lm, xc);
    const T yl = lv_left<T, N>(lm, yc), yr = lv_right<T, N>(lm, yc);
#pragma unroll
    for (int c = 0; c < N; ++c) {
        if (((c + PAR) & 1) != COLOR) continue;
        const T pE = c == L ? pr : pc.a[c == L ? c : c + 1], pW = c == 0 ? pl : pc.a[c == 0 ? 0 : c - 1];
        const T xE = c == L ? xr : xc.a[c == L ? c : c + 1], xW = c == 0 ? xl : xc.a[c == 0 ? 0 : c - 1];
        const T yE = c == L ? yr : yc.a[c == L ? c : c + 1], yW = c == 0 ? yl : yc.a[c == 0 ? 0 : c - 1];
        T s2, s3;
        source_from<DM>(k, xE, xW, yE, yW, xp.a[c], xm.a[c], yp.a[c], ym.a[c], s2, s3);
        const T pred = predict_from(pE, pW, pp.a[c], pm.a[c], s2, s3);
        const T val = k.om1 * pc.a[c] + k.om * pred;
        out.a[c] = (fluid & (1u << c)) ? val : out.a[c];
    }
}

template <int N, int RT, int DM, typename T>
__global__ __launch_bounds__(256) void k_rbsor_iter_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *pn, const T *pc, const T *vc)
{
    using R = LV<T, N>;
    constexpr int HL = 1;
    int wx, ty, cg;
    if (!tile_coords_nz<N, 1, HL>(g, nbx, nby, jb, je, RT, wx, ty, cg)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;

    unsigned fl[RT + 2];   // rows j0-1 .. j0+RT
    bool any = false;
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) {
        fl[r] = lv_sel_fluid<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 1 + r)));
        if (r >= 1 && r <= RT && j0 - 1 + r < je) any |= fl[r] != 0u;
    }
    if (!__any(any)) return;

    R PC[RT + 4], VX[RT + 4], VY[RT + 4];        // rows j0-2 .. j0+RT+1 (clamped)
#pragma unroll
    for (int r = 0; r < RT + 4; ++r) {
        const int j = clampy(g, j0 - 2 + r);
        PC[r] = lv_field<1, T, N>(pc, g, 0, i0, j);
        VX[r] = lv_field<2, T, N>(vc, g, 0, i0, j);
        VY[r] = lv_field<2, T, N>(vc, g, 1, i0, j);
    }
    R PO[RT + 2];                                // p.next after the odd pass, rows j0-1 .. j0+RT
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) PO[r] = lv_field<1, T, N>(pn, g, 0, i0, clampy(g, j0 - 1 + r));
    // odd pass on rows j0-1 .. j0+RT (slot r <-> field slot r+1); i0 is even, so the colour of cell c in row j is (c + ybase + j) & 1
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) {
        const int j = j0 - 1 + r;
        if (j < g.jlo || j > g.jhi) continue;    // virtual row outside the domain: never consumed
        if ((g.ybase + j) & 1) lv_rb_relax_row<1, 1, DM>(k, lm, fl[r], PC[r], PC[r + 1], PC[r + 2], VX[r], VX[r + 1], VX[r + 2], VY[r], VY[r + 1], VY[r + 2], PO[r]);
        else                   lv_rb_relax_row<0, 1, DM>(k, lm, fl[r], PC[r], PC[r + 1], PC[r + 2], VX[r], VX[r + 1], VX[r + 2], VY[r], VY[r + 1], VY[r + 2], PO[r]);
    }
    // even pass on rows j0 .. j0+RT-1, in place on the odd-pass state; the clamped neighbour of the first / last domain row is the row itself.
    // All rows are relaxed BEFORE anything is stored: pn is input and output of this kernel.
    R OUT[RT];
#pragma unroll
    for (int r = 1; r <= RT; ++r) {
        const int j = j0 - 1 + r;
        if (j >= je) break;
        const R &pm = (j - 1 < g.jlo) ? PO[r] : PO[r - 1];
        const R &pp = (j + 1 > g.jhi) ? PO[r] : PO[r + 1];
        const R ctr = PO[r];
        R out = ctr;
        if ((g.ybase + j) & 1) lv_rb_relax_row<1, 0, DM>(k, lm, fl[r], pm, ctr, pp, VX[r], VX[r + 1], VX[r + 2], VY[r], VY[r + 1], VY[r + 2], out);
        else                   lv_rb_relax_row<0, 0, DM>(k, lm, fl[r], pm, ctr, pp, VX[r], VX[r + 1], VX[r + 2], VY[r], VY[r + 1], VY[r + 2], out);
        OUT[r - 1] = out;
    }
#pragma unroll
    for (int r = 1; r <= RT; ++r) {
        const int j = j0 - 1 + r;
        if (j >= je) break;
        if (lm.owner && fl[r]) lv_store_sel<T, N>(pn + idx<1, T>(g, 0, i0, j), OUT[r - 1], fl[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// Source pair of predict_p (fs/pressure_updater.py:23-38) for every cell of the row range, once per step for the Jacobi runs that read
// it (fs_kernels.h k_poisson_source is the one-cell-per-lane form): lanes of N cells, tiles of RT rows, one halo lane per side.
// ------------------------------------------------------------------------------------------------
template <int N, int RT, int DM, typename T>
__global__ __launch_bounds__(256) void k_poisson_source_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *src, const T *vc)
{
    using R = LV<T, N>;
    constexpr int HL = 1, L = N - 1;
    int wx, ty, cg;
    if (!tile_coords_nz<N, 1, HL>(g, nbx, nby, jb, je, RT, wx, ty, cg)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    R VX[RT + 2], VY[RT + 2];                    // rows j0-1 .. j0+RT (clamped: sample())
#pragma unroll
    for (int u = 0; u < RT + 2; ++u) {
        const int row = clampy(g, j0 - 1 + u);
        VX[u] = lv_field<2, T, N>(vc, g, 0, i0, row);
        VY[u] = lv_field<2, T, N>(vc, g, 1, i0, row);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const T xl = lv_left<T, N>(lm, VX[t + 1]), xr = lv_right<T, N>(lm, VX[t + 1]);
        const T yl = lv_left<T, N>(lm, VY[t + 1]), yr = lv_right<T, N>(lm, VY[t + 1]);
        R S2, S3;
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const T xE = c == L ? xr : VX[t + 1].a[c == L ? c : c + 1], xW = c == 0 ? xl : VX[t + 1].a[c == 0 ? 0 : c - 1];
            const T yE = c == L ? yr : VY[t + 1].a[c == L ? c : c + 1], yW = c == 0 ? yl : VY[t + 1].a[c == 0 ? 0 : c - 1];
            source_from<DM>(k, xE, xW, yE, yW, VX[t + 2].a[c], VX[t].a[c], VY[t + 2].a[c], VY[t].a[c], S2.a[c], S3.a[c]);
        }
        if (lm.owner) {
            lv_store<T, N>(src + idx<2, T>(g, 0, i0, j), S2);
            lv_store<T, N>(src + idx<2, T>(g, 1, i0, j), S3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K12  DyeCipMacSolver._non_advection_phase_dye (fs/solver.py:378-383) on lanes of N cells, tiles of RT rows: dn = dc + (lap(dc)/re) dt on
// not-wall cells, the three channels one after the other.  One halo lane per side.
// ------------------------------------------------------------------------------------------------
template <int N, int RT, int DM, typename T, int HL = 1>
__global__ __launch_bounds__(256) void k_cip_nonadv_dye_n(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *dn, const T *dc)
{
    using R = LV<T, N>;
    constexpr int L = N - 1;
    int wx, ty;
    bool plain;
    if (!tile_coords_hint<N, HL>(g, nbx, nby, jb, je, RT, wx, ty, plain)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned nw[RT];
    if (plain) {
#pragma unroll
        for (int t = 0; t < RT; ++t) nw[t] = j0 + t < je ? (1u << N) - 1u : 0u;
    } else {
        bool any = false;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            nw[t] = j0 + t < je ? lv_sel_nw<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 + t))) : 0u;
            any = any || (lm.owner && nw[t] != 0u);
        }
        if (!__any(any)) return;
    }
    R D[3][RT + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int u = 0; u < RT + 2; ++u) D[c][u] = lv_field<3, T, N>(dc, g, c, i0, clampy(g, j0 - 1 + u));
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int j = j0 + t;
            if (j >= je) break;
            const R &f1 = D[c][t + 1];
            const T l = lv_left<T, N>(lm, f1), r = lv_right<T, N>(lm, f1);
            R O;
#pragma unroll
            for (int q = 0; q < N; ++q) {
                const T f0 = f1.a[q];
                const T fE = q == L ? r : f1.a[q == L ? q : q + 1], fW = q == 0 ? l : f1.a[q == 0 ? 0 : q - 1];
                const T d2x = xdiv<DM>((fE - (T)2.0 * f0) + fW, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T d2y = xdiv<DM>((D[c][t + 2].a[q] - (T)2.0 * f0) + D[c][t].a[q], k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
                const T dif = rdiv<DM>(d2x + d2y, k.re, k.r_re);
                O.a[q] = f0 + dif * k.dt;
            }
            if (lm.owner && nw[t]) lv_store_sel<T, N>(dn + idx<3, T>(g, c, i0, j), O, nw[t]);
        }
}

}  // namespace fs
