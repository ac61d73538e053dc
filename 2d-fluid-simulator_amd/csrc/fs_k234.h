// fs_k234.h - K2 + K3 + K4 of the CIP velocity step in ONE pass over HBM, for the tiles that see nothing but fluid.
//
// Reference: fs/solver.py:213-227 (_update_velocities): K2 _non_advection_phase :229-240 writes v.next from v.current and p, the buffers
// stay, K3 _non_advection_phase_grad :242-261 and K4 _advection_phase :267-332 read v.next back on a 5 x 5 neighbourhood.  As kernels of
// their own K2 moves 20 B per cell and the fused K3+K4 pass (fs_k34n.h) reads its 8 B per cell again: at res 4096 that is 540 MB of the
// 2.7 GB a step moves.  Here K2 is evaluated where it is consumed:
//
//   * a workgroup is TWO waves on one tile of RT rows x 120 cells (lanes of 2 cells, 60 owner lanes + 2 halo lanes per side: K2 o K3 o K4
//     reaches 3 cells) - wave c owns velocity component c, as in fs_k34n.h;
//   * wave c evaluates K2 for ITS component on rows j0-2 .. j0+RT+1 from rows j0-3 .. j0+RT+2 of v.current and p (packed f32, fs_device.h
//     v2f) and keeps the result in registers: the "Nn" window of the K3+K4 core;
//   * K4 advects with BOTH components of the post-K2 velocity: rows j0-1 .. j0+RT of a wave's result go to LDS (6 KB per workgroup), one
//     barrier, and the sibling wave reads them as its advecting component;
//   * K3 + K4: the packed core of fs_k34n.h, unchanged.
//
// Same expression trees, same operation order, one rounding per operation: the bits of the three-kernel sequence (tests compare at
// tolerance 0).  What is NOT the same is the content of the intermediate buffer: the post-K2 velocity of these tiles never reaches HBM.
// Nothing reads it there - the reference's own sequence overwrites every fluid cell of that buffer (K2 of the next step, or the vorticity
// confinement of this one) before anything looks at it.  The tiles that are NOT all fluid take the same route with the masks (k234_bnd_phase1 / 2
// below), and ONE launch covers both kinds (k_cip_step_all: the list entry's hint picks the body; fs_transport.hip fs_cip_step).  Until late in
// round 5 the other tiles ran the general K3 + K4 kernel of fs_k34n.h behind a K2 launch over their rows.
// Plain tile: every cell within 2 rows and within the halo lanes is fluid and inside the domain (fs_core.hip tile_list) - K2's own reads
// one cell further out take whatever the buffers hold there, as the reference's K2 does.
#pragma once
#include <type_traits>

#include "fs_k34n.h"

namespace fs {

// The loads take a scalar row base + ONE 32-bit lane offset (the `saddr` form, fs_march.h load_row_quad).  hipcc selects that form only when it
// sees the zero-extension of the offset in the basic block of the load; the two component bodies of this kernel are separate blocks and the
// extension of a value computed in front of them is hoisted out - each load then pays a 64-bit VALU add.  An empty asm makes the offset opaque
// inside the body, so the extension stays there.
#if defined(__HIP_DEVICE_COMPILE__)
#define FS_PIN_LANE_OFFSET(i) asm volatile("" : "+v"(i))
#else
#define FS_PIN_LANE_OFFSET(i) (void)(i)
#endif

template <int RT> struct K234State {
    v2f Nn[RT + 4], Fc[RT + 4], GX[RT + 2], GY[RT + 2];
};

// K2 for component c of one wave: rows j0-2 .. j0+RT+1 (slot u <-> row j0-2+u), every cell not-wall by the launch's construction.
//   fn = fc + ((-grad p) + (diff2_x(fc) + diff2_y(fc)) / re) * dt      fs/solver.py:234-239, 263-265
template <int c, int RT, int DM>
__device__ __forceinline__ void k234_nonadv(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0,
                                            const float *fc, const float *pc, K234State<RT> &st)
{
    using T = float;
    constexpr int N = 2;
    v2f F[RT + 6];                               // rows j0-3 .. j0+RT+2
    v2f P[RT + 6];                               // c == 0: rows j0-2 .. j0+RT+1 in slots 1 .. RT+4;  c == 1: rows j0-3 .. j0+RT+2
#pragma unroll
    for (int u = 0; u < RT + 6; ++u) {
        const int row = clampy(g, j0 - 3 + u);
        F[u] = pk(lv_field<2, T, N>(fc, g, c, i0, row));
        if (c == 1 || (u >= 1 && u <= RT + 4)) P[u] = pk(lv_field<1, T, N>(pc, g, 0, i0, row));
    }
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        st.Nn[u] = nonadv_pk_row<c, DM>(k, lm, F[u], F[u + 1], F[u + 2], P[u + 1], P[c == 0 ? u + 1 : u], P[c == 0 ? u + 1 : u + 2]);
        st.Fc[u] = F[u + 1];
    }
}

template <int c, int RT, int DM>
__device__ __forceinline__ void k234_phase1(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0,
                                            const float *fc, const float *pc, const float *gxc, const float *gyc, K234State<RT> &st, v2f (*xch)[64])
{
    using T = float;
    constexpr int N = 2;
    FS_PIN_LANE_OFFSET(i0);
    // the old gradients of the tile: requested before K2 so that they are on their way while it runs
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        st.GX[s] = pk(lv_field<2, T, N>(gxc, g, c, i0, row));
        st.GY[s] = pk(lv_field<2, T, N>(gyc, g, c, i0, row));
    }
    k234_nonadv<c, RT, DM>(g, k, lm, i0, j0, fc, pc, st);
    // rows j0-1 .. j0+RT of this component: the sibling wave's advecting velocity
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) xch[c * (RT + 2) + s][lane] = st.Nn[s + 1];
}

template <int c, int RT, int DM>
__device__ __forceinline__ void k234_phase2(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je,
                                            float *out, float *gxo, float *gyo, unsigned *hot, const K234State<RT> &st, const v2f (*xch)[64])
{
    const int lane = threadIdx.x & 63;
    cip_k34_pk_core<2, c, RT, DM, true, false>(g, k, lm, i0, j0, je, MaskPlain{je - j0}, st.Nn, st.Fc, st.GX, st.GY, AdvLds<RT>{xch + (1 - c) * (RT + 2), lane}, out, gxo, gyo, hot);
}

// one workgroup = 2 waves = the two velocity components of ONE listed tile (compact launch only: Grid::tiles, one entry per tile)
template <int RT, int DM>
__global__ __launch_bounds__(128) void k_cip_step_plain(Grid g, Konst<float> k, int nbx, int nby, int jb, int je,
                                                        float *out, float *gxo, float *gyo, float *, const float *fc, const float *pc,
                                                        const float *gxc, const float *gyc, unsigned *hot, unsigned *)      // (the argument list of k_cip_step_bnd / _all)
{
    constexpr int N = 2, HL = 2, OW = 64 - 2 * HL;
    __shared__ v2f xch[2 * (RT + 2)][64];
    int wx, ty, cg;
    if (!band_coords<1>(g, nbx, nby, wx, ty, cg)) return;                        // (workgroup-uniform: both waves leave, or neither)
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LaneMapN<N> lm_in = lane_map_n<N, HL>(g, wx);
    const LaneMapN<N> lm{lm_in.i0, lm_in.owner, false, false};                  // (a plain tile holds no lane at the domain's first / last column)
    const int i0 = lm.i0, j0 = jb + ty * RT;
    K234State<RT> st;
    if (c == 0) k234_phase1<0, RT, DM>(g, k, lm, i0, j0, fc, pc, gxc, gyc, st, xch);
    else        k234_phase1<1, RT, DM>(g, k, lm, i0, j0, fc, pc, gxc, gyc, st, xch);
    __syncthreads();
    if (c == 0) k234_phase2<0, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, hot, st, xch);
    else        k234_phase2<1, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, hot, st, xch);
}

// ---- the boundary tiles of the same call -------------------------------------------------------------------------------------------------
// Same two waves per tile, with the masks: K2 is the reference's on every not-wall cell of the window (clamped rows, clamped edge lanes - the
// lane map and row function of k_cip_nonadv_n), a WALL cell takes the value the intermediate buffer holds (nothing writes it in this step;
// the load is predicated on the lanes that have one), and a slot that stands for a row outside the domain takes the slot of the edge row it
// clamps onto.  The window then is, bit for bit, what the general K3 + K4 kernel (fs_k34n.h) reads from memory behind a K2 launch - without
// that launch and without the 8 B per cell it wrote and the 14 rows per wave that were read back.  Stored from K2: the not-wall cells of
// the tile's own rows that are not fluid (inflow / outflow: the reference's buffer keeps them, and a later step may look at them as stale
// data); the fluid cells are dead as on the plain tiles.  K3 + K4: the packed core with its selectors.
template <int c, int RT, int DM>
__device__ __forceinline__ void k234_bnd_phase1(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je,
                                                float *fn, const float *fc, const float *pc, const float *gxc, const float *gyc, unsigned *hot_fn,
                                                K234State<RT> &st, unsigned &nwbits, unsigned &flbits, v2f (*xch)[64])
{
    using T = float;
    constexpr int N = 2;
    FS_PIN_LANE_OFFSET(i0);
    // the masks of the window: bits 2u, 2u+1 of nwbits = not-wall bits of row j0-2+u; bits 2t, 2t+1 of flbits = fluid bits of row j0+t
    nwbits = 0u; flbits = 0u;
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 2 + u));
        nwbits |= lv_sel_nw<N>(m) << (2 * u);
        if (u >= 2 && u < RT + 2 && j0 + u - 2 < je) flbits |= lv_sel_fluid<N>(m) << (2 * (u - 2));
    }
#define FS_NWK(u) ((nwbits >> (2 * (u))) & 3u)
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        st.GX[s] = pk(lv_field<2, T, N>(gxc, g, c, i0, row));
        st.GY[s] = pk(lv_field<2, T, N>(gyc, g, c, i0, row));
    }
    v2f F[RT + 6], P[RT + 6];
#pragma unroll
    for (int u = 0; u < RT + 6; ++u) {
        const int row = clampy(g, j0 - 3 + u);
        F[u] = pk(lv_field<2, T, N>(fc, g, c, i0, row));
        if (c == 1 || (u >= 1 && u <= RT + 4)) P[u] = pk(lv_field<1, T, N>(pc, g, 0, i0, row));
    }
    // what the buffer holds on wall cells
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        st.Nn[u] = v2f{0.0f, 0.0f};
        if (FS_NWK(u) != 3u) st.Nn[u] = pk(lv_field<2, T, N>(fn, g, c, i0, clampy(g, j0 - 2 + u)));
    }
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        const v2f n = nonadv_pk_row<c, DM>(k, lm, F[u], F[u + 1], F[u + 2], P[u + 1], P[c == 0 ? u + 1 : u], P[c == 0 ? u + 1 : u + 2]);
        st.Nn[u] = sel2(FS_NWK(u), n, st.Nn[u]);
        st.Fc[u] = F[u + 1];
    }
    // slots of rows outside the domain (wave-uniform): K2 of the edge row, which was evaluated from ITS neighbours
#pragma unroll
    for (int u = RT + 2; u >= 0; --u) if (j0 - 2 + u < g.jlo) st.Nn[u] = st.Nn[u + 1];
#pragma unroll
    for (int u = 1; u < RT + 4; ++u) if (j0 - 2 + u > g.jhi) st.Nn[u] = st.Nn[u - 1];
    // K2's own output where it outlives the step
    if (lm.owner) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const unsigned sel = FS_NWK(t + 2) & ~(flbits >> (2 * t)) & 3u;
            if (j0 + t < je && sel) {
                const LV<T, N> O = unpk(st.Nn[t + 2]);
#pragma unroll
                for (int q = 0; q < N; ++q) raise_hot(hot_fn, ((sel >> q) & 1u) && hot1(O.a[q]));
                lv_store_row_sel<2, T, N>(fn, g, c, i0, j0 + t, O, sel);
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) xch[c * (RT + 2) + s][lane] = st.Nn[s + 1];
#undef FS_NWK
}

template <int c, int RT, int DM>
__device__ __forceinline__ void k234_bnd_phase2(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je,
                                                float *out, float *gxo, float *gyo, const float *gxc, const float *gyc, unsigned *hot, const K234State<RT> &st,
                                                unsigned nwbits, unsigned flbits, const v2f (*xch)[64])
{
    const int lane = threadIdx.x & 63;
    cip_k34_pk_core<2, c, RT, DM, false, false>(g, k, lm, i0, j0, je, MaskPacked{nwbits >> 2, flbits}, st.Nn, st.Fc, st.GX, st.GY, AdvLds<RT>{xch + (1 - c) * (RT + 2), lane}, out, gxo, gyo, hot, gxc, gyc);
}

// one workgroup = 2 waves = the two velocity components of ONE listed boundary tile (compact launch, class 2 list with one entry per tile)
#ifndef FS_K234_BND_WAVES
#define FS_K234_BND_WAVES 4
#endif
template <int RT, int DM>
__global__ __launch_bounds__(128, FS_K234_BND_WAVES) void k_cip_step_bnd(Grid g, Konst<float> k, int nbx, int nby, int jb, int je,
                                                      float *out, float *gxo, float *gyo, float *fn, const float *fc, const float *pc,
                                                      const float *gxc, const float *gyc, unsigned *hot, unsigned *hot_fn)
{
    constexpr int N = 2, HL = 2, OW = 64 - 2 * HL;
    __shared__ v2f xch[2 * (RT + 2)][64];
    int wx, ty, cg;
    if (!band_coords<1>(g, nbx, nby, wx, ty, cg)) return;
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    K234State<RT> st;
    unsigned nwbits, flbits;
    if (c == 0) k234_bnd_phase1<0, RT, DM>(g, k, lm, i0, j0, je, fn, fc, pc, gxc, gyc, hot_fn, st, nwbits, flbits, xch);
    else        k234_bnd_phase1<1, RT, DM>(g, k, lm, i0, j0, je, fn, fc, pc, gxc, gyc, hot_fn, st, nwbits, flbits, xch);
    __syncthreads();
    if (c == 0) k234_bnd_phase2<0, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, gxc, gyc, hot, st, nwbits, flbits, xch);
    else        k234_bnd_phase2<1, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, gxc, gyc, hot, st, nwbits, flbits, xch);
}

// ... and both kinds of tile in ONE launch (class 0 list with the per-tile plain hint): the boundary tiles are a launch of 8 569 workgroups at bc5 res
// 4096 - a few rounds of what the chip holds - and the all-fluid launch ends on a partly filled round too; in one launch the boundary tiles fill in as
// others finish.  One register budget: the boundary body fits the all-fluid body's 128 VGPRs since the selective stores take the scalar row base, the
// carried gradients are re-read and the sibling's rows come from LDS where they are used.
template <int RT, int DM>
__global__ __launch_bounds__(128, 4) void k_cip_step_all(Grid g, Konst<float> k, int nbx, int nby, int jb, int je,
                                                         float *out, float *gxo, float *gyo, float *fn, const float *fc, const float *pc,
                                                         const float *gxc, const float *gyc, unsigned *hot, unsigned *hot_fn)
{
    constexpr int N = 2, HL = 2, OW = 64 - 2 * HL;
    __shared__ v2f xch[2 * (RT + 2)][64];
    int wx, ty, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby, wx, ty, cg, 0, &cls)) return;
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const LaneMapN<N> lm_in = lane_map_n<N, HL>(g, wx);
    const int i0 = lm_in.i0, j0 = jb + ty * RT;
    K234State<RT> st;
    if (cls & 1u) {                                                             // (workgroup-uniform: the list's hint for this tile)
        const LaneMapN<N> lm{lm_in.i0, lm_in.owner, false, false};
        if (c == 0) k234_phase1<0, RT, DM>(g, k, lm, i0, j0, fc, pc, gxc, gyc, st, xch);
        else        k234_phase1<1, RT, DM>(g, k, lm, i0, j0, fc, pc, gxc, gyc, st, xch);
        __syncthreads();
        if (c == 0) k234_phase2<0, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, hot, st, xch);
        else        k234_phase2<1, RT, DM>(g, k, lm, i0, j0, je, out, gxo, gyo, hot, st, xch);
    } else {
        unsigned nwbits, flbits;
        if (c == 0) k234_bnd_phase1<0, RT, DM>(g, k, lm_in, i0, j0, je, fn, fc, pc, gxc, gyc, hot_fn, st, nwbits, flbits, xch);
        else        k234_bnd_phase1<1, RT, DM>(g, k, lm_in, i0, j0, je, fn, fc, pc, gxc, gyc, hot_fn, st, nwbits, flbits, xch);
        __syncthreads();
        if (c == 0) k234_bnd_phase2<0, RT, DM>(g, k, lm_in, i0, j0, je, out, gxo, gyo, gxc, gyc, hot, st, nwbits, flbits, xch);
        else        k234_bnd_phase2<1, RT, DM>(g, k, lm_in, i0, j0, je, out, gxo, gyo, gxc, gyc, hot, st, nwbits, flbits, xch);
    }
}

// ---- the dye: K12 + K3 + K4 (fs/solver.py:385-401 _update_dye) over the all-fluid tiles -----------------------------------------------
// K12 (_non_advection_phase_dye :378-383: dn = dc + (lap(dc) / re) dt, no pressure term) of one channel needs nothing from the others, and
// the advecting velocity is the finished flow step's - read from memory: one wave per tile and channel, no exchange (k_cip_dye below).
// K12 for one row of one channel on packed operands: dn = dc + (lap(dc) / re) dt
template <int DM>
__device__ __forceinline__ v2f nonadv_dye_pk_row(const Konst<float> &k, const LaneMapN<2> &lm, v2f fm, v2f f1, v2f fp)
{
    const float l = lv_left<float, 2>(lm, unpk(f1)), r = lv_right<float, 2>(lm, unpk(f1));
    const v2f two_f = 2.0f * f1;
    const v2f d2x = xdiv<DM>((east(f1, r) - two_f) + west(l, f1), k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
    const v2f d2y = xdiv<DM>((fp - two_f) + fm, k.dx_sq, k.inv_dx_sq, k.r_dx_sq);
    const v2f lap = d2x + d2y;
    v2f dif;
    dif.x = rdiv<DM>(lap.x, k.re, k.r_re);
    dif.y = rdiv<DM>(lap.y, k.re, k.r_re);
    return f1 + dif * k.dt;
}

// one channel of one all-fluid tile
template <int c, int RT, int DM, bool CLAMP>
__device__ __forceinline__ void k234_dye_plain(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je,
                                               float *out, float *gxo, float *gyo, const float *fc, const float *gxc, const float *gyc, const float *v)
{
    using T = float;
    constexpr int N = 2;
    v2f F[RT + 6], Nn[RT + 4], Fc[RT + 4], GX[RT + 2], GY[RT + 2], AX[RT + 2], AY[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 6; ++u) F[u] = pk(lv_field<3, T, N>(fc, g, c, i0, clampy(g, j0 - 3 + u)));
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        GX[s] = pk(lv_field<3, T, N>(gxc, g, c, i0, row));
        GY[s] = pk(lv_field<3, T, N>(gyc, g, c, i0, row));
        AX[s] = pk(lv_field<2, T, N>(v, g, 0, i0, row));
        AY[s] = pk(lv_field<2, T, N>(v, g, 1, i0, row));
    }
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        Nn[u] = nonadv_dye_pk_row<DM>(k, lm, F[u], F[u + 1], F[u + 2]);
        Fc[u] = F[u + 1];
    }
    cip_k34_pk_core<3, c, RT, DM, true, CLAMP>(g, k, lm, i0, j0, je, MaskPlain{je - j0}, Nn, Fc, GX, GY, AdvRows<RT>{AX, AY}, out, gxo, gyo, nullptr);
}

// ... of a tile that is not all fluid: k234_bnd_phase1 / 2 without the exchange (masks, K12 of the reference on the not-wall cells of the window, what
// the buffer holds on wall cells, edge-row slots; the not-wall cells that are not fluid are stored)
template <int c, int RT, int DM, bool CLAMP>
__device__ __forceinline__ void k234_dye_bnd(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm, int i0, int j0, int je,
                                             float *out, float *gxo, float *gyo, float *fn, const float *fc, const float *gxc, const float *gyc, const float *v)
{
    using T = float;
    constexpr int N = 2;
    FS_PIN_LANE_OFFSET(i0);
    unsigned nwbits = 0u, flbits = 0u;      // (as k234_bnd_phase1)
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 2 + u));
        nwbits |= lv_sel_nw<N>(m) << (2 * u);
        if (u >= 2 && u < RT + 2 && j0 + u - 2 < je) flbits |= lv_sel_fluid<N>(m) << (2 * (u - 2));
    }
#define FS_NWK(u) ((nwbits >> (2 * (u))) & 3u)
    v2f F[RT + 6], Nn[RT + 4], Fc[RT + 4], GX[RT + 2], GY[RT + 2], AX[RT + 2], AY[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 6; ++u) F[u] = pk(lv_field<3, T, N>(fc, g, c, i0, clampy(g, j0 - 3 + u)));
#pragma unroll
    for (int s = 0; s < RT + 2; ++s) {
        const int row = clampy(g, j0 - 1 + s);
        GX[s] = pk(lv_field<3, T, N>(gxc, g, c, i0, row));
        GY[s] = pk(lv_field<3, T, N>(gyc, g, c, i0, row));
        AX[s] = pk(lv_field<2, T, N>(v, g, 0, i0, row));
        AY[s] = pk(lv_field<2, T, N>(v, g, 1, i0, row));
    }
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        Nn[u] = v2f{0.0f, 0.0f};
        if (FS_NWK(u) != 3u) Nn[u] = pk(lv_field<3, T, N>(fn, g, c, i0, clampy(g, j0 - 2 + u)));
    }
#pragma unroll
    for (int u = 0; u < RT + 4; ++u) {
        Nn[u] = sel2(FS_NWK(u), nonadv_dye_pk_row<DM>(k, lm, F[u], F[u + 1], F[u + 2]), Nn[u]);
        Fc[u] = F[u + 1];
    }
#pragma unroll
    for (int u = RT + 2; u >= 0; --u) if (j0 - 2 + u < g.jlo) Nn[u] = Nn[u + 1];
#pragma unroll
    for (int u = 1; u < RT + 4; ++u) if (j0 - 2 + u > g.jhi) Nn[u] = Nn[u - 1];
    if (lm.owner) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const unsigned sel = FS_NWK(t + 2) & ~(flbits >> (2 * t)) & 3u;
            if (j0 + t < je && sel) lv_store_row_sel<3, T, N>(fn, g, c, i0, j0 + t, unpk(Nn[t + 2]), sel);
        }
    }
    cip_k34_pk_core<3, c, RT, DM, false, CLAMP>(g, k, lm, i0, j0, je, MaskPacked{nwbits >> 2, flbits}, Nn, Fc, GX, GY, AdvRows<RT>{AX, AY}, out, gxo, gyo, nullptr, gxc, gyc);
#undef FS_NWK
}

// one workgroup = one wave = one channel of one listed tile.  KIND: 1 - the all-fluid tiles (class 1 list), 2 - the others (class 2 list), 0 - both in ONE
// launch (class 0 list: the entry's hint bit says which body; as k_cip_step_all)
template <int RT, int DM, bool CLAMP, int KIND>
__global__ __launch_bounds__(64, 4) void k_cip_dye(Grid g, Konst<float> k, int nbx, int nby, int jb, int je,
                                                   float *out, float *gxo, float *gyo, float *fn, const float *fc, const float *gxc, const float *gyc, const float *v)
{
    constexpr int N = 2, HL = 2, OW = 64 - 2 * HL;
    int wx, ty, cg;
    unsigned cls = 0u;
    if (!band_coords<3>(g, nbx, nby, wx, ty, cg, 0, KIND == 0 ? &cls : nullptr)) return;
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const LaneMapN<N> lm_in = lane_map_n<N, HL>(g, wx);
    const int i0 = lm_in.i0, j0 = jb + ty * RT;
    // The three channels run ONE body (round 6): channel cg of a 3-channel field at (row, i) is element ((row 3 + cg) P + i) - the bodies are instantiated for
    // channel 0 and handed the planes of channel cg (every pointer moved by cg P; workgroup-uniform).  A third of the code: the kernel was 74 KB (f64-multiply
    // divisions) to 139 KB (IEEE) against an instruction cache of 64 KB per two CUs.
    const size_t co = (size_t)cg * (size_t)g.P;
    if (KIND == 1 || (KIND == 0 && (cls & 1u))) {
        const LaneMapN<N> lm{lm_in.i0, lm_in.owner, false, false};
        k234_dye_plain<0, RT, DM, CLAMP>(g, k, lm, i0, j0, je, out + co, gxo + co, gyo + co, fc + co, gxc + co, gyc + co, v);
    } else {
        k234_dye_bnd<0, RT, DM, CLAMP>(g, k, lm_in, i0, j0, je, out + co, gxo + co, gyo + co, fn + co, fc + co, gxc + co, gyc + co, v);
    }
}

}  // namespace fs
