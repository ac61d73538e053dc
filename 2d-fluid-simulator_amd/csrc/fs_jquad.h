// fs_jquad.h - FOUR Jacobi sweeps, and the pressure boundary passes in front of each, in one pass over the grid.
//
// Reference (fs/pressure_updater.py:56-66): n x { K7(p.current); p.next[not wall] = predict_p(p.current); swap }.  Long runs (BASELINE
// configs[1]: 50 sweeps per step on a 3200 x 1600 grid whose 80 MB never leave the Infinity Cache) are bound by launches and
// latency, not by HBM: round 2's two-sweep pass (k_jacobi_pair) takes 21.6 us per pass = 66 % of that step.  Here a lane holds rows
// j0-4 .. j0+RT+3 of the iterate and rows j0-3 .. j0+RT+2 of the precomputed source pair in registers and runs four sweeps on
// shrinking row ranges (x-neighbours through the halo lanes, whose 4 cells are exactly the reach of four radius-1 stages), with K7
// evaluated as a view in front of every sweep from the recipe bytes (fs_march.h lazy_value; lv_bc_row in fs_rbpair.h).
//
// The buffers hold RAW sweep output during such a run (the host issues the last two sweeps with the real boundary kernel, which leaves
// both physical buffers as the reference does).  A pass writes the not-wall cells of `pn` and nothing else: wall cells with a recipe are
// recomputed by whoever reads them, wall cells without one hold the same value in both buffers - the host checks that (Field.static_id:
// equal unless somebody uploaded into one of them) and otherwise keeps the two-sweep passes, which tell the two buffers' histories apart.
//
// Validity (host-checked per mask, fs_api.hip build_bc_ops -> jq_ok): no recipe reads a source on the far side of its target as seen
// from a cell whose raw value is live (a fluid cell, or the source of some recipe) - then the plain stencil's footprint suffices, as in
// fs_rbpair.h; and the first / last domain row hold no not-wall cell.
#pragma once
#include "fs_rbpair.h"

namespace fs {

template <int N> __device__ __forceinline__ unsigned lv_sel_not_wall(uint32_t m)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((m >> (8 * c)) & 0xffu) != 1u ? (1u << c) : 0u;
    return s;
}

// one sweep of one row from finished rows: out[c] = computed ? predict_p(neighbours) : ctr[c]
template <typename T, int N>
__device__ __forceinline__ LV<T, N> jq_row(const LaneMapN<N> &lm, unsigned computed, const LV<T, N> &m, const LV<T, N> &ctr, const LV<T, N> &p,
                                           const LV<T, N> &s2, const LV<T, N> &s3)
{
    const T pl = lv_left<T, N>(lm, ctr), pr = lv_right<T, N>(lm, ctr);
    if constexpr (N == 2 && sizeof(T) == 4) {
        // the lane's two cells as one packed operand (fs_device.h v2f): predict_p's five additions and its multiplication pair up; pE + pW of both
        // cells is ONE v_pk_add_f32 on (c.x, c.y) and (left, right) (ew_sum; the second cell adds in the other order - the same bits)
        const v2f c2 = pk(ctr);
        const v2f val = (0.25f * ((ew_sum(c2, pl, pr) + pk(p)) + pk(m)) + pk(s2)) - pk(s3);
        return unpk(sel2(computed, val, c2));
    } else {
        LV<T, N> o;
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const T pE = c == N - 1 ? pr : ctr.a[c == N - 1 ? c : c + 1], pW = c == 0 ? pl : ctr.a[c == 0 ? 0 : c - 1];
            const T val = predict_from(pE, pW, p.a[c], m.a[c], s2.a[c], s3.a[c]);
            o.a[c] = (computed & (1u << c)) ? val : ctr.a[c];
        }
        return o;
    }
}

template <int N, int RT, bool BND, typename T>
__device__ __forceinline__ void jacobi_quad_tile(const Grid &g, const LaneMapN<N> &lm_in, int i0, int j0, int je, const unsigned (&nw)[RT + 8],
                                                 const uint8_t *bcmap, T *pn, const T *pc, const T *src)
{
    constexpr int W = RT + 8;                  // window rows w = 0 .. W-1  <->  local rows j0-4 .. j0+RT+3 (clamped into the domain)
    using R = LV<T, N>;
    const LaneMapN<N> lm = BND ? lm_in : LaneMapN<N>{lm_in.i0, lm_in.owner, false, false};      // (no lane of an all-fluid tile sits at a domain end: fs_rbpair.h rbsor_pair_tile)
    constexpr unsigned ALL = (1u << N) - 1u;
#define FS_NW(w) (BND ? nw[w] : ALL)
    R P[W], S2[W], S3[W];
    uint32_t code[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int j = clampy(g, j0 - 4 + w);
        P[w] = lv_field<1, T, N>(pc, g, 0, i0, j);
        if (w >= 1 && w <= W - 2) {
            S2[w] = lv_field<2, T, N>(src, g, 0, i0, j);
            S3[w] = lv_field<2, T, N>(src, g, 1, i0, j);
        }
        code[w] = BND ? lv_bytes<N>(bcmap, g, i0, j) : 0u;
    }
    // sweep s = 1 .. 4 on rows s .. W-1-s from the view of the rows s-1 .. W-s of the previous iterate (a row missing at the edge of the
    // window is stood in for by the row itself: what that produces is only read where the validity argument of the header excludes it)
#pragma unroll
    for (int s = 1; s <= 4; ++s) {
        R V[W];
#pragma unroll
        for (int w = s - 1; w <= W - s; ++w)
            V[w] = BND ? lv_bc_row<T, N>(lm, P[w == s - 1 ? w : w - 1], P[w], P[w == W - s ? w : w + 1], code[w]) : P[w];
#pragma unroll
        for (int w = s; w <= W - 1 - s; ++w) P[w] = jq_row<T, N>(lm, FS_NW(w), V[w - 1], V[w], V[w + 1], S2[w], S3[w]);
    }
#pragma unroll
    for (int w = 4; w <= W - 5; ++w) {
        const int j = j0 - 4 + w;
        if (j >= je) break;
        const unsigned sel = FS_NW(w);
        if (lm.owner && sel) lv_store_sel<T, N>(pn + idx<1, T>(g, 0, i0, j), P[w], sel);
    }
#undef FS_NW
}

// ------------------------------------------------------------------------------------------------
// K8J, the literal sweep (reads p and v like fs/pressure_updater.py:62-66) on lanes of 2 cells, packed (round 5): the source term's four
// differences, its products and sums and predict_p's additions pair up (fs_device.h v2f; sxx / sxy and pE + pW as ONE packed add each,
// ew_diff / ew_sum); the division by 8 dt stays per half.  Tiles of RT rows, one halo lane per side (62 owner lanes = 124 cells).  Same
// expression per cell as k_jacobi_ov - same bits; the quad form holds 4 cells per lane, whose inner x-neighbours are an unaligned register
// pair and do not pack.  f32 only.
// ------------------------------------------------------------------------------------------------
template <int RT, int DM, bool PLAIN>
__device__ __forceinline__ void jacobi_ov2_tile(const Grid &g, const Konst<float> &k, const LaneMapN<2> &lm_in, int i0, int j0, int je, const unsigned (&nw)[RT],
                                                float *pn, const float *pc, const float *vc)
{
    using T = float;
    constexpr int N = 2;
    const LaneMapN<N> lm = PLAIN ? LaneMapN<N>{lm_in.i0, lm_in.owner, false, false} : lm_in;      // (a plain tile holds no lane at the domain's first / last column)
    v2f P[RT + 2], VX[RT + 2], VY[RT + 2];
#pragma unroll
    for (int u = 0; u < RT + 2; ++u) {
        const int row = clampy(g, j0 - 1 + u);
        P[u] = pk(lv_field<1, T, N>(pc, g, 0, i0, row));
        VX[u] = pk(lv_field<2, T, N>(vc, g, 0, i0, row));
        VY[u] = pk(lv_field<2, T, N>(vc, g, 1, i0, row));
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int j = j0 + t;
        if (j >= je) break;
        const v2f xc = VX[t + 1], yc = VY[t + 1], pcur = P[t + 1];
        const T xl = lv_left<T, N>(lm, unpk(xc)), xr = lv_right<T, N>(lm, unpk(xc));
        const T yl = lv_left<T, N>(lm, unpk(yc)), yr = lv_right<T, N>(lm, unpk(yc));
        const T pl = lv_left<T, N>(lm, unpk(pcur)), pr = lv_right<T, N>(lm, unpk(pcur));
        // fs_march.h source_from: s2 = ((sxx sxx + syy syy) + syx sxy) / 8,  s3 = (dx (sxx + syy)) / (8 dt)
        const v2f sxx = ew_diff(xc, xl, xr), sxy = ew_diff(yc, yl, yr), syx = VX[t + 2] - VX[t], syy = VY[t + 2] - VY[t];
        const v2f s2 = ((sxx * sxx + syy * syy) + (syx * sxy)) / 8.0f;
        const v2f s3n = k.dx * (sxx + syy);
        v2f s3;
        s3.x = cdiv<DM>(s3n.x, k.eight_dt, k.r_eight_dt);
        s3.y = cdiv<DM>(s3n.y, k.eight_dt, k.r_eight_dt);
        // predict_from: (0.25 (((pE + pW) + pN) + pS) + s2) - s3
        const v2f val = (0.25f * ((ew_sum(pcur, pl, pr) + P[t + 2]) + P[t]) + s2) - s3;
        if (lm.owner) {
            if (PLAIN) lv_store_row<1, T, N>(pn, g, 0, i0, j, unpk(val));
            else if (nw[t]) lv_store_sel<T, N>(pn + idx<1, T>(g, 0, i0, j), unpk(val), nw[t]);
        }
    }
}

// the list's per-wave hint (fs_core.hip tile_list, reach 1): nothing but fluid in and around the wave's tile - no mask loads, whole-lane stores
template <int RT, int DM>
__global__ __launch_bounds__(256) void k_jacobi_ov2(Grid g, Konst<float> k, int nbx, int nby, int jb, int je, float *pn, const float *pc, const float *vc)
{
    constexpr int N = 2, HL = 1, OW = 64 - 2 * HL;
    int bx, by, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby, bx, by, cg, 0, &cls)) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = blockDim.x >> 6;
    const int wx = (nby & FS_STACKED) ? bx : bx * nwv + w, ty = (nby & FS_STACKED) ? by * nwv + w : by;
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const LaneMapN<N> lm = lane_map_n<N, HL>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned nw[RT];
    if ((cls >> w) & 1u) {
#pragma unroll
        for (int t = 0; t < RT; ++t) nw[t] = 3u;
        jacobi_ov2_tile<RT, DM, true>(g, k, lm, i0, j0, je, nw, pn, pc, vc);
        return;
    }
    bool any = false;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        nw[t] = j0 + t < je ? lv_sel_not_wall<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 + t))) : 0u;
        any = any || (lm.owner && nw[t] != 0u);
    }
    if (!__any(any)) return;
    jacobi_ov2_tile<RT, DM, false>(g, k, lm, i0, j0, je, nw, pn, pc, vc);
}

// PATH 2: classify the tile here (mask loads); 3: the host listed this workgroup as plain - nothing but fluid within reach (fs_api.hip
// tile_list): the plain path without looking, as its own kernel with its own (small) register budget
template <int N, int RT, int PATH, typename T>
__global__ __launch_bounds__(256) void k_jacobi_quad(Grid g, int nbx, int nby, int jb, int je, const uint8_t *bcmap, T *pn, const T *pc, const T *src)
{
    constexpr int W = RT + 8;
    int wx, ty;
    bool hint = false;
    if (!tile_coords_n<N>(g, nbx, nby, jb, je, RT, wx, ty, PATH == 2 ? &hint : nullptr)) return;
    const LaneMapN<N> lm = lane_map_n<N>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned nw[W];
    bool plain = PATH == 3 || hint;            // (hint: this wave's bit of the list entry, fs_march.h band_coords)
    if (!plain) {
        bool own = false, all_fluid = true;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 4 + w));
            nw[w] = lv_sel_not_wall<N>(m);
            all_fluid = all_fluid && m == 0u;
            if (w >= 4 && w <= W - 5 && j0 - 4 + w < je) own = own || (lm.owner && nw[w] != 0u);
        }
        if (!__any(own)) return;               // no not-wall cell in the rows this tile stores
        plain = __all(all_fluid);
    }
    if (plain) jacobi_quad_tile<N, RT, false, T>(g, lm, i0, j0, je, nw, bcmap, pn, pc, src);
    else if constexpr (PATH != 3) jacobi_quad_tile<N, RT, true, T>(g, lm, i0, j0, je, nw, bcmap, pn, pc, src);
}

// ------------------------------------------------------------------------------------------------
// The LAST two rounds of a lazily-bounded run in one pass.  The reference ends with
//     K7(A = p.current: iterate n-2);  B[not wall] = sweep(A);  swap;     K7(B: iterate n-1);  A[not wall] = sweep(B);  swap
// which leaves   p.current = A: iterate n on the not-wall cells, K7(iterate n-2) on the wall cells with a recipe,
//                p.next    = B: iterate n-1 with K7 applied (wall targets, inflow = its right neighbour, outflow = 0).
// Run as launches that is 2 x (boundary kernel + sweep) - 47 of the 585 us of a BASELINE configs[1] step.  Here a lane reads rows j0-2 ..
// j0+RT+1 of the raw iterate n-2, forms view -> sweep -> view -> sweep in registers and stores both buffers' final content: B in place (this
// pass does not read it), A's content into a THIRD buffer C (other tiles still read A's rows in their halo; the caller rotates
// p.current <- C, spare <- A).  Every cell some kernel writes (not-wall cells and recipe targets) is stored, the others are equal in
// all three buffers (Field.static_id, checked by the caller).  Same validity conditions as the four-sweep pass.
// ------------------------------------------------------------------------------------------------
template <int N, int RT, bool BND, typename T>
__device__ __forceinline__ void jacobi_finish_tile(const Grid &g, const LaneMapN<N> &lm_in, int i0, int j0, int je, const unsigned (&nw)[RT + 4],
                                                   const uint8_t *bcmap, T *pc_out, T *pn, const T *pc, const T *src)
{
    constexpr int W = RT + 4;                  // window rows w = 0 .. W-1  <->  local rows j0-2 .. j0+RT+1 (clamped into the domain)
    using R = LV<T, N>;
    const LaneMapN<N> lm = BND ? lm_in : LaneMapN<N>{lm_in.i0, lm_in.owner, false, false};
    constexpr unsigned ALL = (1u << N) - 1u;
#define FS_NW(w) (BND ? nw[w] : ALL)
    R P[W], S2[W], S3[W];
    uint32_t code[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int j = clampy(g, j0 - 2 + w);
        P[w] = lv_field<1, T, N>(pc, g, 0, i0, j);
        if (w >= 1 && w <= W - 2) {
            S2[w] = lv_field<2, T, N>(src, g, 0, i0, j);
            S3[w] = lv_field<2, T, N>(src, g, 1, i0, j);
        }
        code[w] = BND ? lv_bytes<N>(bcmap, g, i0, j) : 0u;
    }
    // view of iterate n-2 (what K7 leaves in A), sweep -> iterate n-1 on rows 1 .. W-2
    R V0[W], P1[W];
#pragma unroll
    for (int w = 0; w < W; ++w) V0[w] = BND ? lv_bc_row<T, N>(lm, P[w == 0 ? w : w - 1], P[w], P[w == W - 1 ? w : w + 1], code[w]) : P[w];
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) P1[w] = jq_row<T, N>(lm, FS_NW(w), V0[w - 1], V0[w], V0[w + 1], S2[w], S3[w]);
    // view of iterate n-1 (B's final content), sweep -> iterate n on rows 2 .. W-3
    R V1[W];
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) V1[w] = BND ? lv_bc_row<T, N>(lm, P1[w == 1 ? w : w - 1], P1[w], P1[w == W - 2 ? w : w + 1], code[w]) : P1[w];
#pragma unroll
    for (int w = 2; w <= W - 3; ++w) {
        const int j = j0 - 2 + w;
        if (j >= je) break;
        const R p2 = jq_row<T, N>(lm, FS_NW(w), V1[w - 1], V1[w], V1[w + 1], S2[w], S3[w]);
        const unsigned nwb = FS_NW(w), sel = nwb | (BND ? lv_sel_target<N>(code[w]) : 0u);      // not-wall cells and recipe targets
        if (lm.owner && sel) {
            R c;                                                                                // A's final content: iterate n, K7(iterate n-2) on walls
#pragma unroll
            for (int q = 0; q < N; ++q) c.a[q] = (nwb >> q) & 1u ? p2.a[q] : V0[w].a[q];
            lv_store_sel<T, N>(pc_out + idx<1, T>(g, 0, i0, j), c, sel);
            lv_store_sel<T, N>(pn + idx<1, T>(g, 0, i0, j), V1[w], sel);
        }
    }
#undef FS_NW
}

template <int N, int RT, typename T>
__global__ __launch_bounds__(256) void k_jacobi_finish(Grid g, int nbx, int nby, int jb, int je, const uint8_t *bcmap, T *pc_out, T *pn, const T *pc, const T *src)
{
    constexpr int W = RT + 4;
    int wx, ty;
    bool hint = false;
    if (!tile_coords_n<N>(g, nbx, nby, jb, je, RT, wx, ty, &hint)) return;
    const LaneMapN<N> lm = lane_map_n<N>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned nw[W];
    bool plain = hint;
    if (!plain) {
        bool near = false, all_fluid = true;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint32_t m = lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 2 + w));
            nw[w] = lv_sel_not_wall<N>(m);
            all_fluid = all_fluid && m == 0u;
            if (w >= 1 && w <= W - 2) near = near || nw[w] != 0u;
        }
        // a cell this tile stores is a not-wall cell or a recipe target, i.e. a wall cell next to a not-wall cell: none without a not-wall cell
        // in the rows around the tile's (halo lanes included)
        if (!__any(near)) return;
        plain = __all(all_fluid);
    }
    if (plain) jacobi_finish_tile<N, RT, false, T>(g, lm, i0, j0, je, nw, bcmap, pc_out, pn, pc, src);
    else jacobi_finish_tile<N, RT, true, T>(g, lm, i0, j0, je, nw, bcmap, pc_out, pn, pc, src);
}

}  // namespace fs
