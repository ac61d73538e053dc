// fs_rbpair.h - TWO red-black SOR iterations, and both pressure boundary passes between them, in one pass over HBM.
//
// Reference (fs/pressure_updater.py:86-114, n_iter = 2 is what fs/fluid_simulator.py:76-78 wires into every create()):
//     K7(A)                                     A = p.current, B = p.next
//     odd : B[odd fluid]  = (1-w) A + w predict_p(A)            even: B[even fluid] = (1-w) B + w predict_p(B)      swap
//     K7(B)
//     odd : A[odd fluid]  = (1-w) B + w predict_p(B)            even: A[even fluid] = (1-w) A + w predict_p(A)      swap
// i.e. four dependent half sweeps of radius 1.  As four launches (two fused iterations + two boundary kernels) that is 2 x 21 B/cell
// and 30 % of the headline step.  Here a lane holds rows j0-4 .. j0+RT+3 of its cells in registers, runs the four half sweeps on
// shrinking row ranges (x-neighbours through the halo lanes, whose 4 cells are exactly the reach of four radius-1 stages) and stores
// RT rows of both results: mask 1 + A 4 + B 4 + v 8 read, 8 written = 25 B per fluid cell for TWO iterations.  The Poisson source
// (s2, s3 of predict_p) depends on v only: it is evaluated once per cell and serves both iterations.
//
// Lane width N (cells per lane): the window is RT + 8 rows of four planes - with quads (N = 4, one 16-byte load per row and plane, like
// every other tile kernel) that is 240 VGPRs = 2 waves per SIMD, and a wave that computes cannot hide behind one that loads: measured
// 252 us against 2 x (8.7 + 124) for the launches it replaces.  N = 2 (8-byte loads, 60 owner lanes + 2 x 2 halo lanes) halves the
// registers per row.
//
// Out of place: other tiles read A and B in their halo rows while this one produces its rows, so the results go to a SECOND pair of
// pressure buffers (C <- A after iteration 2, D <- B after iteration 1 + K7); the caller rotates (p.current, p.next) <-> spare pair.
//
// K7 (fs/boundary_condition.py:41-65) is evaluated where it is consumed, from the recipe bytes built at mask upload (fs_march.h
// lazy_value: a boundary cell takes the value, or the mean of two values, of 4-neighbours; outflow 0; inflow its right neighbour):
// "view(P)" below is buffer state P as K7 would leave it.  Cells no kernel ever writes (wall cells without a recipe) hold the same
// value in all four buffers (zero, or - after an upload - whatever one FULL pass carries over), so only fluid cells and K7 targets
// are stored.
//
// Validity of the shrinking-window argument (host-checked per mask, fs_api.hip build_bc_ops -> rb_pair_ok): a recipe never reads a
// source that lies on the far side of its target as seen from a fluid reader (a wall one cell thick between two fluid regions) -
// then whoever reads a boundary value finds the recipe's sources inside its own radius-1 neighbourhood, and the plain stencil's
// footprint suffices; and the first / last domain row hold no fluid cell (no clamped y neighbour of a relaxed cell).  All of the
// reference's scenes qualify at their usual resolutions; masks that do not keep the two-launch iterations.
#pragma once
#include "fs_march.h"

namespace fs {

#ifndef FS_RBP_MIRROR
#define FS_RBP_MIRROR 1    // every second row of plain tiles of the two-part launch works bottom-up (rbsor_pair_tile MIRROR; 0: A/B)
#endif

// N consecutive cells of one row
template <typename T, int N> struct LV { T a[N]; };
template <typename T, int N> struct LVec;
template <> struct LVec<float, 4> { using type = float4; };
template <> struct LVec<float, 2> { using type = float2; };
template <> struct LVec<double, 2> { using type = double2; };
template <int N> struct LMaskWord;
template <> struct LMaskWord<4> { using type = uint32_t; };
template <> struct LMaskWord<2> { using type = uint16_t; };

template <typename T, int N>
__device__ __forceinline__ LV<T, N> lv_load(const T *row, int i0)
{
    const typename LVec<T, N>::type q = load_row_quad<typename LVec<T, N>::type, T>(row, i0);
    LV<T, N> r;
    if constexpr (N == 4) { r.a[0] = q.x; r.a[1] = q.y; r.a[2] = q.z; r.a[3] = q.w; }
    else { r.a[0] = q.x; r.a[1] = q.y; }
    return r;
}
template <int C, typename T, int N>
__device__ __forceinline__ LV<T, N> lv_field(const T *f, const Grid &g, int c, int i0, int j)
{ return lv_load<T, N>(f + ((size_t)j * C + c) * g.P, i0); }
template <int N>
__device__ __forceinline__ uint32_t lv_bytes(const uint8_t *plane, const Grid &g, int i0, int j)        // N mask / recipe bytes, byte k = cell k
{ return (uint32_t)load_row_quad<typename LMaskWord<N>::type, uint8_t>(plane + (size_t)j * g.Pm, i0); }

template <int N> __device__ __forceinline__ unsigned lv_sel_fluid(uint32_t m)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((m >> (8 * c)) & 0xffu) == 0u ? (1u << c) : 0u;
    return s;
}
template <int N> __device__ __forceinline__ unsigned lv_sel_target(uint32_t code)
{
    unsigned s = 0u;
#pragma unroll
    for (int c = 0; c < N; ++c) s |= ((code >> (8 * c)) & 1u) << c;
    return s;
}
template <typename T, int N>
__device__ __forceinline__ void lv_store_sel(T *dst, const LV<T, N> &v, unsigned sel)
{
    if (sel == (1u << N) - 1u) {
        typename LVec<T, N>::type q;
        if constexpr (N == 4) { q.x = v.a[0]; q.y = v.a[1]; q.z = v.a[2]; q.w = v.a[3]; }
        else { q.x = v.a[0]; q.y = v.a[1]; }
        *reinterpret_cast<typename LVec<T, N>::type *>(dst) = q;
        return;
    }
#pragma unroll
    for (int c = 0; c < N; ++c)
        if (sel & (1u << c)) dst[c] = v.a[c];
}

// the whole lane into row j, channel c of a C-channel field: wave-uniform row base + the lane offset (the addressing of lv_field; rows must be wave-uniform)
template <int C, typename T, int N>
__device__ __forceinline__ void lv_store_row(T *f, const Grid &g, int c, int i0, int j, const LV<T, N> &v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using Q = typename LVec<T, N>::type;
    Q q;
    if constexpr (N == 4) { q.x = v.a[0]; q.y = v.a[1]; q.z = v.a[2]; q.w = v.a[3]; }
    else { q.x = v.a[0]; q.y = v.a[1]; }
    // (the empty asm keeps the zero-extension of the lane offset in the basic block of the store - usually the body of `if (owner)`: hipcc selects
    //  the scalar-base form only when it sees the extension there; hoisted out, every store pays a 64-bit VALU add and a register pair)
    asm volatile("" : "+v"(i0));
    *reinterpret_cast<__attribute__((address_space(1))) Q *>((fs_gptr)uniform64((uint64_t)(f + ((size_t)j * C + c) * g.P)) + (unsigned)i0 * (unsigned)sizeof(T)) = q;
#else
    lv_store_sel<T, N>(f + ((size_t)j * C + c) * g.P + i0, v, (1u << N) - 1u);
#endif
}

// ... and the cells of the lane that `sel` names (bit k = cell k): the same addressing - no 64-bit address pair per store (round 5: the general bodies'
// selective stores went through `dst + idx(...)`, a v_lshl_add_u64 and two VGPRs each)
template <int C, typename T, int N>
__device__ __forceinline__ void lv_store_row_sel(T *f, const Grid &g, int c, int i0, int j, const LV<T, N> &v, unsigned sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using Q = typename LVec<T, N>::type;
    asm volatile("" : "+v"(i0));
    const fs_gptr row = (fs_gptr)uniform64((uint64_t)(f + ((size_t)j * C + c) * g.P));
    const unsigned off = (unsigned)i0 * (unsigned)sizeof(T);
    if (sel == (1u << N) - 1u) {
        Q q;
        if constexpr (N == 4) { q.x = v.a[0]; q.y = v.a[1]; q.z = v.a[2]; q.w = v.a[3]; }
        else { q.x = v.a[0]; q.y = v.a[1]; }
        *reinterpret_cast<__attribute__((address_space(1))) Q *>(row + off) = q;
        return;
    }
#pragma unroll
    for (int q = 0; q < N; ++q)
        if (sel & (1u << q)) *reinterpret_cast<__attribute__((address_space(1))) T *>(row + off + (unsigned)(q * sizeof(T))) = v.a[q];
#else
    lv_store_sel<T, N>(f + ((size_t)j * C + c) * g.P + i0, v, sel);
#endif
}

// ---- the two cells of a lane as one packed operand (fs_device.h v2f) ----------------------------------------------------------------------
__device__ __forceinline__ v2f pk(const LV<float, 2> &r) { v2f v; v.x = r.a[0]; v.y = r.a[1]; return v; }
__device__ __forceinline__ LV<float, 2> unpk(v2f v) { LV<float, 2> r; r.a[0] = v.x; r.a[1] = v.y; return r; }
__device__ __forceinline__ v2f east(v2f c, float r) { v2f v; v.x = c.y; v.y = r; return v; }      // the cells right of the lane's two
__device__ __forceinline__ v2f west(float l, v2f c) { v2f v; v.x = l; v.y = c.x; return v; }      // ... left of them
__device__ __forceinline__ v2f sel2(unsigned bits, v2f a, v2f b) { v2f r; r.x = (bits & 1u) ? a.x : b.x; r.y = (bits & 2u) ? a.y : b.y; return r; }
// east(c, r) - west(l, c) = (c.y - l, r - c.x) as ONE packed add on the register pairs (c.x, c.y) and (l, r): the halves are picked by op_sel
// and negated by neg_lo / neg_hi ((-c.x) + r has the bits of r - c.x: IEEE addition commutes).  Building the two shifted pairs costs a
// v_mov each - hipcc does not fold a VGPR swizzle into the modifiers.
__device__ __forceinline__ v2f ew_diff(v2f c, float l, float r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v2f z, o;
    z.x = l; z.y = r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(o) : "v"(c), "v"(z));
    return o;
#else
    return east(c, r) - west(l, c);
#endif
}

// east(c, r) + west(l, c) = (c.y + l, c.x + r) the same way (the second half adds in the other order: IEEE addition commutes)
__device__ __forceinline__ v2f ew_sum(v2f c, float l, float r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    v2f z, o;
    z.x = l; z.y = r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(o) : "v"(c), "v"(z));
    return o;
#else
    return east(c, r) + west(l, c);
#endif
}

// overlapped-wave column mapping for lanes of N cells: HL halo lanes on each side (default 4 / N = 4 cells: the reach of four radius-1 stages)
template <int N> struct LaneMapN { int i0; bool owner, at_lo, at_hi; };
template <int N, int HL = 4 / N>
__device__ __forceinline__ LaneMapN<N> lane_map_n(const Grid &g, int wave)
{
    constexpr int OW = 64 - 2 * HL;
    const int lane = threadIdx.x & 63, nu = g.X / N;
    int q = wave * OW - HL + lane;
    LaneMapN<N> m;
    m.owner = lane >= HL && lane < 64 - HL && q >= 0 && q < nu;
    q = q < 0 ? 0 : (q > nu - 1 ? nu - 1 : q);
    m.i0 = q * N;
    m.at_lo = q == 0;
    m.at_hi = q == nu - 1;
    return m;
}
template <typename T, int N> __device__ __forceinline__ T lv_left(const LaneMapN<N> &m, const LV<T, N> &v)
{ const T l = lane_prev(v.a[N - 1]); return m.at_lo ? v.a[0] : l; }
template <typename T, int N> __device__ __forceinline__ T lv_right(const LaneMapN<N> &m, const LV<T, N> &v)
{ const T r = lane_next(v.a[0]); return m.at_hi ? v.a[N - 1] : r; }

// plain_hint (optional): the host's list says this wave sees nothing but fluid within the kernel's reach (band_coords cls)
// bnd_fluid (optional; one-wave workgroups of a boundary list): the host saw a fluid cell in the tile's own rows (bit 1 of the entry's hint)
template <int N>
__device__ __forceinline__ bool tile_coords_n(const Grid &g, int nbx, int nby_packed, int jb, int je, int rt, int &wave_x, int &tile_y, bool *plain_hint = nullptr, bool *bnd_fluid = nullptr)
{
    constexpr int OW = 64 - 2 * (4 / N);
    int bx, by, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby_packed, bx, by, cg, 0, plain_hint ? &cls : nullptr)) return false;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    if (plain_hint) *plain_hint = ((cls >> w) & 1u) != 0u;
    if (bnd_fluid) *bnd_fluid = nw == 1 && (cls & 2u) != 0u;
    if (nby_packed & FS_STACKED) { wave_x = bx; tile_y = by * nw + w; }
    else { wave_x = bx * nw + w; tile_y = by; }
    return wave_x * OW < g.X / N && jb + tile_y * rt < je;
}

// the same for kernels with ZG channel groups and HL halo lanes per side (fs_k34n.h, fs_k234.h, fs_jquad.h k_jacobi_ov2)
template <int N, int ZG, int HL>
__device__ __forceinline__ bool tile_coords_nz(const Grid &g, int nbx, int nby_packed, int jb, int je, int rt, int &wave_x, int &tile_y, int &cg, unsigned *cls = nullptr)
{
    constexpr int OW = 64 - 2 * HL;
    int bx, by;
    if (!band_coords<ZG>(g, nbx, nby_packed, bx, by, cg, 0, cls)) return false;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    if (nby_packed & FS_STACKED) { wave_x = bx; tile_y = by * nw + w; }
    else { wave_x = bx * nw + w; tile_y = by; }
    return wave_x * OW < g.X / N && jb + tile_y * rt < je;
}

// ... and with the launch list's per-wave hint: `plain` = the host found nothing but fluid within the kernel's reach of this wave's tile (fs_core.hip
// tile_list, lists built with a reach), false on dense launches.  A kernel that takes it skips its mask loads - and with them the round trip in
// front of its window loads (round 5: the literal Jacobi sweep 81.6 -> 75 us, vorticity confinement 98 -> 81 us).
template <int N, int HL>
__device__ __forceinline__ bool tile_coords_hint(const Grid &g, int nbx, int nby_packed, int jb, int je, int rt, int &wave_x, int &tile_y, bool &plain)
{
    constexpr int OW = 64 - 2 * HL;
    int bx, by, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby_packed, bx, by, cg, 0, &cls)) return false;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    if (nby_packed & FS_STACKED) { wave_x = bx; tile_y = by * nw + w; }
    else { wave_x = bx * nw + w; tile_y = by; }
    plain = ((cls >> w) & 1u) != 0u;
    return wave_x * OW < g.X / N && jb + tile_y * rt < je;
}

// one row of the buffer as K7 would leave it, from raw rows m / c / n = rows j-1 / j / j+1 and the recipe bytes of row j (whole wave)
template <typename T, int N>
__device__ __forceinline__ LV<T, N> lv_bc_row(const LaneMapN<N> &lm, const LV<T, N> &m, const LV<T, N> &c, const LV<T, N> &n, uint32_t code)
{
    if (!__any(lv_sel_target<N>(code) != 0u)) return c;            // wave-uniform: no target in this row of the wave
    const T cl = lv_left<T, N>(lm, c), cr = lv_right<T, N>(lm, c);
    LV<T, N> o;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const T sL = q == 0 ? cl : c.a[q == 0 ? 0 : q - 1], sR = q == N - 1 ? cr : c.a[q == N - 1 ? q : q + 1];
        o.a[q] = lazy_value((code >> (8 * q)) & 0xffu, c.a[q], sL, sR, m.a[q], n.a[q]);
    }
    return o;
}

// one colour of one row from finished rows: cells c with ((c + PAR) & 1) == COLOR and a fluid bit are relaxed
//   out[c] = (1-w) ctr[c] + w predict_p(neighbours of ctr; rows below / above: m / p)
template <int PAR, int COLOR, typename T, int N>
__device__ __forceinline__ void rbp_relax(const Konst<T> &k, const LaneMapN<N> &lm, unsigned fluid, const LV<T, N> &m, const LV<T, N> &ctr, const LV<T, N> &p,
                                          const LV<T, N> &s2, const LV<T, N> &s3, LV<T, N> &out)
{
    const T pl = lv_left<T, N>(lm, ctr), pr = lv_right<T, N>(lm, ctr);
#pragma unroll
    for (int c = 0; c < N; ++c) {
        if (((c + PAR) & 1) != COLOR) continue;
        const T pE = c == N - 1 ? pr : ctr.a[c == N - 1 ? c : c + 1], pW = c == 0 ? pl : ctr.a[c == 0 ? 0 : c - 1];
        const T pred = predict_from(pE, pW, p.a[c], m.a[c], s2.a[c], s3.a[c]);
        const T val = k.om1 * ctr.a[c] + k.om * pred;
        out.a[c] = (fluid & (1u << c)) ? val : out.a[c];
    }
}

// PAR0: parity of (g.ybase + j0 - 4), the first window row - a launch constant because RT is even and all tiles start at jb + k RT.
// BND: the tile has non-fluid cells within reach (K7 views are evaluated); FULL: store every cell (carry pass after an upload).
// MIRROR (plain tiles of the two-part launch, every second tile row; round 4): the window is held upside down - w <-> local row j0+RT+3-w - so
// that the tile works, and asks for its rows, from the bottom up.  The pass is bound by its traffic and a wave's 64 row requests trickle out
// over most of its ~20 us of life (the window of an 8-row tile is 124 VGPRs: the compiler loads it as registers fall free): the tile below
// wants the rows it shares with this one FIRST, this one wants them LAST - ten microseconds later they have left the L2 (measured: 36 % hits
// where 50 % are possible, 590 MB read for 350 algorithmic).  With every second tile row mirrored both sharers want a shared row in the same
// phase of their lives.  Same expression per cell: only which register holds which row changes.
template <int N, int RT, int PAR0, int DM, bool BND, bool FULL, typename T, bool MIRROR = false>
__device__ __forceinline__ void rbsor_pair_tile(const Grid &g, const Konst<T> &k, const LaneMapN<N> &lm_in, int i0, int j0, int je, const unsigned (&fl)[RT + 8],
                                                const uint8_t *bcmap, T *C, T *D, const T *A, const T *B, const T *v)
{
    static_assert(!(MIRROR && (BND || FULL)), "only plain tiles are mirrored");
    constexpr int W = RT + 8;                  // window rows w = 0 .. W-1  <->  local rows j0-4 .. j0+RT+3 (clamped into the domain; MIRROR: j0+RT+3 .. j0-4)
    constexpr int UP = MIRROR ? -1 : 1;        // window step towards the row above (j + 1)
    constexpr int PARW = MIRROR ? PAR0 + 1 : PAR0;      // parity of window row w = (PARW + w) & 1  (RT is even: j0+RT+3-w and j0-4+w+1 have the same parity)
#define FS_ROW(w) (MIRROR ? j0 + RT + 3 - (w) : j0 - 4 + (w))
    using R = LV<T, N>;
    constexpr unsigned ALL = (1u << N) - 1u;
    // A tile without a non-fluid cell within reach (halo lanes included) holds no lane at the domain's first / last column - those cells are
    // inflow, outflow or wall in every scene and a clamped halo lane repeats them: the sample() clamp of the x-neighbours (one v_cndmask per
    // DPP shift) folds away (round 4: 64 of the plain kernel's 680 VALU instructions).
    const LaneMapN<N> lm = BND ? lm_in : LaneMapN<N>{lm_in.i0, lm_in.owner, false, false};
    // fluid selector of window row w: in a tile without a single non-fluid cell within reach it is a constant (no registers, no selects)
#define FS_FL(w) (BND ? fl[w] : ALL)
    R PA[W], VX[W], VY[W], PB[W];              // PB[w] is loaded for w = 1 .. W-2
    uint32_t code[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int j = clampy(g, FS_ROW(w));
        PA[w] = lv_field<1, T, N>(A, g, 0, i0, j);
        VX[w] = lv_field<2, T, N>(v, g, 0, i0, j);
        VY[w] = lv_field<2, T, N>(v, g, 1, i0, j);
        if (w >= 1 && w <= W - 2) PB[w] = lv_field<1, T, N>(B, g, 0, i0, j);
        code[w] = BND ? lv_bytes<N>(bcmap, g, i0, j) : 0u;
    }
    // Poisson source of rows 1 .. W-2, once per cell for both iterations
    R S2[W], S3[W];
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) {
        const T xl = lv_left<T, N>(lm, VX[w]), xr = lv_right<T, N>(lm, VX[w]);
        const T yl = lv_left<T, N>(lm, VY[w]), yr = lv_right<T, N>(lm, VY[w]);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const T xE = c == N - 1 ? xr : VX[w].a[c == N - 1 ? c : c + 1], xW = c == 0 ? xl : VX[w].a[c == 0 ? 0 : c - 1];
            const T yE = c == N - 1 ? yr : VY[w].a[c == N - 1 ? c : c + 1], yW = c == 0 ? yl : VY[w].a[c == 0 ? 0 : c - 1];
            source_from<DM>(k, xE, xW, yE, yW, VX[w + UP].a[c], VX[w - UP].a[c], VY[w + UP].a[c], VY[w - UP].a[c], S2[w].a[c], S3[w].a[c]);
        }
    }
    // view(A): rows 0 .. W-1 (a row missing at the window's edge is stood in for by the row itself - see the header)
    R VA[W];
#pragma unroll
    for (int w = 0; w < W; ++w) VA[w] = BND ? lv_bc_row<T, N>(lm, PA[w == 0 ? 0 : w - 1], PA[w], PA[w == W - 1 ? w : w + 1], code[w]) : PA[w];
    // stage 1: odd pass of iteration 1 on rows 1 .. W-2:  B[odd] <- view(A)
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) {
        if ((PARW + w) & 1) rbp_relax<1, 1>(k, lm, FS_FL(w), VA[w - UP], VA[w], VA[w + UP], S2[w], S3[w], PB[w]);
        else                rbp_relax<0, 1>(k, lm, FS_FL(w), VA[w - UP], VA[w], VA[w + UP], S2[w], S3[w], PB[w]);
    }
    // stage 2: even pass of iteration 1 on rows 2 .. W-3, in place on B (all rows from the stage-1 state: results go to P2)
    R P2[W];
#pragma unroll
    for (int w = 2; w <= W - 3; ++w) {
        P2[w] = PB[w];
        if ((PARW + w) & 1) rbp_relax<1, 0>(k, lm, FS_FL(w), PB[w - UP], PB[w], PB[w + UP], S2[w], S3[w], P2[w]);
        else                rbp_relax<0, 0>(k, lm, FS_FL(w), PB[w - UP], PB[w], PB[w + UP], S2[w], S3[w], P2[w]);
    }
    // view(B') on rows 2 .. W-3
    R VB[W];
#pragma unroll
    for (int w = 2; w <= W - 3; ++w) VB[w] = BND ? lv_bc_row<T, N>(lm, P2[w == 2 ? 2 : w - 1], P2[w], P2[w == W - 3 ? w : w + 1], code[w]) : P2[w];
    // stage 3: odd pass of iteration 2 on rows 3 .. W-4:  A[odd] <- view(B'),  the other cells of the row stay view(A)
    R P3[W];
#pragma unroll
    for (int w = 3; w <= W - 4; ++w) {
        P3[w] = VA[w];
        if ((PARW + w) & 1) rbp_relax<1, 1>(k, lm, FS_FL(w), VB[w - UP], VB[w], VB[w + UP], S2[w], S3[w], P3[w]);
        else                rbp_relax<0, 1>(k, lm, FS_FL(w), VB[w - UP], VB[w], VB[w + UP], S2[w], S3[w], P3[w]);
    }
    // stage 4: even pass of iteration 2 on the tile's own rows 4 .. W-5
    R P4[W];
#pragma unroll
    for (int w = 4; w <= W - 5; ++w) {
        P4[w] = P3[w];
        if ((PARW + w) & 1) rbp_relax<1, 0>(k, lm, FS_FL(w), P3[w - UP], P3[w], P3[w + UP], S2[w], S3[w], P4[w]);
        else                rbp_relax<0, 0>(k, lm, FS_FL(w), P3[w - UP], P3[w], P3[w + UP], S2[w], S3[w], P4[w]);
    }
#pragma unroll
    for (int w = 4; w <= W - 5; ++w) {
        const int j = FS_ROW(w);
        if (j >= je) continue;
        const unsigned sel = FULL ? ALL : (FS_FL(w) | (BND ? lv_sel_target<N>(code[w]) : 0u));
        if (lm.owner && sel) {
            lv_store_sel<T, N>(C + idx<1, T>(g, 0, i0, j), P4[w], sel);
            lv_store_sel<T, N>(D + idx<1, T>(g, 0, i0, j), VB[w], sel);
        }
    }
#undef FS_FL
#undef FS_ROW
}

// ---- plain tiles, two stacked waves per workgroup (round 5) ------------------------------------------------------------------------------
// The shrinking window of rbsor_pair_tile requests RT + 8 rows for RT: two rows per output row at RT = 8, and the L2 gives back only part of
// what neighbouring tiles request twice (PMC: 1.3 x the algorithmic bytes on the plain part, which runs at the chip's copy rate on what it
// really moves).  Here a workgroup is TWO waves on a tile of 2 RT rows: the lower wave owns rows j0 .. j0+RT-1 and keeps the shrinking halo
// of 4 rows BELOW, the upper wave owns rows j0+RT .. j0+2RT-1, holds its window upside down (rbsor_pair_tile MIRROR) and keeps the halo ABOVE.
// Towards each other they do not shrink: each loads ONE raw row of the partner (window row RT + 4) for the first half sweep, and after each of
// the first three half sweeps hands its edge row (window row RT + 3) to the partner through LDS (3 x 1 KiB, one barrier each).  Per 2 RT = 16
// output rows: 26 row requests per plane instead of 32, 38 instead of 44 relaxed rows per wave - and the same expression per cell, so the
// same bits.  Plain tiles only (the host lists them per 2 RT rows; the boundary tiles keep the general kernel).
template <int N, int RT, int PAR0, int DM, bool MIRROR, typename T>
__device__ __forceinline__ void rbsor_pair_stack_tile(const Grid &g, const Konst<T> &k, const LaneMapN<N> &lm_in, int i0, int j0,
                                                      T *C, T *D, const T *A, const T *B, const T *v, typename LVec<T, N>::type (*xch)[2][64], int slot)
{
    constexpr int W = RT + 5;                  // window rows: 0 .. 3 the outer halo, 4 .. RT+3 the wave's own rows, RT+4 the partner's edge row
    constexpr int UP = MIRROR ? -1 : 1;
    constexpr int PARW = MIRROR ? PAR0 + 1 : PAR0;
#define FS_ROW(w) (MIRROR ? j0 + RT + 3 - (w) : j0 - 4 + (w))
    using R = LV<T, N>;
    using Q = typename LVec<T, N>::type;
    constexpr unsigned ALL = (1u << N) - 1u;
    const LaneMapN<N> lm{lm_in.i0, lm_in.owner, false, false};
    const int lane = threadIdx.x & 63;
    R PA[W], VX[W], VY[W], PB[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int j = clampy(g, FS_ROW(w));
        PA[w] = lv_field<1, T, N>(A, g, 0, i0, j);
        VX[w] = lv_field<2, T, N>(v, g, 0, i0, j);
        VY[w] = lv_field<2, T, N>(v, g, 1, i0, j);
        if (w >= 1 && w <= W - 2) PB[w] = lv_field<1, T, N>(B, g, 0, i0, j);
    }
    R S2[W], S3[W];
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) {
        const T xl = lv_left<T, N>(lm, VX[w]), xr = lv_right<T, N>(lm, VX[w]);
        const T yl = lv_left<T, N>(lm, VY[w]), yr = lv_right<T, N>(lm, VY[w]);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            const T xE = c == N - 1 ? xr : VX[w].a[c == N - 1 ? c : c + 1], xW = c == 0 ? xl : VX[w].a[c == 0 ? 0 : c - 1];
            const T yE = c == N - 1 ? yr : VY[w].a[c == N - 1 ? c : c + 1], yW = c == 0 ? yl : VY[w].a[c == 0 ? 0 : c - 1];
            source_from<DM>(k, xE, xW, yE, yW, VX[w + UP].a[c], VX[w - UP].a[c], VY[w + UP].a[c], VY[w - UP].a[c], S2[w].a[c], S3[w].a[c]);
        }
    }
    auto put = [&](int e, const R &r) { Q q; q.x = r.a[0]; q.y = r.a[1]; xch[e][slot][lane] = q; };
    auto get = [&](int e) { const Q q = xch[e][1 - slot][lane]; R r; r.a[0] = q.x; r.a[1] = q.y; return r; };
    // stage 1: odd pass of iteration 1 on rows 1 .. W-2:  B[odd] <- A   (a plain tile: view(A) = A)
#pragma unroll
    for (int w = 1; w <= W - 2; ++w) {
        if ((PARW + w) & 1) rbp_relax<1, 1>(k, lm, ALL, PA[w - UP], PA[w], PA[w + UP], S2[w], S3[w], PB[w]);
        else                rbp_relax<0, 1>(k, lm, ALL, PA[w - UP], PA[w], PA[w + UP], S2[w], S3[w], PB[w]);
    }
    put(0, PB[W - 2]);
    __syncthreads();
    PB[W - 1] = get(0);
    // stage 2: even pass of iteration 1 on rows 2 .. W-2, in place on B
    R P2[W];
#pragma unroll
    for (int w = 2; w <= W - 2; ++w) {
        P2[w] = PB[w];
        if ((PARW + w) & 1) rbp_relax<1, 0>(k, lm, ALL, PB[w - UP], PB[w], PB[w + UP], S2[w], S3[w], P2[w]);
        else                rbp_relax<0, 0>(k, lm, ALL, PB[w - UP], PB[w], PB[w + UP], S2[w], S3[w], P2[w]);
    }
    put(1, P2[W - 2]);
    __syncthreads();
    P2[W - 1] = get(1);
    // stage 3: odd pass of iteration 2 on rows 3 .. W-2:  A[odd] <- B'
    R P3[W];
#pragma unroll
    for (int w = 3; w <= W - 2; ++w) {
        P3[w] = PA[w];
        if ((PARW + w) & 1) rbp_relax<1, 1>(k, lm, ALL, P2[w - UP], P2[w], P2[w + UP], S2[w], S3[w], P3[w]);
        else                rbp_relax<0, 1>(k, lm, ALL, P2[w - UP], P2[w], P2[w + UP], S2[w], S3[w], P3[w]);
    }
    put(2, P3[W - 2]);
    __syncthreads();
    P3[W - 1] = get(2);
    // stage 4: even pass of iteration 2 on the wave's own rows 4 .. W-2
    R P4[W];
#pragma unroll
    for (int w = 4; w <= W - 2; ++w) {
        P4[w] = P3[w];
        if ((PARW + w) & 1) rbp_relax<1, 0>(k, lm, ALL, P3[w - UP], P3[w], P3[w + UP], S2[w], S3[w], P4[w]);
        else                rbp_relax<0, 0>(k, lm, ALL, P3[w - UP], P3[w], P3[w + UP], S2[w], S3[w], P4[w]);
    }
    if (lm.owner) {
#pragma unroll
        for (int w = 4; w <= W - 2; ++w) {
            lv_store_row<1, T, N>(C, g, 0, i0, FS_ROW(w), P4[w]);
            lv_store_row<1, T, N>(D, g, 0, i0, FS_ROW(w), P2[w]);
        }
    }
#undef FS_ROW
}

// PATH: 2 - classify the tile here (mask loads) and take the plain or the boundary path; 3 - the plain path without looking (compact launch
// of the workgroups the host found to be plain: its own kernel, so its own register budget - 126 VGPRs = 4 waves per SIMD, where the
// boundary path with its recipe bytes and views needs 156); 0 / 1 - classify and run only the plain / only the boundary tiles (A/B).
template <int N, int RT, int PAR0, int DM, int PATH, bool FULL, typename T>
__device__ __forceinline__ void rbsor_pair_wave_at(const Grid &g, const Konst<T> &k, int wx, int ty, bool hint, bool bnd_fluid, int jb, int je,
                                                   const uint8_t *bcmap, T *C, T *D, const T *A, const T *B, const T *v)
{
    constexpr int W = RT + 8;
    const LaneMapN<N> lm = lane_map_n<N>(g, wx);
    const int i0 = lm.i0, j0 = jb + ty * RT;
    unsigned fl[W];
    // plain: the host listed this workgroup (PATH 3) / this wave (hint) as seeing nothing but fluid within reach (fs_api.hip tile_list) - or the
    // masks say so.  ONE instance of each path in the kernel.
    bool plain = PATH == 3 || hint;
    if (!plain && bnd_fluid) {
        // boundary list of the two-part launch, and the host saw fluid in the tile's own rows: not plain, something to store - the masks are
        // requested with the window instead of in front of it
#pragma unroll
        for (int w = 0; w < W; ++w) fl[w] = lv_sel_fluid<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 4 + w)));
    } else if (!plain) {
        bool own_fluid = false, all_fluid = true;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            fl[w] = lv_sel_fluid<N>(lv_bytes<N>(g.mask, g, i0, clampy(g, j0 - 4 + w)));
            all_fluid = all_fluid && fl[w] == (1u << N) - 1u;
            if (w >= 4 && w <= W - 5 && j0 - 4 + w < je) own_fluid = own_fluid || (lm.owner && fl[w] != 0u);
        }
        if (!FULL && !__any(own_fluid)) {
            // no fluid cell in the rows this tile stores: only K7 targets could change, and a tile without any has nothing to do
            bool tgt = false;
#pragma unroll
            for (int w = 4; w <= W - 5; ++w)
                if (j0 - 4 + w < je) tgt = tgt || (lm.owner && lv_sel_target<N>(lv_bytes<N>(bcmap, g, i0, clampy(g, j0 - 4 + w))) != 0u);
            if (!__any(tgt)) return;
        }
        plain = __all(all_fluid);
    }
    if (plain) {
        if constexpr (PATH == 3 && FS_RBP_MIRROR && !FULL) {
            if (ty & 1) { rbsor_pair_tile<N, RT, PAR0, DM, false, FULL, T, true>(g, k, lm, i0, j0, je, fl, bcmap, C, D, A, B, v); return; }
        }
        if constexpr (PATH != 1) rbsor_pair_tile<N, RT, PAR0, DM, false, FULL, T>(g, k, lm, i0, j0, je, fl, bcmap, C, D, A, B, v);
        return;
    }
    if constexpr (PATH != 0 && PATH != 3) rbsor_pair_tile<N, RT, PAR0, DM, true, FULL, T>(g, k, lm, i0, j0, je, fl, bcmap, C, D, A, B, v);
}

template <int N, int RT, int PAR0, int DM, int PATH, bool FULL, typename T>
__device__ __forceinline__ void rbsor_pair_wave(const Grid &g, const Konst<T> &k, int nbx, int nby, int jb, int je,
                                                const uint8_t *bcmap, T *C, T *D, const T *A, const T *B, const T *v)
{
    int wx, ty;
    bool hint = false, bnd_fluid = false;
    if (!tile_coords_n<N>(g, nbx, nby, jb, je, RT, wx, ty, PATH == 2 && !FULL ? &hint : nullptr, PATH == 2 && !FULL ? &bnd_fluid : nullptr)) return;
    rbsor_pair_wave_at<N, RT, PAR0, DM, PATH, FULL, T>(g, k, wx, ty, hint, bnd_fluid, jb, je, bcmap, C, D, A, B, v);
}

// ---- both kinds of tile in ONE launch (round 6) -----------------------------------------------------------------------------------------
// The two-part launch ends twice on a partly filled chip: 10 422 stacked workgroups are 4.07 rounds of what the chip holds, the 9 425 boundary
// waves 3.07 rounds of long-lived waves at 3 per SIMD.  Measured (round 6): the stacked plain part does not care about its occupancy - 113.9 /
// 113.5 / 112.9 us at 5 / 3 / 2.5 waves per SIMD (dynamic LDS holding it down) - so both bodies fit ONE kernel at the boundary body's register
// budget, and the boundary tiles, listed first, fill in while the all-fluid ones stream (what k_cip_step_all did for fs_cip_step).
// The list (fs_core.hip tile_list, class 3) is over units of 8 rows: an all-fluid 16-row parent tile is ONE entry at its lower unit (hint bit 0) and
// runs the two stacked waves; a unit of any other parent is an entry whose two waves take its two 4-row tiles with the masked body (hint bits
// 1 / 2: fluid in the rows of tile 0 / 1 - the window is then requested with the masks, not behind them).
template <int N, int PAR0, int DM, typename T>
__global__ __launch_bounds__(128) void k_rbsor_pair_all(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, const uint8_t *bcmap, T *C, T *D, const T *A, const T *B, const T *v)
{
    constexpr int RT = 8;
    __shared__ typename LVec<T, N>::type xch[3][2][64];
    constexpr int OW = 64 - 2 * (4 / N);
    int wx, ty, cg;
    unsigned cls = 0u;
    if (!band_coords<1>(g, nbx, nby, wx, ty, cg, 0, &cls)) return;              // (workgroup-uniform)
    if (!(wx * OW < g.X / N && jb + ty * RT < je)) return;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (cls & 1u) {
        const LaneMapN<N> lm = lane_map_n<N>(g, wx);
        const int j0 = jb + ty * RT;
        if (w == 0) rbsor_pair_stack_tile<N, RT, PAR0, DM, false, T>(g, k, lm, lm.i0, j0, C, D, A, B, v, xch, 0);
        else        rbsor_pair_stack_tile<N, RT, PAR0, DM, true, T>(g, k, lm, lm.i0, j0 + RT, C, D, A, B, v, xch, 1);
    } else {
        const int t4 = ty * 2 + w;                                              // this wave's 4-row tile
        if (jb + t4 * 4 >= je) return;                                          // (no barrier on this path)
        rbsor_pair_wave_at<N, 4, PAR0, DM, 2, false, T>(g, k, wx, t4, false, ((cls >> (1 + w)) & 1u) != 0u, jb, je, bcmap, C, D, A, B, v);
    }
}

template <int N, int RT, int PAR0, int DM, int PATH, bool FULL, typename T>
__global__ __launch_bounds__(256, (PATH == 3 && sizeof(T) == 4 ? 4 : 1)) void k_rbsor_pair(Grid g, Konst<T> k, int nbx, int nby, int jb, int je,
                                                    const uint8_t *bcmap, T *C, T *D, const T *A, const T *B, const T *v)
{
    static_assert(RT % 2 == 0, "the row parity of a tile is a launch constant only for even tile heights");
    rbsor_pair_wave<N, RT, PAR0, DM, PATH, FULL, T>(g, k, nbx, nby, jb, je, bcmap, C, D, A, B, v);
}

// (Round 4: since the DPP shifts lost their init moves the plain part needs 97 VGPRs, one more than the 96 that admit a fifth wave per SIMD;
//  built with __launch_bounds__(256, 5): 95 VGPRs, no scratch, 5 waves - and the same 182-187 us: the pass is bound by its traffic, not by occupancy.)

}  // namespace fs
