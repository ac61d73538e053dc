// fs_march.h - register-tile stencil kernels for gfx950 (the fast paths of the step() kernels).
//
// Mapping (CDNA4): a lane owns a QUAD of 4 consecutive x cells (one 16-byte global_load_dwordx4 per field row: a wave moves
// a full 1 KiB row segment per instruction, perfectly coalesced) times RT rows; all RT + 2*radius rows of every input are
// requested up front (tens of KiB in flight per wave - the memory-level parallelism that saturates HBM at 4-6 waves/SIMD;
// an earlier row-MARCHING variant with three rolling rows was latency-bound and is gone: 100 us vs 74 us per Jacobi sweep).
// The x-neighbours of a quad come from the adjacent lanes through DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1 -
// no LDS, no barrier); a wave covers 64 quads of which the inner 62 store ("overlapped waves", see lane_map).
//
// Work skipping: the mask quad (one u32 per lane and row) tells a wave whether any of its cells is active; tiles that are
// all wall neither load nor compute (scene 5 is one third wall).
//
// Blocks are dealt to the 8 XCDs in groups of tile rows (band_coords) so that halo rows are re-read from the local L2.
//
// Arithmetic: identical expression trees / operation order as the one-cell-per-lane kernels in
// fs_kernels.h (and the reference); results are bit-identical.
#pragma once
#include "fs_device.h"
#include "fs_kernels.h"

namespace fs {

template <typename T> struct Quad;
template <> struct Quad<float> { using type = float4; };
template <> struct Quad<double> { using type = double4; };

// value of the previous / next lane (0 in lane 0 / lane 63 and where the source lane is switched off: the caller patches those).
// bound_ctrl:1 says exactly that - with bound_ctrl:0 and an `old` operand of 0 the compiler emitted a v_mov_b32 0 in front of every shift
// (round 4: one VALU instruction less per neighbour exchange in every tile kernel, 5 - 6 % of their VALU count; same values).
__device__ __forceinline__ float lane_prev(float x)
{ return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)); }   // wave_shr:1
__device__ __forceinline__ float lane_next(float x)
{ return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true)); }   // wave_shl:1
// f64: the two halves through the same DPP shifts (round 4; __shfl_up / __shfl_down went through ds_bpermute)
__device__ __forceinline__ double lane_prev(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Addressing of the quad accesses: wave-uniform row base (an SGPR pair) + ONE 32-bit lane offset shared by every access of the lane - the
// `saddr` form of global_load / global_store.  Left to itself the compiler reassociates base + row + lane into (base + lane), a 64-bit
// VGPR pair, + row: one v_lshl_add_u64 and two more live VGPRs per access (a CIP tile has ~40).  Passing the row base through
// readfirstlane (it IS uniform: every row index here derives from tile_coords' wave index) pins the split.  Rows must be wave-uniform.
typedef __attribute__((address_space(1))) const char *fs_gcptr;
typedef __attribute__((address_space(1))) char *fs_gptr;
__device__ __forceinline__ uint64_t uniform64(uint64_t u)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename Q, typename T>
__device__ __forceinline__ Q load_row_quad(const T *row, int i0)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<__attribute__((address_space(1))) const Q *>((fs_gcptr)uniform64((uint64_t)row) + (unsigned)i0 * (unsigned)sizeof(T));
#else       // (the host pass only parses device code; it has no address spaces)
    return *reinterpret_cast<const Q *>(row + i0);
#endif
}

template <int C, typename T>
__device__ __forceinline__ typename Quad<T>::type load_quad(const T *f, const Grid &g, int c, int i0, int j)
{ return load_row_quad<typename Quad<T>::type, T>(f + ((size_t)j * C + c) * g.P, i0); }

// predicated load: lanes whose quad (and whose neighbours' quads) are solid wall fetch nothing
template <int C, typename T>
__device__ __forceinline__ typename Quad<T>::type load_quad_if(bool need, const T *f, const Grid &g, int c, int i0, int j)
{
    typename Quad<T>::type q;
    q.x = q.y = q.z = q.w = (T)0;
    if (need) q = *reinterpret_cast<const typename Quad<T>::type *>(f + idx<C, T>(g, c, i0, j));
    return q;
}

__device__ __forceinline__ unsigned lane_prev_u(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned lane_next_u(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true); }
// does this lane, or a lane next to it, have any active cell?  (its loads feed the neighbours through DPP)
__device__ __forceinline__ bool lane_needed(unsigned active) { return (active | lane_prev_u(active) | lane_next_u(active)) != 0u; }

__device__ __forceinline__ uint32_t mask_quad(const Grid &g, int i0, int j)
{ return load_row_quad<uint32_t, uint8_t>(g.mask + (size_t)j * g.Pm, i0); }

// recipe bytes of the boundary kernels, one per cell (fs_api.hip build_bc_ops): bits 0-6 the pressure recipe (k_jacobi_lazy below),
// bit 7: the cell is a target of the velocity boundary kernel.  Rows are clamped into the domain.
__device__ __forceinline__ uint32_t bcmap_quad(const Grid &g, const uint8_t *bcmap, int i0, int row)
{ return load_row_quad<uint32_t, uint8_t>(bcmap + (size_t)clampy(g, row) * g.Pm, i0); }
__device__ __forceinline__ unsigned sel_bit7(uint32_t c4)
{ return ((c4 >> 7) & 1u) | ((c4 >> 14) & 2u) | ((c4 >> 21) & 4u) | ((c4 >> 28) & 8u); }

// store the components of `v` whose mask byte satisfies the predicate encoded in `sel` (bit k = cell k)
template <typename T>
__device__ __forceinline__ void store_quad_sel(T *dst, const typename Quad<T>::type &v, unsigned sel)
{
    if (sel == 0xfu) { *reinterpret_cast<typename Quad<T>::type *>(dst) = v; return; }
    if (sel & 1u) dst[0] = v.x;
    if (sel & 2u) dst[1] = v.y;
    if (sel & 4u) dst[2] = v.z;
    if (sel & 8u) dst[3] = v.w;
}

// bit k set when byte k of m4 != 1 (not wall) / == 0 (fluid)
__device__ __forceinline__ unsigned sel_not_wall(uint32_t m4)
{
    return ((m4 & 0xffu) != 1u ? 1u : 0u) | (((m4 >> 8) & 0xffu) != 1u ? 2u : 0u) |
           (((m4 >> 16) & 0xffu) != 1u ? 4u : 0u) | (((m4 >> 24) & 0xffu) != 1u ? 8u : 0u);
}
__device__ __forceinline__ unsigned sel_fluid(uint32_t m4)
{
    return ((m4 & 0xffu) == 0u ? 1u : 0u) | (((m4 >> 8) & 0xffu) == 0u ? 2u : 0u) |
           (((m4 >> 16) & 0xffu) == 0u ? 4u : 0u) | (((m4 >> 24) & 0xffu) == 0u ? 8u : 0u);
}

// predict_p on one cell from already-gathered neighbours (fs/pressure_updater.py:23-38, literal order)
template <typename T>
__device__ __forceinline__ T predict_from(T pE, T pW, T pN, T pS, T s2, T s3)
{ return ((T)0.25 * (((pE + pW) + pN) + pS) + s2) - s3; }

template <int DM, typename T>
__device__ __forceinline__ void source_from(const Konst<T> &k, T vxE, T vxW, T vyE, T vyW, T vxN, T vxS, T vyN, T vyS, T &s2, T &s3)
{
    T sxx = vxE - vxW, sxy = vyE - vyW, syx = vxN - vxS, syy = vyN - vyS;
    s2 = ((sxx * sxx + syy * syy) + (syx * sxy)) / (T)8.0;
    s3 = cdiv<DM>(k.dx * (sxx + syy), k.eight_dt, k.r_eight_dt);
}

// ------------------------------------------------------------------------------------------------
// Overlapped-wave column mapping for multi-stage stencils: a wave covers 64 consecutive quads of which the
// inner 62 produce output; lanes 0 and 63 are x-halo lanes (they load and compute so that their inner
// neighbour can read a valid value through DPP, but never store).  3 % redundant loads (L1/L2 hits) buy a
// kernel with no edge loads, no LDS and no barriers, for any number of fused radius-1 stages up to 4.
// ------------------------------------------------------------------------------------------------
struct LaneMap {
    int i0;        // first cell of this lane's quad (clamped into the row)
    bool owner;    // lane produces output
    bool at_lo;    // quad starts at i = 0  (left neighbour of .x is .x itself: sample() clamp)
    bool at_hi;    // quad ends at i = X-1
};
// XCD-aware block mapping for the tile kernels.  Workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8), each with a private 4 MiB L2.  A 1-D launch is decoded so that groups of FS_XCD_GROUP
// consecutive tile rows (x-blocks innermost) stay on one XCD: vertically adjacent tiles, which re-read each
// other's halo rows, then run on the same XCD close in time and the re-read is a local L2 hit.  Groups are dealt
// to the XCDs cyclically, so regions with little work (solid walls) are spread evenly - one contiguous band per
// XCD measured 15 % slower on scene 5 because the dispatcher does not rebalance.  Placement only affects speed.
// Launch geometry of the tile kernels: grid = (8 * nbx, G * ZG, ceil(groups / 8)) with
//   blockIdx.x = 8 * bx + xcd      bx: block column, xcd: which XCD runs it (the dispatcher deals consecutive workgroups of the
//                                  linearised grid to the XCDs round-robin, and the x extent is a multiple of 8)
//   blockIdx.y = ly * ZG + cg      ly: tile row inside the group of G rows, cg: channel group (ZG = 1, 2 or 3 passes over the same
//                                  rows, adjacent in dispatch order: the second pass finds the shared planes in that XCD's L2)
//   blockIdx.z = lg                the lg-th group of this XCD:  by = (lg * 8 + xcd) * G + ly
// No integer division in the decode (the first version, a 1-D grid decoded with two divisions per wave, spent ~70 of the ~700
// instructions of a one-row K2 tile on them).
constexpr int FS_CG_INNER = 1 << 22;
// cls (optional): bits 28 .. 31 of a compact list's entry - bit w: the host found wave w of this workgroup to be PLAIN, nothing but fluid within the
// kernel's reach (fs_api.hip tile_list, lists built with a hint reach; round 4): a kernel that holds both paths takes the plain one
// without loading a mask byte.  0 where the launch is dense or the list carries no hints.
template <int ZG = 1>
__device__ __forceinline__ bool band_coords(const Grid &g, int nbx, int nby_packed, int &bx, int &by, int &cg, int zoff = 0, unsigned *cls = nullptr)
{
    if (cls) *cls = 0u;
    if (g.tiles) {
        // compact launch: blockIdx.x = (k * ZG + cg) * 8 + xcd  ->  the k-th listed workgroup of that XCD, channel group cg (the passes
        // over one tile are consecutive workgroups of one XCD)
        const int t = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
        const int k = ZG == 1 ? t : t / ZG;
        cg = ZG == 1 ? 0 : t - k * ZG;
        const uint32_t e = g.tiles[k * 8 + xcd];
        bx = (int)(e & 0xfffu);
        by = (int)((e >> 12) & 0xffffu);
        if (cls) *cls = e >> 28;
        return e != 0xffffffffu;
    }
    const int nby = nby_packed & 0x3fffff, FS_XCD_GROUP = (nby_packed >> 24) + 1;   // group size rides in the top byte, bit 23 = stacked, bit 22 = channel groups innermost
    const int xcd = blockIdx.x & 7;
    int ly;
    if (ZG > 1 && (nby_packed & FS_CG_INNER)) {
        // channel groups innermost: blockIdx.x = (bx * ZG + cg) * 8 + xcd - the ZG passes over one tile are consecutive workgroups of ONE XCD
        // (they share most of their input rows: the second finds them in that XCD's L2 while they are hot)
        const int t = (int)blockIdx.x >> 3;
        bx = t / ZG;                                                        // division by a compile-time 2 or 3
        cg = t - bx * ZG;
        ly = (int)blockIdx.y;
    } else {
        bx = blockIdx.x >> 3;
        ly = ZG == 1 ? (int)blockIdx.y : (int)blockIdx.y / ZG;
        cg = ZG == 1 ? 0 : (int)blockIdx.y - ly * ZG;
    }
    by = (((int)blockIdx.z - zoff) * 8 + xcd) * FS_XCD_GROUP + ly;       // zoff: leading z slices that belong to someone else
    return by < nby;
}
// Workgroup shape.  Side by side (default): the waves of a workgroup are consecutive wave columns of ONE tile row.  Stacked (bit 23 of
// nby_packed): they are consecutive tile rows of ONE wave column, so the halo rows a tile shares with the tile below are re-read
// by the same CU within microseconds (L1 / local L2 hits).  Returns this wave's column and tile row, false if it has no work.
// Everything here is wave-uniform and SAID to be (readfirstlane on the wave index): row numbers, and with them the row part of
// every address, then live in scalar registers - the loads take a scalar base + one 32-bit lane offset instead of a 64-bit
// multiply-add per load and lane.
constexpr int FS_STACKED = 1 << 23;
template <int ZG = 1>
__device__ __forceinline__ bool tile_coords(const Grid &g, int nbx, int nby_packed, int jb, int je, int rt, int &wave_x, int &tile_y, int &cg, int zoff = 0)
{
    int bx, by;
    if (!band_coords<ZG>(g, nbx, nby_packed, bx, by, cg, zoff)) return false;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    if (nby_packed & FS_STACKED) { wave_x = bx; tile_y = by * nw + w; }
    else { wave_x = bx * nw + w; tile_y = by; }
    return wave_x * 62 < (g.X >> 2) && jb + tile_y * rt < je;
}
__device__ __forceinline__ bool tile_coords(const Grid &g, int nbx, int nby_packed, int jb, int je, int rt, int &wave_x, int &tile_y)
{
    int cg;
    return tile_coords<1>(g, nbx, nby_packed, jb, je, rt, wave_x, tile_y, cg);
}

__device__ __forceinline__ LaneMap lane_map_wave(const Grid &g, int wave)
{
    const int lane = threadIdx.x & 63;
    const int nq = g.X >> 2;
    int q = wave * 62 - 1 + lane;
    LaneMap m;
    m.owner = lane >= 1 && lane <= 62 && q >= 0 && q < nq;
    q = q < 0 ? 0 : (q > nq - 1 ? nq - 1 : q);
    m.i0 = q << 2;
    m.at_lo = q == 0;
    m.at_hi = q == nq - 1;
    return m;
}

template <typename T>
__device__ __forceinline__ T quad_left(const LaneMap &m, const typename Quad<T>::type &q)
{ T l = lane_prev(q.w); return m.at_lo ? q.x : l; }
template <typename T>
__device__ __forceinline__ T quad_right(const LaneMap &m, const typename Quad<T>::type &q)
{ T r = lane_next(q.x); return m.at_hi ? q.w : r; }

// x / dx for the central differences: a true IEEE division unless dx is a power of two, where the
// (exact) multiplication by 1/dx gives the same bits for a fraction of the instructions.
template <int DM, typename T>
__device__ __forceinline__ T div_dx(T x, const Konst<T> &k) { return xdiv<DM & (DM_P2 | DM_F64)>(x, k.dx, k.inv_dx, k.r_dx); }   // P2 only when k.p2

// (K5 + K6 fused and K2 live in fs_k34n.h, on lanes of 2 cells)

// (the fused red-black iteration lives in fs_k34n.h, on lanes of 2 cells)

template <typename T>
struct Q4 {
    T a[4];
    __device__ __forceinline__ Q4() {}
    __device__ __forceinline__ Q4(const typename Quad<T>::type &q) { a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w; }
    __device__ __forceinline__ typename Quad<T>::type quad() const { typename Quad<T>::type q; q.x = a[0]; q.y = a[1]; q.z = a[2]; q.w = a[3]; return q; }
};

// ------------------------------------------------------------------------------------------------
// K4  CIP advection (fs/solver.py:267-332), quad form.  A lane advects NC channels [c0, c0+NC) of a C-channel
// field for its 4 cells of one row: rows j-1, j, j+1 of the value field and of both gradient fields are
// requested up front as 16-byte loads (9*NC + 6 independent loads per lane), the x-neighbours come from the
// adjacent lanes (DPP; overlapped-wave mapping, no edge loads), and the data-dependent upwind cell
// (i - sign(u), j - sign(v)) becomes a per-cell select among the 3x3 gathered values.  blockIdx.z selects the
// channel group (dye: 3 single-channel passes sharing the advecting velocity).
// ------------------------------------------------------------------------------------------------
template <int C, int NC, bool SELF, int DM, bool CLAMP01, typename T>
__device__ __forceinline__ void cip_advect_quad_tile(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fn, T *fxn, T *fyn,
                                                         const T *fc, const T *fxc, const T *fyc, const T *v, unsigned *hot)
{
    int bx, by, cg;
    if (!tile_coords<C / NC>(g, nbx, nby, jb, je, 1, bx, by, cg)) return;   // bx: wave column, by: tile row, cg: channel group
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0;
    const int j = jb + by;
    const int c0 = cg * NC;
    const unsigned fl = sel_fluid(mask_quad(g, i0, j));
    if (!__any(fl != 0u)) return;
    const int jm = clampy(g, j - 1), jp = clampy(g, j + 1);

    Q4<T> F[NC][3], FX[NC][3], FY[NC][3];      // rows j-1, j, j+1
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        F[c][0] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, jm)); F[c][1] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, j)); F[c][2] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, jp));
        FX[c][0] = Q4<T>(load_quad<C>(fxc, g, c0 + c, i0, jm)); FX[c][1] = Q4<T>(load_quad<C>(fxc, g, c0 + c, i0, j)); FX[c][2] = Q4<T>(load_quad<C>(fxc, g, c0 + c, i0, jp));
        FY[c][0] = Q4<T>(load_quad<C>(fyc, g, c0 + c, i0, jm)); FY[c][1] = Q4<T>(load_quad<C>(fyc, g, c0 + c, i0, j)); FY[c][2] = Q4<T>(load_quad<C>(fyc, g, c0 + c, i0, jp));
    }
    Q4<T> VX[3], VY[3];
    if (SELF) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { VX[r] = F[0][r]; VY[r] = F[1][r]; }
    } else {
        VX[0] = Q4<T>(load_quad<2>(v, g, 0, i0, jm)); VX[1] = Q4<T>(load_quad<2>(v, g, 0, i0, j)); VX[2] = Q4<T>(load_quad<2>(v, g, 0, i0, jp));
        VY[0] = Q4<T>(load_quad<2>(v, g, 1, i0, jm)); VY[1] = Q4<T>(load_quad<2>(v, g, 1, i0, j)); VY[2] = Q4<T>(load_quad<2>(v, g, 1, i0, jp));
    }
    // x-neighbours (left of cell 0 / right of cell 3)
    const T vxl = quad_left<T>(lm, VX[1].quad()), vxr = quad_right<T>(lm, VX[1].quad());
    const T vyl = quad_left<T>(lm, VY[1].quad()), vyr = quad_right<T>(lm, VY[1].quad());
    T fl_[NC][3], fr_[NC][3], fxl[NC], fxr[NC], fyl[NC], fyr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { fl_[c][r] = quad_left<T>(lm, F[c][r].quad()); fr_[c][r] = quad_right<T>(lm, F[c][r].quad()); }
        fxl[c] = quad_left<T>(lm, FX[c][1].quad()); fxr[c] = quad_right<T>(lm, FX[c][1].quad());
        fyl[c] = quad_left<T>(lm, FY[c][1].quad()); fyr[c] = quad_right<T>(lm, FY[c][1].quad());
    }
    Q4<T> OF[NC], OFX[NC], OFY[NC];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const T vx = VX[1].a[q], vy = VY[1].a[q];
        const bool nx = vx < (T)0.0, ny = vy < (T)0.0;       // upwind cell is E / N when the velocity is negative
        const T vxE = q == 3 ? vxr : VX[1].a[q == 3 ? 3 : q + 1], vxW = q == 0 ? vxl : VX[1].a[q == 0 ? 0 : q - 1];
        const T vyE = q == 3 ? vyr : VY[1].a[q == 3 ? 3 : q + 1], vyW = q == 0 ? vyl : VY[1].a[q == 0 ? 0 : q - 1];
        const T dxx = xdiv<DM>((T)0.5 * (vxE - vxW), k.dx, k.inv_dx, k.r_dx), dxy = xdiv<DM>((T)0.5 * (vyE - vyW), k.dx, k.inv_dx, k.r_dx);
        const T dyx = xdiv<DM>((T)0.5 * (VX[2].a[q] - VX[0].a[q]), k.dx, k.inv_dx, k.r_dx), dyy = xdiv<DM>((T)0.5 * (VY[2].a[q] - VY[0].a[q]), k.dx, k.inv_dx, k.r_dx);
        const int ru = ny ? 2 : 0;                             // row of the upwind cell
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const T fE1 = q == 3 ? fr_[c][1] : F[c][1].a[q == 3 ? 3 : q + 1], fW1 = q == 0 ? fl_[c][1] : F[c][1].a[q == 0 ? 0 : q - 1];
            const T fE0 = q == 3 ? fr_[c][0] : F[c][0].a[q == 3 ? 3 : q + 1], fW0 = q == 0 ? fl_[c][0] : F[c][0].a[q == 0 ? 0 : q - 1];
            const T fE2 = q == 3 ? fr_[c][2] : F[c][2].a[q == 3 ? 3 : q + 1], fW2 = q == 0 ? fl_[c][2] : F[c][2].a[q == 0 ? 0 : q - 1];
            const T fxE = q == 3 ? fxr[c] : FX[c][1].a[q == 3 ? 3 : q + 1], fxW = q == 0 ? fxl[c] : FX[c][1].a[q == 0 ? 0 : q - 1];
            const T fyE = q == 3 ? fyr[c] : FY[c][1].a[q == 3 ? 3 : q + 1], fyW = q == 0 ? fyl[c] : FY[c][1].a[q == 0 ? 0 : q - 1];
            const T f00 = F[c][1].a[q];
            const T f0m = ny ? F[c][2].a[q] : F[c][0].a[q];
            const T fm0 = nx ? fE1 : fW1;
            const T fmm = ny ? (nx ? fE2 : fW2) : (nx ? fE0 : fW0);
            const T fx00 = FX[c][1].a[q], fxm0 = nx ? fxE : fxW, fx0m = ny ? FX[c][2].a[q] : FX[c][0].a[q];
            const T fy00 = FY[c][1].a[q], fy0m = ny ? FY[c][2].a[q] : FY[c][0].a[q], fym0 = nx ? fyE : fyW;
            (void)ru;
            cip_point<DM>(k, vx, vy, dxx, dxy, dyx, dyy, f00, f0m, fm0, fmm, fx00, fxm0, fx0m, fy00, fy0m, fym0,
                      OF[c].a[q], OFX[c].a[q], OFY[c].a[q]);
        }
    }
    if (CLAMP01) {      // clamp_field(dye, 0, 1) (fs/solver.py:46-49) folded into the store of the advected value
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) OF[c].a[q] = tmin(tmax(OF[c].a[q], (T)0.0), (T)1.0);
    }
    if (lm.owner && fl) {
        if (C == 2) {       // the advected field is a velocity: keep its "hot" flag honest (per component when the pass holds one)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                raise_hot(hot, ((fl >> q) & 1u) && (NC == 2 ? hot2(OF[0].a[q], OF[NC - 1].a[q]) : hot1(OF[0].a[q])));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            store_quad_sel<T>(fn + idx<C, T>(g, c0 + c, i0, j), OF[c].quad(), fl);
            store_quad_sel<T>(fxn + idx<C, T>(g, c0 + c, i0, j), OFX[c].quad(), fl);
            store_quad_sel<T>(fyn + idx<C, T>(g, c0 + c, i0, j), OFY[c].quad(), fl);
        }
    }
}
template <int C, int NC, bool SELF, int DM, bool CLAMP01, typename T>
__global__ __launch_bounds__(256) void k_cip_advect_quad(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fn, T *fxn, T *fyn,
                                                         const T *fc, const T *fxc, const T *fyc, const T *v, unsigned *hot)
{
    cip_advect_quad_tile<C, NC, SELF, DM, CLAMP01, T>(g, k, nbx, nby, jb, je, fn, fxn, fyn, fc, fxc, fyc, v, hot);
}


// ------------------------------------------------------------------------------------------------
// K3  _non_advection_phase_grad (fs/solver.py:242-261), quad form, NC channels [c0, c0+NC) per lane (blockIdx.y = group).
// Out-of-range neighbours are clamped (SURVEY.md H2 policy), as in the one-cell-per-lane kernel.
// ------------------------------------------------------------------------------------------------
template <int C, int NC, int DM, typename T>
__device__ __forceinline__ void cip_nonadv_grad_quad_tile(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fxn, T *fyn,
                                                              const T *fxc, const T *fyc, const T *fc, const T *fn)
{
    int bx, by, cg;
    if (!tile_coords<C / NC>(g, nbx, nby, jb, je, 1, bx, by, cg)) return;   // bx: wave column, by: tile row, cg: channel group
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0, j = jb + by, c0 = cg * NC;
    const unsigned nw = sel_not_wall(mask_quad(g, i0, j));
    if (!__any(nw != 0u)) return;
    const int jm = clampy(g, j - 1), jp = clampy(g, j + 1);
    Q4<T> N[NC][3], Fc[NC][3], GX[NC], GY[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        N[c][0] = Q4<T>(load_quad<C>(fn, g, c0 + c, i0, jm)); N[c][1] = Q4<T>(load_quad<C>(fn, g, c0 + c, i0, j)); N[c][2] = Q4<T>(load_quad<C>(fn, g, c0 + c, i0, jp));
        Fc[c][0] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, jm)); Fc[c][1] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, j)); Fc[c][2] = Q4<T>(load_quad<C>(fc, g, c0 + c, i0, jp));
        GX[c] = Q4<T>(load_quad<C>(fxc, g, c0 + c, i0, j));
        GY[c] = Q4<T>(load_quad<C>(fyc, g, c0 + c, i0, j));
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const T nl = quad_left<T>(lm, N[c][1].quad()), nr = quad_right<T>(lm, N[c][1].quad());
        const T cl = quad_left<T>(lm, Fc[c][1].quad()), cr = quad_right<T>(lm, Fc[c][1].quad());
        Q4<T> OX, OY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T nE = q == 3 ? nr : N[c][1].a[q == 3 ? 3 : q + 1], nW = q == 0 ? nl : N[c][1].a[q == 0 ? 0 : q - 1];
            const T cE = q == 3 ? cr : Fc[c][1].a[q == 3 ? 3 : q + 1], cW = q == 0 ? cl : Fc[c][1].a[q == 0 ? 0 : q - 1];
            const T sx = ((nE - cE) - nW) + cW;
            const T sy = ((N[c][2].a[q] - Fc[c][2].a[q]) - N[c][0].a[q]) + Fc[c][0].a[q];
            OX.a[q] = GX[c].a[q] + xdiv<DM>(sx, k.two_dx, k.inv_two_dx, k.r_two_dx);
            OY.a[q] = GY[c].a[q] + xdiv<DM>(sy, k.two_dx, k.inv_two_dx, k.r_two_dx);
        }
        if (lm.owner && nw) {
            store_quad_sel<T>(fxn + idx<C, T>(g, c0 + c, i0, j), OX.quad(), nw);
            store_quad_sel<T>(fyn + idx<C, T>(g, c0 + c, i0, j), OY.quad(), nw);
        }
    }
}
template <int C, int NC, int DM, typename T>
__global__ __launch_bounds__(256) void k_cip_nonadv_grad_quad(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fxn, T *fyn,
                                                              const T *fxc, const T *fyc, const T *fc, const T *fn)
{
    cip_nonadv_grad_quad_tile<C, NC, DM, T>(g, k, nbx, nby, jb, je, fxn, fyn, fxc, fyc, fc, fn);
}


// (K3 + K4 in one pass - velocity and dye - live in fs_k34n.h)

// ------------------------------------------------------------------------------------------------
// K8J  JacobiPressureUpdater._update (fs/pressure_updater.py:62-66), overlapped-wave register tile: x-neighbours of the wave-edge quads come
// from halo lanes (DPP) instead of per-row edge loads, and blocks are dealt to the XCDs in groups of tile rows.
// ------------------------------------------------------------------------------------------------
template <bool SRC, int RT, int DM, typename T>
__device__ __forceinline__ void jacobi_ov_tile(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *pn, const T *pc, const T *vs)
{
    int bx, by;
    if (!tile_coords(g, nbx, nby, jb, je, RT, bx, by)) return;   // bx: wave column, by: tile row
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0;
    const int j0 = jb + by * RT;

    unsigned sel[RT];
    unsigned any = 0u;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        sel[r] = j0 + r < je ? sel_not_wall(mask_quad(g, i0, j0 + r)) : 0u;
        any |= sel[r];
    }
    if (!__any(any != 0u)) return;

    Q4<T> P[RT + 2], VX[RT + 2], VY[RT + 2], S2[RT], S3[RT];
#pragma unroll
    for (int r = 0; r < RT + 2; ++r) {
        const int j = clampy(g, j0 - 1 + r);
        P[r] = Q4<T>(load_quad<1>(pc, g, 0, i0, j));
        if (!SRC) { VX[r] = Q4<T>(load_quad<2>(vs, g, 0, i0, j)); VY[r] = Q4<T>(load_quad<2>(vs, g, 1, i0, j)); }
    }
    if (SRC) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int j = clampy(g, j0 + r);
            S2[r] = Q4<T>(load_quad<2>(vs, g, 0, i0, j));      // (per-lane predication of loads measured slower: each becomes its own
            S3[r] = Q4<T>(load_quad<2>(vs, g, 1, i0, j));      //  exec-masked basic block and the loads no longer issue back to back)
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int j = j0 + r;
        if (j >= je) break;
        const Q4<T> &c = P[r + 1], &n = P[r + 2], &m = P[r];
        const T pl = quad_left<T>(lm, c.quad()), pr = quad_right<T>(lm, c.quad());
        T xl = 0, xr = 0, yl = 0, yr = 0;
        if (!SRC) {
            xl = quad_left<T>(lm, VX[r + 1].quad()); xr = quad_right<T>(lm, VX[r + 1].quad());
            yl = quad_left<T>(lm, VY[r + 1].quad()); yr = quad_right<T>(lm, VY[r + 1].quad());
        }
        Q4<T> o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            T s2, s3;
            if (SRC) { s2 = S2[r].a[q]; s3 = S3[r].a[q]; }
            else {
                const Q4<T> &xc = VX[r + 1], &yc = VY[r + 1];
                const T xE = q == 3 ? xr : xc.a[q == 3 ? 3 : q + 1], xW = q == 0 ? xl : xc.a[q == 0 ? 0 : q - 1];
                const T yE = q == 3 ? yr : yc.a[q == 3 ? 3 : q + 1], yW = q == 0 ? yl : yc.a[q == 0 ? 0 : q - 1];
                source_from<DM>(k, xE, xW, yE, yW, VX[r + 2].a[q], VX[r].a[q], VY[r + 2].a[q], VY[r].a[q], s2, s3);
            }
            const T pE = q == 3 ? pr : c.a[q == 3 ? 3 : q + 1], pW = q == 0 ? pl : c.a[q == 0 ? 0 : q - 1];
            o.a[q] = predict_from(pE, pW, n.a[q], m.a[q], s2, s3);
        }
        if (lm.owner && sel[r]) store_quad_sel<T>(pn + idx<1, T>(g, 0, i0, j), o.quad(), sel[r]);
    }
}
template <bool SRC, int RT, int DM, typename T>
__global__ __launch_bounds__(256) void k_jacobi_ov(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *pn, const T *pc, const T *vs)
{
    jacobi_ov_tile<SRC, RT, DM, T>(g, k, nbx, nby, jb, je, pn, pc, vs);
}

// (Round 4 built the same sweep on ALIGNED wave columns - every lane an owner, the row segment exactly 8 lines of 128 B, the cell beyond each
// end fetched by lanes 0 / 63 with one two-lane load per row and plane - to take the 8 % of read traffic the overlapped mapping pays for
// straddling a ninth line (profiles/r4_marchbw.txt).  Bit-identical, and slower: 96.8 against 84.0 us reading v, 89.6 against 72.1 us on
// the source pair at bc5 res 4096 - the exec-masked edge loads and their dependent selects cost more than the line.  Removed.)


// ------------------------------------------------------------------------------------------------
// K8J with the pressure boundary condition evaluated ON THE FLY ("lazy" K7), for long Jacobi runs (BASELINE configs[1]: 50 sweeps
// per step, where the separate boundary kernel - two dependent HBM round trips + a launch, 6.6 us - is 28 % of the step).
//
// K7 (fs/boundary_condition.py:41-65) is a pure gather from the buffer the previous sweep wrote: a wall cell takes the value (or the
// mean of two values) of FLUID neighbours, an outflow cell 0, an inflow cell the old value of its right neighbour; no assignment reads
// a cell an earlier assignment of the same call wrote (sources are fluid, or lie later in the serial order).  So the value K7 would
// have stored in a cell is a function of the RAW sweep output around it, and the next sweep can evaluate it when it needs it:
//     p_bc(c) = not a target ? raw(c) : recipe(c)(raw)         recipe = copy of a 4-neighbour | mean of two | 0
// `bcmap` (one byte per cell, built with the op lists at mask upload) holds the recipe; `flags` (one byte per wave-tile row) says
// whether any not-wall cell of the tile has a target among its 4 neighbours - only those tiles load the two extra rows of p and the
// three rows of bcmap, every other tile is the plain source-pair sweep.  The host runs n - 2 lazy sweeps and then the last two with
// the real K7, which leaves both physical p buffers exactly as the reference's n x (K7, sweep, swap) does (fs/pressure_updater.py).
// Host-checked preconditions (fs_api.hip lazy_ok): every source is a not-wall cell (an inflow cell whose right neighbour is a wall
// would read that buffer's history), and rows 0 / Y-1 hold no not-wall cell (no clamped y neighbour of a computed cell).
// bcmap byte: bit 0 target, bits 1-2 kind (0 copy, 1 mean, 2 zero), bits 3-4 direction of source 1, bits 5-6 of source 2
// (0 = i-1, 1 = i+1, 2 = j-1, 3 = j+1).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T lazy_value(unsigned code, T raw, T sL, T sR, T sD, T sU)
{
    if (!(code & 1u)) return raw;
    const unsigned kind = (code >> 1) & 3u, d1 = (code >> 3) & 3u, d2 = (code >> 5) & 3u;
    const T a = d1 == 0u ? sL : (d1 == 1u ? sR : (d1 == 2u ? sD : sU));
    const T b = d2 == 0u ? sL : (d2 == 1u ? sR : (d2 == 2u ? sD : sU));
    return kind == 0u ? a : (kind == 1u ? (a + b) / (T)2.0 : (T)0.0);
}

// flags[wx * rows + r]: bit 0 - a not-wall owner cell of wave column wx in row r has a boundary-condition target among its 4 neighbours;
// (bit 4 is added by k_pair_list) bit 1 - a wall or a target lies in row r within 2 columns of the wave's owner cells; within 4 columns of them: bit 3 - a target with a
// source in another row, bit 2 - such a target in a wall one cell thick, or a wall cell WITHOUT a recipe that a not-wall cell reads (k_jacobi_pair)
static __global__ __launch_bounds__(256) void k_lazy_flags(Grid g, int nwx, const uint8_t *bcmap, uint8_t *flags)
{
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wx = wid % nwx, r = wid / nwx;
    if (r >= g.rows) return;
    const LaneMap lm = lane_map_wave(g, wx);
    const int i0 = lm.i0;
    const uint32_t m4 = mask_quad(g, i0, r), bm = bcmap_quad(g, bcmap, i0, r - 1), bc = bcmap_quad(g, bcmap, i0, r), bp = bcmap_quad(g, bcmap, i0, r + 1);
    // codes of cells i0-1 / i0+4 (the shifts sit OUTSIDE the conditional: inside an arm of ?: only the lanes taking that arm would execute
    // them, and a DPP read from a lane that is switched off returns 0)
    const uint32_t bc_prev = lane_prev_u(bc), bc_next = lane_next_u(bc);
    const uint32_t bl = lm.at_lo ? (bc << 24) : bc_prev, br = lm.at_hi ? (bc >> 24) : bc_next;
    bool hit = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (((m4 >> (8 * q)) & 0xffu) == 1u) continue;                                   // wall cells are not computed
        const unsigned cE = q == 3 ? (br & 0xffu) : ((bc >> (8 * (q + 1))) & 0xffu), cW = q == 0 ? (bl >> 24) : ((bc >> (8 * (q - 1))) & 0xffu);
        hit |= ((cE | cW | (bm >> (8 * q)) | (bp >> (8 * q))) & 1u) != 0u;
    }
    // per cell: wall or target
    uint32_t d4 = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (((m4 >> (8 * q)) & 0xffu) == 1u || ((bc >> (8 * q)) & 1u)) d4 |= 1u << q;
    // per cell, v4: a target with a source in another row (copy of the cell below / above, mean).  x4: such a target whose OTHER vertical
    // neighbour is computed too (a wall one cell thick: the reader on one side needs the row beyond the other side), or a wall cell
    // without a recipe that a not-wall cell reads (its content is history, and the two buffers' histories differ)
    const uint32_t mm = mask_quad(g, i0, clampy(g, r - 1)), mp = mask_quad(g, i0, clampy(g, r + 1));
    const uint32_t m_prev = lane_prev_u(m4), m_next = lane_next_u(m4);
    uint32_t ml = lm.at_lo ? (m4 << 24) : m_prev, mr = lm.at_hi ? (m4 >> 24) : m_next;
    // the halo lanes have no outer neighbour in this wave: a wall, as far as this wave column is concerned (who reads the outermost halo
    // cell from outside is 5 columns away from the owner cells - beyond the reach of two sweeps)
    if (lane == 0 && !lm.at_lo) ml = 0x01000000u;
    if (lane == 63 && !lm.at_hi) mr = 0x00000001u;
    uint32_t x4 = 0u, v4 = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned code = (bc >> (8 * q)) & 0xffu, kind = (code >> 1) & 3u, d1 = (code >> 3) & 3u, d2 = (code >> 5) & 3u;
        const unsigned mE = q == 3 ? (mr & 0xffu) : ((m4 >> (8 * (q + 1))) & 0xffu), mW = q == 0 ? (ml >> 24) : ((m4 >> (8 * (q - 1))) & 0xffu);
        const unsigned mS = (mm >> (8 * q)) & 0xffu, mN = (mp >> (8 * q)) & 0xffu;
        const bool wall = ((m4 >> (8 * q)) & 0xffu) == 1u;
        if (code & 1u) {
            const unsigned dv = kind == 0u ? d1 : (kind == 1u ? d2 : 0u);       // the vertical source, if any: 2 = the cell below, 3 = above
            if (kind != 2u && dv >= 2u) {
                v4 |= 1u << q;
                if ((dv == 2u ? mN : mS) != 1u) x4 |= 1u << q;
            }
        } else if (wall && (mE != 1u || mW != 1u || mS != 1u || mN != 1u)) x4 |= 1u << q;
    }
    const uint32_t d_prev = lane_prev_u(d4), d_next = lane_next_u(d4), x_prev = lane_prev_u(x4), x_next = lane_next_u(x4);
    const uint32_t v_prev = lane_prev_u(v4), v_next = lane_next_u(v4);
    const bool near = d4 != 0u || (!lm.at_lo && (d_prev & 0xcu)) || (!lm.at_hi && (d_next & 0x3u));
    const bool hard = x4 != 0u || (!lm.at_lo && x_prev) || (!lm.at_hi && x_next);
    const bool vert = v4 != 0u || (!lm.at_lo && v_prev) || (!lm.at_hi && v_next);
    const bool any = __any(hit && lm.owner), any2 = __any(near && lm.owner), any4 = __any(hard && lm.owner), any8 = __any(vert && lm.owner);
    if (lane == 0) flags[(size_t)wx * g.rows + r] = (any ? 1 : 0) | (any2 ? 2 : 0) | (any4 ? 4 : 0) | (any8 ? 8 : 0);
}

// One row of the plain source-pair sweep on register rows: m / c / n = rows j-1 / j / j+1 of the previous iterate.
template <typename T>
__device__ __forceinline__ Q4<T> plain_row(const LaneMap &lm, const Q4<T> &m, const Q4<T> &c, const Q4<T> &n, const Q4<T> &S2, const Q4<T> &S3)
{
    const T pl = quad_left<T>(lm, c.quad()), pr = quad_right<T>(lm, c.quad());
    Q4<T> o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const T pE = q == 3 ? pr : c.a[q == 3 ? 3 : q + 1], pW = q == 0 ? pl : c.a[q == 0 ? 0 : q - 1];
        o.a[q] = predict_from(pE, pW, n.a[q], m.a[q], S2.a[q], S3.a[q]);
    }
    return o;
}

// K7 on one register row whose targets are all "0" or "copy of the left / right neighbour" (inflow / outflow columns): no other row involved
template <typename T>
__device__ __forceinline__ Q4<T> bc_row_h(const LaneMap &lm, const Q4<T> &c, uint32_t code)
{
    const T cl = quad_left<T>(lm, c.quad()), cr = quad_right<T>(lm, c.quad());
    Q4<T> o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned k = (code >> (8 * q)) & 0xffu;
        const T sL = q == 0 ? cl : c.a[q == 0 ? 0 : q - 1], sR = q == 3 ? cr : c.a[q == 3 ? 3 : q + 1];
        const T v = (k & 4u) ? (T)0.0 : ((k & 8u) ? sR : sL);            // kind 2 (bit 2 of the byte) = zero; direction bit 3: 0 left, 1 right
        o.a[q] = (k & 1u) ? v : c.a[q];
    }
    return o;
}

// One row of the buffer as K7 would leave it, from RAW rows: m / c / n = rows j-1 / j / j+1, code = recipe bytes of row j.  Each cell's
// boundary value is evaluated ONCE here and the stencil then runs on the finished rows (evaluating it per stencil neighbour costs 5x the
// selects).  At the domain's first / last column sample() clamps onto the cell itself - no recipe points outside (fs_api.hip build_bc_ops).
// Must be called by the whole wave (cross-lane shifts).
template <typename T>
__device__ __forceinline__ Q4<T> bc_row(const LaneMap &lm, const Q4<T> &m, const Q4<T> &c, const Q4<T> &n, uint32_t code)
{
    if (!__any((code & 0x01010101u) != 0u)) return c;            // wave-uniform: no target in this row of the wave
    const T cl = quad_left<T>(lm, c.quad()), cr = quad_right<T>(lm, c.quad());
    Q4<T> o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const T sL = q == 0 ? cl : c.a[q == 0 ? 0 : q - 1], sR = q == 3 ? cr : c.a[q == 3 ? 3 : q + 1];
        o.a[q] = lazy_value((code >> (8 * q)) & 0xffu, c.a[q], sL, sR, m.a[q], n.a[q]);
    }
    return o;
}

template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_lazy(Grid g, int nbx, int nby, int jb, int je, const uint8_t *bcmap, const uint8_t *flags,
                                                     T *pn, const T *pc, const T *src)
{
    int bx, by;
    if (!tile_coords(g, nbx, nby, jb, je, 1, bx, by)) return;   // bx: wave column, by: tile row
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0, j = jb + by;
    const uint32_t m4 = mask_quad(g, i0, j);
    const unsigned sel = sel_not_wall(m4);
    if (!__any(sel != 0u)) return;
    const bool lazy = (flags[(size_t)bx * g.rows + j] & 1u) != 0;      // wave-uniform (scalar load)
    const Q4<T> S2(load_quad<2>(src, g, 0, i0, j)), S3(load_quad<2>(src, g, 1, i0, j));
    Q4<T> o;
    if (!lazy) {      // nothing around this tile takes a boundary value: the plain sweep
        const Q4<T> m(load_quad<1>(pc, g, 0, i0, clampy(g, j - 1))), c(load_quad<1>(pc, g, 0, i0, j)), n(load_quad<1>(pc, g, 0, i0, clampy(g, j + 1)));
        o = plain_row<T>(lm, m, c, n, S2, S3);
    } else {
        Q4<T> P[5];                                              // rows j-2 .. j+2 of the raw buffer
#pragma unroll
        for (int r = 0; r < 5; ++r) P[r] = Q4<T>(load_quad<1>(pc, g, 0, i0, clampy(g, j - 2 + r)));
        const Q4<T> m = bc_row<T>(lm, P[0], P[1], P[2], bcmap_quad(g, bcmap, i0, j - 1)), c = bc_row<T>(lm, P[1], P[2], P[3], bcmap_quad(g, bcmap, i0, j)),
                    n = bc_row<T>(lm, P[2], P[3], P[4], bcmap_quad(g, bcmap, i0, j + 1));
        o = plain_row<T>(lm, m, c, n, S2, S3);
    }
    if (lm.owner && sel) store_quad_sel<T>(pn + idx<1, T>(g, 0, i0, j), o.quad(), sel);
}

// One row of the FIRST of two sweeps, general case: raw rows r-2 .. r+2 of the input iterate -> K7 on rows r-1 .. r+1 -> the plain stencil;
// cells the sweep does not compute take the intermediate buffer's wall content.  (The workgroup form of the general path: a listed row is
// computed by the four waves of a workgroup - its five first-sweep rows side by side, exchanged through LDS - instead of one wave walking
// the nine-step pipeline above: one or two memory round trips instead of nine.  That pipeline was the tail of the launch.)
template <bool SW, typename T>
__device__ __forceinline__ Q4<T> first_sweep_row(const Grid &g, const LaneMap &lm, int i0, int r, const uint8_t *bcmap, const T *pn, const T *pc, const T *src)
{
    Q4<T> A[5], B0[3];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int row = clampy(g, r - 2 + u);
        A[u] = Q4<T>(load_quad<1>(pc, g, 0, i0, row));
        if (SW) {            // the input iterate's wall cells live in the other buffer
            const Q4<T> W(load_quad<1>(pn, g, 0, i0, row));
            const unsigned s = sel_not_wall(mask_quad(g, i0, row));
#pragma unroll
            for (int q = 0; q < 4; ++q) A[u].a[q] = (s >> q) & 1u ? A[u].a[q] : W.a[q];
        }
    }
    const int row = clampy(g, r);
    const Q4<T> S2(load_quad<2>(src, g, 0, i0, row)), S3(load_quad<2>(src, g, 1, i0, row));
    const Q4<T> H(load_quad<1>(SW ? pc : pn, g, 0, i0, row));          // the intermediate iterate's wall cells
    const unsigned s = sel_not_wall(mask_quad(g, i0, row));
#pragma unroll
    for (int u = 0; u < 3; ++u) B0[u] = bc_row<T>(lm, A[u], A[u + 1], A[u + 2], bcmap_quad(g, bcmap, i0, r - 1 + u));
    const Q4<T> v = plain_row<T>(lm, B0[0], B0[1], B0[2], S2, S3);
    Q4<T> o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o.a[q] = (s >> q) & 1u ? v.a[q] : H.a[q];
    return o;
}

// list[k] = (wave column << 20) | row of every wave-tile row that k_jacobi_pair's general path must compute: a "hard" or "vertical" flag
// (bits 2, 3; list_a) or a "hard" flag (list_b) in one of rows r-2 .. r+2, and a not-wall owner cell in row r.  Built once per mask.
static __global__ __launch_bounds__(256) void k_pair_list(Grid g, int nwx, uint8_t *flags, uint32_t *list_a, uint32_t *list_b, unsigned *count)
{
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wx = wid % nwx, r = wid / nwx;
    if (r >= g.rows) return;
    const LaneMap lm = lane_map_wave(g, wx);
    unsigned f = 0u;
    for (int d = -2; d <= 2; ++d) f |= flags[(size_t)wx * g.rows + clampy(g, r + d)];
    const bool computed = __any(lm.owner && sel_not_wall(mask_quad(g, lm.i0, r)) != 0u);
    if (lane == 0 && (f & 12u)) {
        // bit 4: the general path owns this row when the main tiles know horizontal recipes only (HV = false); bit 5: when they also handle
        // the vertical ones (HV = true).  (The other waves read bits 2, 3 of this byte only.)
        flags[(size_t)wx * g.rows + r] |= (f & 4u) ? 48u : 16u;
        if (computed) {
            list_a[atomicAdd(count, 1u)] = ((uint32_t)wx << 20) | (uint32_t)r;
            if (f & 4u) list_b[atomicAdd(count + 1, 1u)] = ((uint32_t)wx << 20) | (uint32_t)r;
        }
    }
}

// TWO sweeps in one pass ("temporal blocking"): the first sweep's rows j0-2 .. j0+RT+1 stay in registers (x-neighbours through the halo
// lanes, whose 4 cells are exactly the reach of two lazily-bounded sweeps), the second sweep's RT rows are stored.  Half the launches and
// half the bytes of two k_jacobi_lazy passes; every cell goes through the same two predict_p evaluations, so the bits are the same.
// Per wave-tile row (flags, k_lazy_flags): bit 1 - a wall or a boundary-condition target within 2 columns of the wave's owner cells.  A tile
// with no such row among j0-2 .. j0+RT+1 runs the plain two-sweep path; the others apply K7 row by row in registers, which is exact for
// the targets that are 0 or copy their left / right neighbour (inflow / outflow columns, vertical wall faces).  Output rows with anything
// else within 2 rows and 4 columns (bits 2, 3: targets with a source in another row - floors, ceilings, corners -, wall cells whose content
// is history) are left out here and computed, one row per workgroup, by the first `zoff` z slices of the same launch from
// `list` (k_pair_list): general, register-frugal, dispatched first.  HV = true adds a third tile path that also applies the
// recipes with a source in the row below / above (walls thicker than one cell) inside the plain footprint, leaving only bit-2 rows to the
// general path: for masks whose outlines are staircases (bc3's cylinders: 31 % -> 2 % general rows, +16 .. 28 %); on masks with few such
// rows the extra code costs 6 % (bc2 res 1600), so the host picks per mask.
// The two physical buffers differ in the wall cells NOTHING ever writes (no K7 assignment: e.g. the frame cells beside an inflow column),
// and not-wall cells next to them read them.  The reference's rotation keeps every even iterate in one buffer and every odd one in the
// other; a sequence of passes pc -> pn -> pc ... puts every second pass the other way round.  SW = false: the input iterate's wall cells
// are pc's, the intermediate's are pn's.  SW = true: the input's are pn's, the intermediate's are pc's.
template <int RT, bool SW, bool HV, typename T>
__global__ __launch_bounds__(256) void k_jacobi_pair(Grid g, int nbx, int nby, int jb, int je, const uint8_t *bcmap, const uint8_t *flags,
                                                     const uint32_t *list, int nlist, int zoff, T *pn, const T *pc, const T *src)
{
    if ((int)blockIdx.z < zoff) {        // the general rows: one listed row per WORKGROUP
        __shared__ __attribute__((aligned(32))) T s1[5][64][4];                  // first-sweep rows j-2 .. j+2 of the wave column
        const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
        const int k = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
        if (k >= nlist) return;
        const uint32_t e = list[k];
        const int wx = (int)(e >> 20), j = (int)(e & 0xfffffu);
        if (j < jb || j >= je) return;                                           // (workgroup-uniform, like everything up to the barrier)
        const LaneMap lm = lane_map_wave(g, wx);                                 // all four waves: the lanes of wave column wx
        // wave 0: rows j-2 and j-1, waves 1 .. 3: rows j, j+1, j+2
#pragma unroll 1
        for (int t = 0; t < (w == 0 ? 2 : 1); ++t) {
            const int sl = w == 0 ? t : w + 1;
            const Q4<T> row = first_sweep_row<SW, T>(g, lm, lm.i0, j - 2 + sl, bcmap, pn, pc, src);
            *reinterpret_cast<typename Quad<T>::type *>(&s1[sl][lane][0]) = row.quad();
        }
        __syncthreads();
        if (w != 0) return;
        Q4<T> S1[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) S1[u] = Q4<T>(*reinterpret_cast<const typename Quad<T>::type *>(&s1[u][lane][0]));
        const Q4<T> m = bc_row<T>(lm, S1[0], S1[1], S1[2], bcmap_quad(g, bcmap, lm.i0, j - 1)), c = bc_row<T>(lm, S1[1], S1[2], S1[3], bcmap_quad(g, bcmap, lm.i0, j)),
                    n = bc_row<T>(lm, S1[2], S1[3], S1[4], bcmap_quad(g, bcmap, lm.i0, j + 1));
        const Q4<T> S2(load_quad<2>(src, g, 0, lm.i0, j)), S3(load_quad<2>(src, g, 1, lm.i0, j));
        const Q4<T> out = plain_row<T>(lm, m, c, n, S2, S3);
        const unsigned s_out = sel_not_wall(mask_quad(g, lm.i0, j));
        if (lm.owner && s_out) store_quad_sel<T>(pn + idx<1, T>(g, 0, lm.i0, j), out.quad(), s_out);
        return;
    }
    int bx, by, cg;
    if (!tile_coords<1>(g, nbx, nby, jb, je, RT, bx, by, cg, zoff)) return;   // bx: wave column, by: tile row
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0, j0 = jb + by * RT;
    unsigned F[RT + 4], sel[RT], any = 0u, dirty = 0u;           // flags of rows j0-2 .. j0+RT+1
#pragma unroll
    for (int r = 0; r < RT + 4; ++r) { F[r] = flags[(size_t)bx * g.rows + clampy(g, j0 - 2 + r)]; dirty |= F[r]; }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        sel[r] = j0 + r < je && !(F[r + 2] & (HV ? 32u : 16u)) ? sel_not_wall(mask_quad(g, i0, j0 + r)) : 0u;      // bit 4 / 5: the general path's row
        any |= sel[r];
    }
    if (!__any(any != 0u)) return;
    Q4<T> S2[RT + 4], S3[RT + 4], o[RT];                         // source pair of rows j0-2 .. j0+RT+1 ([1 .. RT+2] are used)
    if (!(dirty & 2u)) {
        Q4<T> A[RT + 4], S1[RT + 2];                             // pc rows j0-2 .. j0+RT+1; first sweep rows j0-1 .. j0+RT
#pragma unroll
        for (int r = 0; r < RT + 4; ++r) A[r] = Q4<T>(load_quad<1>(pc, g, 0, i0, clampy(g, j0 - 2 + r)));
#pragma unroll
        for (int r = 1; r < RT + 3; ++r) {
            S2[r] = Q4<T>(load_quad<2>(src, g, 0, i0, clampy(g, j0 - 2 + r)));
            S3[r] = Q4<T>(load_quad<2>(src, g, 1, i0, clampy(g, j0 - 2 + r)));
        }
#pragma unroll
        for (int r = 0; r < RT + 2; ++r) S1[r] = plain_row<T>(lm, A[r], A[r + 1], A[r + 2], S2[r + 1], S3[r + 1]);
#pragma unroll
        for (int r = 0; r < RT; ++r) o[r] = plain_row<T>(lm, S1[r], S1[r + 1], S1[r + 2], S2[r + 2], S3[r + 2]);
    } else if (HV && (dirty & 8u)) {
        // targets that copy from (or average with) the row below / above - floors, ceilings, corners - in walls thicker than one cell:
        // whoever reads such a target sits on its source's side, so no row beyond the plain path's footprint is needed (a row missing at
        // the edge of the window is stood in for by the row itself; what that produces is read by nobody, or only by rows within 2 of a
        // bit-2 cell, which the general path owns)
        Q4<T> A[RT + 4], B0[RT + 4], S1[RT + 2], B1[RT + 2];
        uint32_t C[RT + 4];
#pragma unroll
        for (int r = 0; r < RT + 4; ++r) {
            A[r] = Q4<T>(load_quad<1>(pc, g, 0, i0, clampy(g, j0 - 2 + r)));
            C[r] = bcmap_quad(g, bcmap, i0, j0 - 2 + r);
        }
#pragma unroll
        for (int r = 1; r < RT + 3; ++r) {
            S2[r] = Q4<T>(load_quad<2>(src, g, 0, i0, clampy(g, j0 - 2 + r)));
            S3[r] = Q4<T>(load_quad<2>(src, g, 1, i0, clampy(g, j0 - 2 + r)));
        }
#pragma unroll
        for (int r = 0; r < RT + 4; ++r) B0[r] = bc_row<T>(lm, A[r == 0 ? 0 : r - 1], A[r], A[r == RT + 3 ? r : r + 1], C[r]);
#pragma unroll
        for (int r = 0; r < RT + 2; ++r) S1[r] = plain_row<T>(lm, B0[r], B0[r + 1], B0[r + 2], S2[r + 1], S3[r + 1]);
#pragma unroll
        for (int r = 0; r < RT + 2; ++r) B1[r] = bc_row<T>(lm, S1[r == 0 ? 0 : r - 1], S1[r], S1[r == RT + 1 ? r : r + 1], C[r + 1]);
#pragma unroll
        for (int r = 0; r < RT; ++r) o[r] = plain_row<T>(lm, B1[r], B1[r + 1], B1[r + 2], S2[r + 2], S3[r + 2]);
    } else {
        // K7 stays inside each row, and what the first sweep leaves in the wall cells it does not compute is never read: the footprint of
        // the plain path plus one cheap pass per register row.  (Rows of this tile that the general path owns come out wrong here and
        // are not stored; the rows that are stored depend on rows without bits 2, 3 only.)
        Q4<T> A[RT + 4], S1[RT + 2];
        uint32_t C[RT + 4];
#pragma unroll
        for (int r = 0; r < RT + 4; ++r) {
            A[r] = Q4<T>(load_quad<1>(pc, g, 0, i0, clampy(g, j0 - 2 + r)));
            C[r] = bcmap_quad(g, bcmap, i0, j0 - 2 + r);
        }
#pragma unroll
        for (int r = 1; r < RT + 3; ++r) {
            S2[r] = Q4<T>(load_quad<2>(src, g, 0, i0, clampy(g, j0 - 2 + r)));
            S3[r] = Q4<T>(load_quad<2>(src, g, 1, i0, clampy(g, j0 - 2 + r)));
        }
#pragma unroll
        for (int r = 0; r < RT + 4; ++r) A[r] = bc_row_h<T>(lm, A[r], C[r]);
#pragma unroll
        for (int r = 0; r < RT + 2; ++r) S1[r] = bc_row_h<T>(lm, plain_row<T>(lm, A[r], A[r + 1], A[r + 2], S2[r + 1], S3[r + 1]), C[r + 1]);
#pragma unroll
        for (int r = 0; r < RT; ++r) o[r] = plain_row<T>(lm, S1[r], S1[r + 1], S1[r + 2], S2[r + 2], S3[r + 2]);
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (lm.owner && sel[r]) store_quad_sel<T>(pn + idx<1, T>(g, 0, i0, j0 + r), o[r].quad(), sel[r]);
}


// ------------------------------------------------------------------------------------------------
// limit_field (fs/solver.py:38-43), quad form: one 16-byte load per channel and lane, one row per blockIdx.y, nothing else in
// flight - the shape that streams fastest on this chip (tools/membw.hip: 6.5 TB/s read-only).  Cells are rewritten only
// where the norm exceeds the limit (never in a healthy run), so the pass is read-only in practice.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_limit_quad(Grid g, int jb, int je, T lim, T *v, unsigned *hot, int gated)
{
    // `gated`: the buffer's flag is authoritative for this limit (fs_device.h) - while it is down no cell can exceed it: done.
    // A fixed grid of gridDim.y row lanes keeps the common case a ~3 us launch; the rare full pass strides over the rows.
    if (gated && (hot[0] | hot[1] | hot[2] | hot[3]) == 0u) return;      // ([1], [2]: raised by the op lists of k_velocity_bc_limit launches)
    const int i0 = (blockIdx.x * 256 + threadIdx.x) << 2;
    if (i0 >= g.X) return;
    for (int j = jb + blockIdx.y; j < je; j += gridDim.y) {
        T *px = v + idx<2, T>(g, 0, i0, j), *py = v + idx<2, T>(g, 1, i0, j);
        const Q4<T> X(*reinterpret_cast<const typename Quad<T>::type *>(px)), Y(*reinterpret_cast<const typename Quad<T>::type *>(py));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + q >= g.X) break;        // (odd res: the last quad of a row ends in the row's padding)
            const T x = X.a[q], y = Y.a[q];
            const T nrm = tsqrt(x * x + y * y);
            if (nrm > lim) {
                px[q] = lim * (x / nrm);
                py[q] = lim * (y / nrm);
            }
            raise_hot(hot, hot2(x, y));      // (ungated calls - a limit of 8 or less - keep the flag's meaning intact)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// limit_field of step n and set_velocity_boundary_condition of step n + 1 in ONE launch (round 4; fs/solver.py:38-43 + :79-81,
// fs/boundary_condition.py:16-39).  The limit pass is the last kernel of a step and, behind its flag, does nothing in a healthy run - yet
// on small grids its launch is a fifth of the step (BASELINE configs[0]: 4 launches of ~6 us).  The host defers it (fs/runtime.py: the
// velocity field carries a pending limit until somebody looks at it) and the boundary kernel of the next step starts with the gate:
//   flag down (always, in a healthy run): two scalar loads, then the op list as in k_velocity_bc;
//   flag up: the workgroups of this launch - the op list is O(perimeter): a few hundred at most, all resident - share the rows of the
//   limit pass among them, meet at a grid barrier (arrive / depart counters, agent-scope fences: MI355X_MICROARCH.md "barrier-counter"),
//   and run the op list on limited values.  Same arithmetic as k_limit_quad, cell by cell.
// ------------------------------------------------------------------------------------------------
// The gate must read the SAME value in every workgroup (those that see the flag up wait for all the others at the barrier), and the op list
// itself may raise the flag (an inflow constant above the limit) while later workgroups have not started yet.  So the flag has three sticky
// words: [0] what every other kernel raises, [1] / [2] what the op list of a launch with parity 0 / 1 raises; consecutive launches on one
// buffer alternate the parity (the caller's contract, fs_hip.h), and a launch reads [0] and the word of the OTHER parity - the one its
// predecessor raised, which nobody writes now.  A word seen up is folded into [0] behind the barrier, where every workgroup has read the gate:
// nothing raised is ever lost, and no launch waits for a "last workgroup out" any more (the exit ticket of the first version of this
// kernel: two dependent atomics at the end of every launch, 0.8 us of a 22 us step).
// the limit pass of a gate that is up, shared by the workgroups of the launch, and the grid barrier behind it (all of them resident)
template <typename T>
__device__ __forceinline__ void limit_pass_and_barrier(const Grid &g, int lb, int le, T lim, T *v, unsigned *hot, unsigned *sync)
{
    for (int j = lb + (int)blockIdx.x; j < le; j += (int)gridDim.x)
        for (int i0 = (int)threadIdx.x << 2; i0 < g.X; i0 += 1024) {
            T *px = v + idx<2, T>(g, 0, i0, j), *py = v + idx<2, T>(g, 1, i0, j);
            const Q4<T> X(*reinterpret_cast<const typename Quad<T>::type *>(px)), Y(*reinterpret_cast<const typename Quad<T>::type *>(py));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (i0 + q >= g.X) break;            // (X = 2 res: a multiple of 4 for even res; at odd res the last quad of a row holds 2 cells + 2 of the row's padding)
                const T x = X.a[q], y = Y.a[q];
                const T nrm = tsqrt(x * x + y * y);
                if (nrm > lim) {
                    px[q] = lim * (x / nrm);
                    py[q] = lim * (y / nrm);
                }
            }
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                                     // my rows are out (agent scope) before I arrive
        atomicAdd(&sync[0], 1u);
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(4);
        __threadfence();                                                     // ... and this CU's L1 forgets what it held of the others' rows
        if (atomicAdd(&sync[1], 1u) == gridDim.x - 1u) {                     // the last one out resets both counters: everybody has left the spin
            __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x == 0) atomicOr(hot, 1u);                              // (every workgroup has read the gate: the fold cannot split it)
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void k_velocity_bc_limit(Grid g, BcOps ops, int jb, int je, int lb, int le, T lim, T *v, const T *bc_const, unsigned *hot, unsigned *sync, int parity)
{
    if ((hot[0] | hot[2 - parity] | hot[3]) != 0u) limit_pass_and_barrier<T>(g, lb, le, lim, v, hot, sync);      // ([3]: nobody raises it in this launch either)
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned *raise = hot + 1 + parity;
    if (n < ops.nsimple) {
        const int4 o = ops.simple[n];
        const int trow = o.w >> 2;
        // a simple mirror / outflow op reads a cell of the same row or of the row +-2 / +-1 next to it: its row travels in .z
        if (trow >= jb && trow < je) velocity_bc_op(g, o.w & 3, o.x, trow, o.y, o.z, v, bc_const, raise);
    } else if (n - ops.nsimple < ops.npair) {
        n -= ops.nsimple;
        velocity_bc_pair(g, ops.pair[2 * n], ops.pair[2 * n + 1], jb, je, v, bc_const, raise);
    } else {
        n -= ops.nsimple + ops.npair;
        if (n < ops.ncomp && !(ops.comp_rhi[n] < jb || ops.comp_rlo[n] >= je))
            for (int o = ops.comp_begin[n]; o < ops.comp_begin[n + 1]; ++o)
                velocity_bc_op(g, ops.kind[o], ops.tgt[o], ops.row[o], ops.s1[o], ops.srow[o], v, bc_const, raise);
    }
}

// The same for the dye solvers (fs/solver.py:148-155, 385-392: limit_field(v) ends the flow step, set_dye_boundary_condition starts the dye step):
// the velocity's limit pass rides with the dye boundary kernel - whose op list does not touch the velocity, so the gate reads all three words.
template <typename T>
__global__ __launch_bounds__(256) void k_dye_bc_limit(Grid g, BcOps ops, int jb, int je, int lb, int le, T lim, T *v, unsigned *hot, unsigned *sync, T *dye, const T *bc_dye)
{
    if ((hot[0] | hot[1] | hot[2] | hot[3]) != 0u) limit_pass_and_barrier<T>(g, lb, le, lim, v, hot, sync);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= ops.nsimple) return;
    const int4 o = ops.simple[n];
    const int trow = o.w >> 2;
    if (trow < jb || trow >= je) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) dye[cell_off(g, o.x, trow, 3, c)] = bc_dye[cell_off(g, o.x, trow, 3, c)];
}

// clamp_field restricted to the inflow cells (op list of the dye boundary kernel): with the clamp folded into the advection
// store these are the only other cells of the dye buffer whose value can lie outside [low, high] (the dye BC rewrites them
// with the scene colour every step).
template <typename T>
__global__ __launch_bounds__(256) void k_clamp_inflow(Grid g, BcOps ops, int jb, int je, T lo, T hi, T *dye)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= ops.nsimple) return;
    const int4 o = ops.simple[n];
    const int trow = o.w >> 2;
    if (trow < jb || trow >= je) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t a = cell_off(g, o.x, trow, 3, c);
        dye[a] = tmin(tmax(dye[a], lo), hi);
    }
}


// ------------------------------------------------------------------------------------------------
// K4 for the 3-channel dye (fs/solver.py:385-401 -> _advection_phase): one pass over the three channels.  The advecting
// velocity (three rows of v.x, v.y), its central differences and the upwind selectors are loaded / formed once per cell and
// kept in registers; the channels are then streamed one after the other (nine 16-byte loads each), re-using the same
// registers.  Replaces three single-channel launches slices that each re-read the velocity.
// ------------------------------------------------------------------------------------------------
template <int DM, bool CLAMP01, typename T>
__device__ __forceinline__ void cip_advect_dye_tile(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fn, T *fxn, T *fyn,
                                                        const T *fc, const T *fxc, const T *fyc, const T *v, unsigned *hot)
{
    int bx, by;
    if (!tile_coords(g, nbx, nby, jb, je, 1, bx, by)) return;   // bx: wave column, by: tile row
    const LaneMap lm = lane_map_wave(g, bx);
    const int i0 = lm.i0, j = jb + by;
    const unsigned fl = sel_fluid(mask_quad(g, i0, j));
    if (!__any(fl != 0u)) return;
    const int jm = clampy(g, j - 1), jp = clampy(g, j + 1);

    T vx[4], vy[4], dxx[4], dxy[4], dyx[4], dyy[4];
    {
        const Q4<T> X0(load_quad<2>(v, g, 0, i0, jm)), X1(load_quad<2>(v, g, 0, i0, j)), X2(load_quad<2>(v, g, 0, i0, jp));
        const Q4<T> Y0(load_quad<2>(v, g, 1, i0, jm)), Y1(load_quad<2>(v, g, 1, i0, j)), Y2(load_quad<2>(v, g, 1, i0, jp));
        const T xl = quad_left<T>(lm, X1.quad()), xr = quad_right<T>(lm, X1.quad());
        const T yl = quad_left<T>(lm, Y1.quad()), yr = quad_right<T>(lm, Y1.quad());
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vx[q] = X1.a[q]; vy[q] = Y1.a[q];
            const T xE = q == 3 ? xr : X1.a[q == 3 ? 3 : q + 1], xW = q == 0 ? xl : X1.a[q == 0 ? 0 : q - 1];
            const T yE = q == 3 ? yr : Y1.a[q == 3 ? 3 : q + 1], yW = q == 0 ? yl : Y1.a[q == 0 ? 0 : q - 1];
            dxx[q] = xdiv<DM>((T)0.5 * (xE - xW), k.dx, k.inv_dx, k.r_dx); dxy[q] = xdiv<DM>((T)0.5 * (yE - yW), k.dx, k.inv_dx, k.r_dx);
            dyx[q] = xdiv<DM>((T)0.5 * (X2.a[q] - X0.a[q]), k.dx, k.inv_dx, k.r_dx); dyy[q] = xdiv<DM>((T)0.5 * (Y2.a[q] - Y0.a[q]), k.dx, k.inv_dx, k.r_dx);
        }
    }
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        Q4<T> F[3], FX[3], FY[3];
        F[0] = Q4<T>(load_quad<3>(fc, g, c, i0, jm)); F[1] = Q4<T>(load_quad<3>(fc, g, c, i0, j)); F[2] = Q4<T>(load_quad<3>(fc, g, c, i0, jp));
        FX[0] = Q4<T>(load_quad<3>(fxc, g, c, i0, jm)); FX[1] = Q4<T>(load_quad<3>(fxc, g, c, i0, j)); FX[2] = Q4<T>(load_quad<3>(fxc, g, c, i0, jp));
        FY[0] = Q4<T>(load_quad<3>(fyc, g, c, i0, jm)); FY[1] = Q4<T>(load_quad<3>(fyc, g, c, i0, j)); FY[2] = Q4<T>(load_quad<3>(fyc, g, c, i0, jp));
        T fl_[3], fr_[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { fl_[r] = quad_left<T>(lm, F[r].quad()); fr_[r] = quad_right<T>(lm, F[r].quad()); }
        const T fxl = quad_left<T>(lm, FX[1].quad()), fxr = quad_right<T>(lm, FX[1].quad());
        const T fyl = quad_left<T>(lm, FY[1].quad()), fyr = quad_right<T>(lm, FY[1].quad());
        Q4<T> OF, OFX, OFY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool nx = vx[q] < (T)0.0, ny = vy[q] < (T)0.0;
            const T fE1 = q == 3 ? fr_[1] : F[1].a[q == 3 ? 3 : q + 1], fW1 = q == 0 ? fl_[1] : F[1].a[q == 0 ? 0 : q - 1];
            const T fE0 = q == 3 ? fr_[0] : F[0].a[q == 3 ? 3 : q + 1], fW0 = q == 0 ? fl_[0] : F[0].a[q == 0 ? 0 : q - 1];
            const T fE2 = q == 3 ? fr_[2] : F[2].a[q == 3 ? 3 : q + 1], fW2 = q == 0 ? fl_[2] : F[2].a[q == 0 ? 0 : q - 1];
            const T fxE = q == 3 ? fxr : FX[1].a[q == 3 ? 3 : q + 1], fxW = q == 0 ? fxl : FX[1].a[q == 0 ? 0 : q - 1];
            const T fyE = q == 3 ? fyr : FY[1].a[q == 3 ? 3 : q + 1], fyW = q == 0 ? fyl : FY[1].a[q == 0 ? 0 : q - 1];
            const T f00 = F[1].a[q], f0m = ny ? F[2].a[q] : F[0].a[q], fm0 = nx ? fE1 : fW1;
            const T fmm = ny ? (nx ? fE2 : fW2) : (nx ? fE0 : fW0);
            const T fx00 = FX[1].a[q], fxm0 = nx ? fxE : fxW, fx0m = ny ? FX[2].a[q] : FX[0].a[q];
            const T fy00 = FY[1].a[q], fy0m = ny ? FY[2].a[q] : FY[0].a[q], fym0 = nx ? fyE : fyW;
            cip_point<DM>(k, vx[q], vy[q], dxx[q], dxy[q], dyx[q], dyy[q], f00, f0m, fm0, fmm, fx00, fxm0, fx0m, fy00, fy0m, fym0,
                          OF.a[q], OFX.a[q], OFY.a[q]);
            if (CLAMP01) OF.a[q] = tmin(tmax(OF.a[q], (T)0.0), (T)1.0);     // clamp_field(dye, 0, 1), fs/solver.py:46-49
        }
        if (lm.owner && fl) {
            store_quad_sel<T>(fn + idx<3, T>(g, c, i0, j), OF.quad(), fl);
            store_quad_sel<T>(fxn + idx<3, T>(g, c, i0, j), OFX.quad(), fl);
            store_quad_sel<T>(fyn + idx<3, T>(g, c, i0, j), OFY.quad(), fl);
        }
    }
}
template <int DM, bool CLAMP01, typename T>
__global__ __launch_bounds__(256) void k_cip_advect_dye(Grid g, Konst<T> k, int nbx, int nby, int jb, int je, T *fn, T *fxn, T *fyn,
                                                        const T *fc, const T *fxc, const T *fyc, const T *v, unsigned *hot)
{
    cip_advect_dye_tile<DM, CLAMP01, T>(g, k, nbx, nby, jb, je, fn, fxn, fyn, fc, fxc, fyc, v, hot);
}


}  // namespace fs
