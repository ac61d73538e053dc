// fs_kernels.h - the step() kernels, one cell per lane, lanes along x (coalesced rows).
//
// Every kernel covers local rows [jb, je) x columns [0, X): blockIdx.y = row, blockIdx.x*256 + tid = x.
// Masked kernels leave the cells they do not own untouched (the reference's DoubleBuffer "stale cell"
// behaviour, SURVEY.md H5, is part of the algorithm).  Reference citations are per kernel.
#pragma once
#include "fs_device.h"

namespace fs {

#define FS_CELL_PROLOGUE                                   \
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   \
    const int j = jb + blockIdx.y;                         \
    if (i >= g.X) return;

// ------------------------------------------------------------------------------------------------
// advection terms (fs/advection.py)
// ------------------------------------------------------------------------------------------------
// advect_upwind, fs/advection.py:12-24
template <int C, typename T>
__device__ __forceinline__ T adv_upwind(const T *phi, const Grid &g, const Konst<T> &k, T ux, T uy, int c, int i, int j)
{
    int kx = ux < (T)0.0 ? i : i - 1;
    T a = ux * fdiff_x<C>(phi, g, k, c, kx, j);
    int ky = uy < (T)0.0 ? j : j - 1;
    T b = uy * fdiff_y<C>(phi, g, k, c, i, ky);
    return a + b;
}

// advect_kk_scheme, fs/advection.py:27-60: 5-point one-sided stencils, coefficients dotted left to right
template <int C, typename T>
__device__ __forceinline__ T adv_kk(const T *phi, const Grid &g, const Konst<T> &k, T ux, T uy, int c, int i, int j)
{
    // u < 0: coef = [-2, 10, -9, 2, -1];  u >= 0: -coef[::-1] = [1, -2, 9, -10, 2]   (over phi[+2], [+1], [0], [-1], [-2])
    const bool nx = ux < (T)0;
    T w0 = nx ? (T)-2 : (T)1, w1 = nx ? (T)10 : (T)-2, w2 = nx ? (T)-9 : (T)9, w3 = nx ? (T)2 : (T)-10, w4 = nx ? (T)-1 : (T)2;
    T acc = smp<C>(phi, g, c, i + 2, j) * w0;
    acc = acc + smp<C>(phi, g, c, i + 1, j) * w1;
    acc = acc + smp<C>(phi, g, c, i, j) * w2;
    acc = acc + smp<C>(phi, g, c, i - 1, j) * w3;
    acc = acc + smp<C>(phi, g, c, i - 2, j) * w4;
    T a = acc / k.six_dx;
    const bool ny = uy < (T)0;
    w0 = ny ? (T)-2 : (T)1; w1 = ny ? (T)10 : (T)-2; w2 = ny ? (T)-9 : (T)9; w3 = ny ? (T)2 : (T)-10; w4 = ny ? (T)-1 : (T)2;
    acc = smp<C>(phi, g, c, i, j + 2) * w0;
    acc = acc + smp<C>(phi, g, c, i, j + 1) * w1;
    acc = acc + smp<C>(phi, g, c, i, j) * w2;
    acc = acc + smp<C>(phi, g, c, i, j - 1) * w3;
    acc = acc + smp<C>(phi, g, c, i, j - 2) * w4;
    T b = acc / k.six_dx;
    return ux * a + uy * b;
}

template <int SCHEME, int C, typename T>
__device__ __forceinline__ T advect(const T *phi, const Grid &g, const Konst<T> &k, T ux, T uy, int c, int i, int j)
{
    if (SCHEME == 0) return adv_upwind<C>(phi, g, k, ux, uy, c, i, j);
    return adv_kk<C>(phi, g, k, ux, uy, c, i, j);
}

// ------------------------------------------------------------------------------------------------
// K2'  MacSolver._update_velocities, fs/solver.py:94-107  (fluid cells)
// ------------------------------------------------------------------------------------------------
template <int SCHEME, typename T>
__global__ __launch_bounds__(256) void k_mac_update(Grid g, Konst<T> k, int jb, T *vn, const T *vc, const T *pc, unsigned *hot)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) != 0) return;
    const T ux = at<2>(vc, g, 0, i, j), uy = at<2>(vc, g, 1, i, j);
    const T gp[2] = {diff_x<1>(pc, g, k, 0, i, j), diff_y<1>(pc, g, k, 0, i, j)};
    T o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        T a = advect<SCHEME, 2>(vc, g, k, ux, uy, c, i, j);
        T lap = (diff2_x<2>(vc, g, k, c, i, j) + diff2_y<2>(vc, g, k, c, i, j)) / k.re;
        o[c] = (c == 0 ? ux : uy) + k.dt * (((-a) - gp[c]) + lap);
        vn[idx<2, T>(g, c, i, j)] = o[c];
    }
    raise_hot(hot, hot2(o[0], o[1]));
}

// K11  DyeMacSolver._update_dye, fs/solver.py:157-161
template <int SCHEME, typename T>
__global__ __launch_bounds__(256) void k_mac_dye(Grid g, Konst<T> k, int jb, T *dn, const T *dc, const T *vc)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) != 0) return;
    const T ux = at<2>(vc, g, 0, i, j), uy = at<2>(vc, g, 1, i, j);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        dn[idx<3, T>(g, c, i, j)] = at<3>(dc, g, c, i, j) - k.dt * advect<SCHEME, 3>(dc, g, k, ux, uy, c, i, j);
}

// ------------------------------------------------------------------------------------------------
// K0  CipMacSolver._set_grad, fs/solver.py:207-211  (all cells)
// ------------------------------------------------------------------------------------------------
template <int C, typename T>
__global__ __launch_bounds__(256) void k_cip_set_grad(Grid g, Konst<T> k, int jb, T *fx, T *fy, const T *f)
{
    FS_CELL_PROLOGUE
#pragma unroll
    for (int c = 0; c < C; ++c) {
        fx[idx<C, T>(g, c, i, j)] = diff_x<C>(f, g, k, c, i, j);
        fy[idx<C, T>(g, c, i, j)] = diff_y<C>(f, g, k, c, i, j);
    }
}

// K2  CipMacSolver._non_advection_phase (+ _calc_diffusion), fs/solver.py:229-240, 263-265  (not-wall cells)
template <bool P2, typename T>
__global__ __launch_bounds__(256) void k_cip_nonadv(Grid g, Konst<T> k, int jb, T *fn, const T *fc, const T *pc, unsigned *hot)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) == 1) return;
    const T gp[2] = {diff_x<1, P2>(pc, g, k, 0, i, j), diff_y<1, P2>(pc, g, k, 0, i, j)};
    T o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        T dif = (diff2_x<2, P2>(fc, g, k, c, i, j) + diff2_y<2, P2>(fc, g, k, c, i, j)) / k.re;
        T gg = (-gp[c]) + dif;
        o[c] = at<2>(fc, g, c, i, j) + gg * k.dt;
        fn[idx<2, T>(g, c, i, j)] = o[c];
    }
    raise_hot(hot, hot2(o[0], o[1]));
}

// K12  DyeCipMacSolver._non_advection_phase_dye, fs/solver.py:378-383
template <typename T>
__global__ __launch_bounds__(256) void k_cip_nonadv_dye(Grid g, Konst<T> k, int jb, T *dn, const T *dc)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) == 1) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        T dif = (diff2_x<3>(dc, g, k, c, i, j) + diff2_y<3>(dc, g, k, c, i, j)) / k.re;
        dn[idx<3, T>(g, c, i, j)] = at<3>(dc, g, c, i, j) + dif * k.dt;
    }
}

// K3  _non_advection_phase_grad, fs/solver.py:242-261  (not-wall cells).  The reference indexes fn/fc
// without sample(); its out-of-range reads at i = 0 / X-1 are defined here as clamped.
template <int C, bool P2, typename T>
__global__ __launch_bounds__(256) void k_cip_nonadv_grad(Grid g, Konst<T> k, int jb, T *fxn, T *fyn,
                                                         const T *fxc, const T *fyc, const T *fc, const T *fn)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) == 1) return;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T sx = ((smp<C>(fn, g, c, i + 1, j) - smp<C>(fc, g, c, i + 1, j)) - smp<C>(fn, g, c, i - 1, j)) + smp<C>(fc, g, c, i - 1, j);
        T sy = ((smp<C>(fn, g, c, i, j + 1) - smp<C>(fc, g, c, i, j + 1)) - smp<C>(fn, g, c, i, j - 1)) + smp<C>(fc, g, c, i, j - 1);
        fxn[idx<C, T>(g, c, i, j)] = at<C>(fxc, g, c, i, j) + qdiv<P2>(sx, k.two_dx, k.inv_two_dx);
        fyn[idx<C, T>(g, c, i, j)] = at<C>(fyc, g, c, i, j) + qdiv<P2>(sy, k.two_dx, k.inv_two_dx);
    }
}

// _cip_advect for one cell and one channel, fs/solver.py:282-332, on already-gathered values:
//   f00 = f[i,j], f0m = f[i,j_m], fm0 = f[i_m,j], fmm = f[i_m,j_m]  with the upwind cell (i_m, j_m) = (i - sign(u), j - sign(v));
//   likewise the x- / y-gradient fields; (vx, vy) the advecting velocity, d?? its central differences.
// V: one cell (float / double) or the two cells of a lane as one packed operand (v2f, fs_device.h): the same expression tree per element.
template <int DM, typename V, typename S>
__device__ __forceinline__ void cip_point(const Konst<S> &k, V vx, V vy, V dxx, V dxy, V dyx, V dyy,
                                          V f00, V f0m, V fm0, V fmm, V fx00, V fxm0, V fx0m, V fy00, V fy0m, V fym0,
                                          V &out_f, V &out_fx, V &out_fy)
{
#if defined(FS_CONTRACT_CIP)       // opt-in build flavour (EXTRA=-DFS_CONTRACT_CIP): FMA contraction inside the CIP polynomial - NOT bit-exact, never the default;
#pragma clang fp contract(fast)    // built once to price bit-exactness (tools/r5_contract.sh)
#endif
    using D = typename VecOf<V>::D;
    const V is = sel_neg(vx, (V)(S)-1, (V)(S)1);   // sign(0) = +1, fs/differentiation.py:12-14
    const V js = sel_neg(vy, (V)(S)-1, (V)(S)1);
    const V i_s_denom = is * k.dx3_fold, j_s_denom = js * k.dx3_fold, is_dx = is * k.dx;
    const V i_s_inv = is * k.inv_dx3_fold, j_s_inv = js * k.inv_dx3_fold, is_dx_inv = is * k.inv_dx;   // +-1 times the reciprocal: exact
    const D i_s_r = to_dbl(is) * k.r_dx3_fold, j_s_r = to_dbl(js) * k.r_dx3_fold, is_dx_r = to_dbl(is) * k.r_dx;   // likewise (1 / (-d) = -(1 / d))
    const V Xd = (-vx) * k.dt, Yd = (-vy) * k.dt;
    const V tmp1 = ((f00 - f0m) - fm0) + fmm;
    const V tmp2 = fm0 - f00;
    const V tmp3 = f0m - f00;
    V a, b, cc, d, gq;
    // (power-of-two dx: x * (+-1/d) is exact either way and the per-lane signed reciprocals are three register pairs - measured: no difference)
    constexpr bool SIGN_AFTER = (DM & DM_F64) != 0 && (DM & DM_P2) == 0;
    if constexpr (SIGN_AFTER) {
        // x / (+-d) through the f64 multiplication by the UNSIGNED reciprocal (a scalar register pair instead of a per-lane double per divisor: 12
        // VGPRs of the packed bodies) and the sign put on afterwards: (float)((double)x * -r) == -(float)((double)x * r), rounding is symmetric
        const V qa = xdiv<DM>((is * (fxm0 + fx00)) * k.dx - (S)2.0 * (-tmp2), k.dx3_fold, k.inv_dx3_fold, k.r_dx3_fold);
        const V qb = xdiv<DM>((js * (fy0m + fy00)) * k.dx - (S)2.0 * (-tmp3), k.dx3_fold, k.inv_dx3_fold, k.r_dx3_fold);
        const V qc = xdiv<DM>((-tmp1) - (is * (fx0m - fx00)) * k.dx, k.dx3_fold, k.inv_dx3_fold, k.r_dx3_fold);
        const V qd = xdiv<DM>((-tmp1) - (js * (fym0 - fy00)) * k.dx, k.dx3_fold, k.inv_dx3_fold, k.r_dx3_fold);
        a = sel_neg(vx, -qa, qa); b = sel_neg(vy, -qb, qb); cc = sel_neg(vy, -qc, qc); d = sel_neg(vx, -qd, qd);
        const V qg = xdiv<DM>((-(fym0 - fy00)) + cc * k.dx2_fold, k.dx, k.inv_dx, k.r_dx);
        gq = sel_neg(vx, -qg, qg);
    } else {
        a = xdiv<DM>((is * (fxm0 + fx00)) * k.dx - (S)2.0 * (-tmp2), i_s_denom, i_s_inv, i_s_r);
        b = xdiv<DM>((js * (fy0m + fy00)) * k.dx - (S)2.0 * (-tmp3), j_s_denom, j_s_inv, j_s_r);
        cc = xdiv<DM>((-tmp1) - (is * (fx0m - fx00)) * k.dx, j_s_denom, j_s_inv, j_s_r);
        d = xdiv<DM>((-tmp1) - (js * (fym0 - fy00)) * k.dx, i_s_denom, i_s_inv, i_s_r);
    }
    const V e = xdiv<DM>((S)3.0 * tmp2 + (is * (fxm0 + (S)2.0 * fx00)) * k.dx, k.dx2_fold, k.inv_dx2_fold, k.r_dx2_fold);
    const V f = xdiv<DM>((S)3.0 * tmp3 + (js * (fy0m + (S)2.0 * fy00)) * k.dx, k.dx2_fold, k.inv_dx2_fold, k.r_dx2_fold);
    if constexpr (!SIGN_AFTER) gq = xdiv<DM>((-(fym0 - fy00)) + cc * k.dx2_fold, is_dx, is_dx_inv, is_dx_r);
    out_f = (((((a * Xd + cc * Yd) + e) * Xd + gq * Yd) + fx00) * Xd + (((b * Yd + d * Xd) + f) * Yd + fy00) * Yd) + f00;
    const V Fx = ((((S)3.0 * a) * Xd + ((S)2.0 * cc) * Yd) + (S)2.0 * e) * Xd + (d * Yd + gq) * Yd + fx00;
    const V Fy = ((((S)3.0 * b) * Yd + ((S)2.0 * d) * Xd) + (S)2.0 * f) * Yd + (cc * Xd + gq) * Xd + fy00;
    out_fx = Fx - (k.dt * (Fx * dxx + Fy * dxy)) / (S)2.0;
    out_fy = Fy - (k.dt * (Fx * dyx + Fy * dyy)) / (S)2.0;
}

// K4  _advection_phase / _cip_advect, fs/solver.py:267-332  (fluid cells; C channels advected by v), one cell per lane
template <int C, bool P2, typename T>
__global__ __launch_bounds__(256) void k_cip_advect(Grid g, Konst<T> k, int jb, T *fn, T *fxn, T *fyn,
                                                    const T *fc, const T *fxc, const T *fyc, const T *v, unsigned *hot)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) != 0) return;
    const T vx = at<2>(v, g, 0, i, j), vy = at<2>(v, g, 1, i, j);
    const int im = i - (vx < (T)0.0 ? -1 : 1), jm = j - (vy < (T)0.0 ? -1 : 1);
    const T dxx = diff_x<2, P2>(v, g, k, 0, i, j), dxy = diff_x<2, P2>(v, g, k, 1, i, j);
    const T dyx = diff_y<2, P2>(v, g, k, 0, i, j), dyy = diff_y<2, P2>(v, g, k, 1, i, j);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        T of, ofx, ofy;
        cip_point<P2 ? DM_P2 : DM_IEEE>(k, vx, vy, dxx, dxy, dyx, dyy,
                  at<C>(fc, g, c, i, j), smp<C>(fc, g, c, i, jm), smp<C>(fc, g, c, im, j), smp<C>(fc, g, c, im, jm),
                  at<C>(fxc, g, c, i, j), smp<C>(fxc, g, c, im, j), smp<C>(fxc, g, c, i, jm),
                  at<C>(fyc, g, c, i, j), smp<C>(fyc, g, c, i, jm), smp<C>(fyc, g, c, im, j), of, ofx, ofy);
        fn[idx<C, T>(g, c, i, j)] = of;
        fxn[idx<C, T>(g, c, i, j)] = ofx;
        fyn[idx<C, T>(g, c, i, j)] = ofy;
        if (C == 2) raise_hot(hot, hot1(of));      // (velocity field; the dye passes hot = nullptr-free C == 3)
    }
}

// ------------------------------------------------------------------------------------------------
// K5  VorticityConfinement._calc_vorticity, fs/vorticity_confinement.py:27-32  (fluid cells)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_vort_calc(Grid g, Konst<T> k, int jb, T *vort, T *vort_abs, const T *vc)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) != 0) return;
    T w = diff_x<2>(vc, g, k, 1, i, j) - diff_y<2>(vc, g, k, 0, i, j);
    vort[idx<1, T>(g, 0, i, j)] = w;
    vort_abs[idx<1, T>(g, 0, i, j)] = tabs(w);
}

// K6  _add_vorticity + _vorticity_vec, fs/vorticity_confinement.py:34-55.  |grad| == 0 gives 0/0 = NaN and the
// NaN-ignoring min/max (fminf/fmaxf) turn it into +0.1 on both components.
template <typename T>
__global__ __launch_bounds__(256) void k_vort_add(Grid g, Konst<T> k, int jb, T *vn, const T *vc, const T *vort, const T *vort_abs, unsigned *hot)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) != 0) return;
    T gx = diff_x<1>(vort_abs, g, k, 0, i, j), gy = diff_y<1>(vort_abs, g, k, 0, i, j);
    T nrm = tsqrt(gx * gx + gy * gy);
    gx = gx / nrm;
    gy = gy / nrm;
    T w = at<1>(vort, g, 0, i, j);
    T f0 = gy * w, f1 = (-gx) * w;
    f0 = tmax(tmin(f0, (T)0.1), (T)-0.1);
    f1 = tmax(tmin(f1, (T)0.1), (T)-0.1);
    const T ox = at<2>(vc, g, 0, i, j) + k.dtw * f0, oy = at<2>(vc, g, 1, i, j) + k.dtw * f1;
    vn[idx<2, T>(g, 0, i, j)] = ox;
    vn[idx<2, T>(g, 1, i, j)] = oy;
    raise_hot(hot, hot2(ox, oy));
}

// ------------------------------------------------------------------------------------------------
// F1  predict_p, fs/pressure_updater.py:23-38:  (0.25*(pE + pW + pN + pS) + s2) - s3
//     s2 = (sxx^2 + syy^2 + syx*sxy) / 8,  s3 = dx*(sxx + syy) / (8*dt),  sub_x/sub_y = clamped velocity differences
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void poisson_source(const T *vc, const Grid &g, const Konst<T> &k, int i, int j, T &s2, T &s3)
{
    T sxx = smp<2>(vc, g, 0, i + 1, j) - smp<2>(vc, g, 0, i - 1, j);
    T sxy = smp<2>(vc, g, 1, i + 1, j) - smp<2>(vc, g, 1, i - 1, j);
    T syx = smp<2>(vc, g, 0, i, j + 1) - smp<2>(vc, g, 0, i, j - 1);
    T syy = smp<2>(vc, g, 1, i, j + 1) - smp<2>(vc, g, 1, i, j - 1);
    s2 = ((sxx * sxx + syy * syy) + (syx * sxy)) / (T)8.0;
    s3 = (k.dx * (sxx + syy)) / k.eight_dt;
}

template <typename T>
__device__ __forceinline__ T p_avg(const T *pc, const Grid &g, int i, int j)
{
    return (T)0.25 * (((smp<1>(pc, g, 0, i + 1, j) + smp<1>(pc, g, 0, i - 1, j)) + smp<1>(pc, g, 0, i, j + 1)) + smp<1>(pc, g, 0, i, j - 1));
}

// K8J  JacobiPressureUpdater._update, fs/pressure_updater.py:62-66  (not-wall cells)
template <bool SRC, typename T>
__global__ __launch_bounds__(256) void k_jacobi(Grid g, Konst<T> k, int jb, T *pn, const T *pc, const T *vs)
{
    FS_CELL_PROLOGUE
    if (mask_at(g, i, j) == 1) return;
    T s2, s3;
    if (SRC) { s2 = at<2>(vs, g, 0, i, j); s3 = at<2>(vs, g, 1, i, j); }
    else poisson_source(vs, g, k, i, j, s2, s3);
    pn[idx<1, T>(g, 0, i, j)] = (p_avg(pc, g, i, j) + s2) - s3;
}

// K8R  RedBlackSorPressureUpdater._update_pressures_odd/_even + _pn_ij, fs/pressure_updater.py:98-114.
// Lanes map to every second cell of the row so that a wave is fully active.  pn may alias pc (even pass).
template <bool SRC, typename T>
__global__ __launch_bounds__(256) void k_rbsor(Grid g, Konst<T> k, int jb, int parity, T *pn, const T *pc, const T *vs)
{
    const int j = jb + blockIdx.y;
    const int gy = g.ybase + j;
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x) + ((gy + parity) & 1);
    if (i >= g.X) return;
    if (mask_at(g, i, j) != 0) return;
    T s2, s3;
    if (SRC) { s2 = at<2>(vs, g, 0, i, j); s3 = at<2>(vs, g, 1, i, j); }
    else poisson_source(vs, g, k, i, j, s2, s3);
    T pred = (p_avg(pc, g, i, j) + s2) - s3;
    pn[idx<1, T>(g, 0, i, j)] = k.om1 * at<1>(pc, g, 0, i, j) + k.om * pred;
}

// source precompute (all cells of the row range; build-side optimisation, see fs_hip.h)
template <typename T>
__global__ __launch_bounds__(256) void k_poisson_source(Grid g, Konst<T> k, int jb, T *src, const T *vc)
{
    FS_CELL_PROLOGUE
    T s2, s3;
    poisson_source(vc, g, k, i, j, s2, s3);
    src[idx<2, T>(g, 0, i, j)] = s2;
    src[idx<2, T>(g, 1, i, j)] = s3;
}

// residual diagnostic: sum over not-wall cells of (predict_p(p) - p)^2 and their count.  Two deterministic stages, no
// atomics: a block owns RES_ROWS rows x 256 columns, every lane sums its column segment, the wave reduces with shuffles,
// the 4 waves meet in LDS and the block writes ONE partial pair; a single block then adds the partials in a fixed tree order.
// (One atomicAdd pair per wave - 524 k contended f64 atomics at res 4096 - took 8.7 ms; this form takes ~0.1 ms.)
constexpr int RES_ROWS = 8;
__device__ __forceinline__ void block_sum2(double &a, double &b, double *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds[2 * w] = a; lds[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = lds[0]; b = lds[1];
        for (int k = 1; k < nw; ++k) { a += lds[2 * k]; b += lds[2 * k + 1]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_residual(Grid g, Konst<T> k, int jb, int je, const T *p, const T *vc, double *partial)
{
    __shared__ double lds[8];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j0 = jb + blockIdx.y * RES_ROWS;
    double r2 = 0.0, n = 0.0;
    if (i < g.X)
        for (int j = j0; j < j0 + RES_ROWS && j < je; ++j)
            if (mask_at(g, i, j) != 1) {
                T s2, s3;
                poisson_source(vc, g, k, i, j, s2, s3);
                T r = ((p_avg(p, g, i, j) + s2) - s3) - at<1>(p, g, 0, i, j);
                r2 += (double)r * (double)r;
                n += 1.0;
            }
    block_sum2(r2, n, lds);
    if (threadIdx.x == 0) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * b] = r2;
        partial[2 * b + 1] = n;
    }
}
template <typename D>
__global__ __launch_bounds__(1024) void k_residual_final(const D *partial, int nblocks, D *acc)
{
    __shared__ double lds[32];
    double r2 = 0.0, n = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 1024) { r2 += partial[2 * b]; n += partial[2 * b + 1]; }
    block_sum2(r2, n, lds);
    if (threadIdx.x == 0) { acc[0] = r2; acc[1] = n; }
}

// ------------------------------------------------------------------------------------------------
// K9  limit_field, fs/solver.py:38-43 ;  K13  clamp_field, fs/solver.py:46-49   (all cells)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_limit(Grid g, int jb, T lim, T *v)
{
    FS_CELL_PROLOGUE
    T x = at<2>(v, g, 0, i, j), y = at<2>(v, g, 1, i, j);
    T nrm = tsqrt(x * x + y * y);
    if (nrm > lim) {
        v[idx<2, T>(g, 0, i, j)] = lim * (x / nrm);
        v[idx<2, T>(g, 1, i, j)] = lim * (y / nrm);
    }
}

// raise the flag of a 2-channel field for what is stored in rows [jb, je) (after an upload / a direct ghost-row transfer)
template <typename T>
__global__ __launch_bounds__(256) void k_scan_hot(Grid g, int jb, const T *v, unsigned *hot)
{
    FS_CELL_PROLOGUE
    raise_hot(hot, hot2(at<2>(v, g, 0, i, j), at<2>(v, g, 1, i, j)));
}

template <int C, typename T>
__global__ __launch_bounds__(256) void k_clamp(Grid g, int jb, T lo, T hi, T *f)
{
    FS_CELL_PROLOGUE
#pragma unroll
    for (int c = 0; c < C; ++c) {
        size_t o = idx<C, T>(g, c, i, j);
        f[o] = tmin(tmax(f[o], lo), hi);
    }
}

// ------------------------------------------------------------------------------------------------
// Visualisation kernels (fs/fluid_simulator.py:38-58, 121-126; colour maps fs/visualization.py:8-22), all cells:
//   MODE 0  _to_norm       rgb = 0.2 * (|v|, |v|, |v|);  rgb += 0.002 * (max(p, 0), 0, max(-p, 0))
//   MODE 1  _to_pressure   rgb = 0.04 * (max(p, 0), 0, max(-p, 0))
//   MODE 2  _to_vorticity  rgb = 0.005 * (max(w, 0), 0, max(-w, 0)),  w = diff_x(v).y - diff_y(v).x
//   MODE 3  _to_dye        rgb = dye
// then wall cells take the wall colour (0.5, 0.7, 0.5) (fs/fluid_simulator.py:17).  The scale factors are Python floats in
// the reference, i.e. constants of the field type.
// ------------------------------------------------------------------------------------------------
template <int MODE, typename T>
__global__ __launch_bounds__(256) void k_visualize(Grid g, Konst<T> k, int jb, T *rgb, const T *a, const T *b)
{
    FS_CELL_PROLOGUE
    T r, gg, bb;
    if (mask_at(g, i, j) == 1) { r = (T)0.5; gg = (T)0.7; bb = (T)0.5; }
    else if (MODE == 0) {
        const T x = at<2>(a, g, 0, i, j), y = at<2>(a, g, 1, i, j), pv = at<1>(b, g, 0, i, j);
        const T c = tsqrt(x * x + y * y);
        r = (T)0.2 * c + (T)0.002 * tmax(pv, (T)0.0);
        gg = (T)0.2 * c + (T)0.002 * (T)0.0;
        bb = (T)0.2 * c + (T)0.002 * tmax(-pv, (T)0.0);
    } else if (MODE == 1) {
        const T pv = at<1>(a, g, 0, i, j);
        r = (T)0.04 * tmax(pv, (T)0.0); gg = (T)0.04 * (T)0.0; bb = (T)0.04 * tmax(-pv, (T)0.0);
    } else if (MODE == 2) {
        const T w = diff_x<2>(a, g, k, 1, i, j) - diff_y<2>(a, g, k, 0, i, j);
        r = (T)0.005 * tmax(w, (T)0.0); gg = (T)0.005 * (T)0.0; bb = (T)0.005 * tmax(-w, (T)0.0);
    } else {
        r = at<3>(a, g, 0, i, j); gg = at<3>(a, g, 1, i, j); bb = at<3>(a, g, 2, i, j);
    }
    rgb[idx<3, T>(g, 0, i, j)] = r;
    rgb[idx<3, T>(g, 1, i, j)] = gg;
    rgb[idx<3, T>(g, 2, i, j)] = bb;
}

// ------------------------------------------------------------------------------------------------
// Boundary-condition kernels as op lists.
//
// The reference's BC kernels (fs/boundary_condition.py:16-65, 94-99) touch only boundary cells, but do so
// with in-kernel read/write hazards whose outcome depends on Taichi's loop order.  The
// library fixes the serial (i-major, j-minor) order as THE semantics: when the mask is uploaded the host
// enumerates the kernel's assignments in that order, groups assignments that touch a common cell (with at
// least one write) into a component, and the device runs one lane per component, executing its
// assignments in serial order.  Independent assignments (nearly all) run in parallel; the result is
// exactly the serial one for any mask.
// ------------------------------------------------------------------------------------------------
struct BcOps {
    // components of ONE assignment (nearly all of them): {target cell, source 1, source 2, kind | row << 2} - one 16-byte load per
    // lane, then the data, then the store: two dependent memory round trips instead of four (the kernels are pure latency)
    int nsimple;
    const int4 *simple;
    // hazard components of TWO assignments (every serial chain of the reference's scenes: a mirror whose source another mirror rewrites):
    // two records of the simple kind side by side - one round trip brings both, then the two dependent load -> store steps (the general
    // form below walks five arrays per assignment behind a [begin, end) lookup: five dependent round trips for the same two assignments)
    int npair;
    const int4 *pair;       // [2 * npair]
    // longer hazard components (several assignments that must run in serial order)
    int ncomp;
    const int *comp_begin;  // [ncomp + 1] into the op arrays
    const int *comp_rlo;    // [ncomp] min / max local target row of the component
    const int *comp_rhi;
    const int *kind;        // per op
    const int *tgt;         // per op: cell offset j*P + i (channel 0 of a C=1 field; scaled by the kernels)
    const int *s1, *s2;     // per op: source cell offsets (or -1)
    const int *row;         // per op: local row of the target (saves the kernels a division by the pitch)
    const int *srow;        // per op: local row of source 1
};

// element offset of channel c of the cell at C=1 offset `cell` in local row `row`, for a C-channel field
__device__ __forceinline__ size_t cell_off(const Grid &g, int cell, int row, int C, int c)
{ return ((size_t)row * C + c) * g.P + (cell - row * g.P); }

// K1  set_velocity_boundary_condition, fs/boundary_condition.py:16-39
//   kind 0: v[t] = -v[s1] (mirror into the 2nd wall layer), 1: v[t] = bc_const[t], 2: v[t].x = max(v[s1].x, 0.05)
template <typename T>
__device__ __forceinline__ void velocity_bc_op(const Grid &g, int kind, int t, int trow, int s, int srow, T *v, const T *bc_const, unsigned *hot)
{
    if (kind == 0) {             // a copy (negated) of a cell of the same buffer: cannot raise the buffer's maximum speed
        v[cell_off(g, t, trow, 2, 0)] = -v[cell_off(g, s, srow, 2, 0)];
        v[cell_off(g, t, trow, 2, 1)] = -v[cell_off(g, s, srow, 2, 1)];
    } else if (kind == 1) {
        const T x = bc_const[cell_off(g, t, trow, 2, 0)], y = bc_const[cell_off(g, t, trow, 2, 1)];
        v[cell_off(g, t, trow, 2, 0)] = x;
        v[cell_off(g, t, trow, 2, 1)] = y;
        raise_hot(hot, hot2(x, y));
    } else {                     // x from the neighbour, y stays: a NEW pair
        const T x = tmax(v[cell_off(g, s, srow, 2, 0)], (T)0.05);
        v[cell_off(g, t, trow, 2, 0)] = x;
        raise_hot(hot, hot2(x, v[cell_off(g, t, trow, 2, 1)]));
    }
}
// a two-assignment chain (BcOps::pair): both run, in order, when the chain's target rows meet [jb, je) - the rule of the general components
template <typename T>
__device__ __forceinline__ void velocity_bc_pair(const Grid &g, const int4 a, const int4 b, int jb, int je, T *v, const T *bc_const, unsigned *hot)
{
    const int ra = a.w >> 2, rb = b.w >> 2;
    if ((ra > rb ? ra : rb) < jb || (ra < rb ? ra : rb) >= je) return;
    velocity_bc_op(g, a.w & 3, a.x, ra, a.y, a.z, v, bc_const, hot);
    velocity_bc_op(g, b.w & 3, b.x, rb, b.y, b.z, v, bc_const, hot);
}
template <typename T>
__global__ __launch_bounds__(256) void k_velocity_bc(Grid g, BcOps ops, int jb, int je, T *v, const T *bc_const, unsigned *hot)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < ops.nsimple) {
        const int4 o = ops.simple[n];
        const int trow = o.w >> 2;
        if (trow < jb || trow >= je) return;
        // a simple mirror / outflow op reads a cell of the same row or of the row +-2 / +-1 next to it: its row travels in .z
        velocity_bc_op(g, o.w & 3, o.x, trow, o.y, o.z, v, bc_const, hot);
        return;
    }
    n -= ops.nsimple;
    if (n < ops.npair) { velocity_bc_pair(g, ops.pair[2 * n], ops.pair[2 * n + 1], jb, je, v, bc_const, hot); return; }
    n -= ops.npair;
    if (n >= ops.ncomp) return;
    if (ops.comp_rhi[n] < jb || ops.comp_rlo[n] >= je) return;
    for (int o = ops.comp_begin[n]; o < ops.comp_begin[n + 1]; ++o)
        velocity_bc_op(g, ops.kind[o], ops.tgt[o], ops.row[o], ops.s1[o], ops.srow[o], v, bc_const, hot);
}

// K7  set_pressure_boundary_condition, fs/boundary_condition.py:41-65
//   kind 0: p[t] = p[s1], 1: p[t] = (p[s1] + p[s2]) / 2, 2: p[t] = 0          (C = 1: cell offsets are element offsets)
template <typename T>
__global__ __launch_bounds__(256) void k_pressure_bc(Grid g, BcOps ops, int jb, int je, T *p)
{
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < ops.nsimple) {
        const int4 o = ops.simple[n];
        const int trow = o.w >> 2, kind = o.w & 3;
        if (trow < jb || trow >= je) return;
        T val = (T)0.0;
        if (kind == 0) val = p[o.y];
        else if (kind == 1) val = (p[o.y] + p[o.z]) / (T)2.0;
        p[o.x] = val;
        return;
    }
    n -= ops.nsimple;
    if (n < ops.npair) {
        const int4 a = ops.pair[2 * n], b = ops.pair[2 * n + 1];
        const int ra = a.w >> 2, rb = b.w >> 2;
        if ((ra > rb ? ra : rb) < jb || (ra < rb ? ra : rb) >= je) return;
        const int4 rec[2] = {a, b};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int kind = rec[q].w & 3;
            T val = (T)0.0;
            if (kind == 0) val = p[rec[q].y];
            else if (kind == 1) val = (p[rec[q].y] + p[rec[q].z]) / (T)2.0;
            p[rec[q].x] = val;
        }
        return;
    }
    n -= ops.npair;
    if (n >= ops.ncomp) return;
    if (ops.comp_rhi[n] < jb || ops.comp_rlo[n] >= je) return;
    for (int o = ops.comp_begin[n]; o < ops.comp_begin[n + 1]; ++o) {
        const int kind = ops.kind[o], t = ops.tgt[o];
        T val;
        if (kind == 0) val = p[ops.s1[o]];
        else if (kind == 1) val = (p[ops.s1[o]] + p[ops.s2[o]]) / (T)2.0;
        else val = (T)0.0;
        p[t] = val;
    }
}

// K10  set_dye_boundary_condition, fs/boundary_condition.py:94-99:  dye[t] = bc_dye[t] on inflow cells (never hazardous: all simple)
template <typename T>
__global__ __launch_bounds__(256) void k_dye_bc(Grid g, BcOps ops, int jb, int je, T *dye, const T *bc_dye)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= ops.nsimple) return;
    const int4 o = ops.simple[n];
    const int trow = o.w >> 2;
    if (trow < jb || trow >= je) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) dye[cell_off(g, o.x, trow, 3, c)] = bc_dye[cell_off(g, o.x, trow, 3, c)];
}

// ------------------------------------------------------------------------------------------------
// layout conversion: host (X, nrows, C) window [y contiguous, channels innermost]  <->  device [row][c][P]
// Viewed as a 2-D transpose of A[i][q] (q = r*C + c, Q = nrows*C columns) into B[q + r0*C][i].
// 64x64 tiles through LDS (pad 1) keep both sides coalesced.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_to_device(const T *__restrict__ A, T *__restrict__ B, int X, int Q, int P, int q0)
{
    __shared__ T tile[64][65];
    const int i0 = blockIdx.x * 64, qb = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {          // read A rows i0+r, columns qb+tx  (contiguous in q)
        int i = i0 + r, q = qb + tx;
        if (i < X && q < Q) tile[r][tx] = A[(size_t)i * Q + q];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {          // write B rows q0+qb+r, columns i0+tx  (contiguous in i)
        int q = qb + r, i = i0 + tx;
        if (i < X && q < Q) B[(size_t)(q0 + q) * P + i] = tile[tx][r];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_to_host(T *__restrict__ A, const T *__restrict__ B, int X, int Q, int P, int q0)
{
    __shared__ T tile[64][65];
    const int i0 = blockIdx.x * 64, qb = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        int q = qb + r, i = i0 + tx;
        if (i < X && q < Q) tile[r][tx] = B[(size_t)(q0 + q) * P + i];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int i = i0 + r, q = qb + tx;
        if (i < X && q < Q) A[(size_t)i * Q + q] = tile[tx][r];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_fill(T *p, size_t n, T v)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) p[t] = v;
}

}  // namespace fs
