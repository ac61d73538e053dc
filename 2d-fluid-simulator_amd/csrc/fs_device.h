// fs_device.h - device-side geometry, constants and stencil primitives shared by all kernels.
//
// Device layout (private to the library): a C-channel field over a slab of `rows`
// local rows is stored as [row][channel][P] with x contiguous and P = row pitch (X rounded up to
// 64 elements), i.e. element (i, r, c) lives at ((r*C + c)*P + i).  Rows are the y index, so a
// y-halo is a contiguous block and wave lanes walk x (coalesced 256 B .. 1 KiB per wave access).
//
// Arithmetic follows the reference's operation order literally; the file is compiled with
// -ffp-contract=off (no FMA fusion) and HIP's default correctly-rounded f32 divide/sqrt so that the
// results are bit-identical to an IEEE evaluation of the same expression tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs {

struct Grid {
    int X;        // cells per row (2*res)
    int P;        // field row pitch in elements
    int Pm;       // mask row pitch in bytes
    int rows;     // local rows allocated (ny_local + 2*halo)
    int jlo, jhi; // local rows of global y = 0 and y = Y-1 (clamp range of sample())
    int ybase;    // global y of local row 0 (red-black parity)
    const uint8_t *mask;  // [rows][Pm]; rows outside the global domain hold 1 (wall)
    // Compact launch (set per launch by the host, nullptr = the dense grid): the workgroups that have anything to do, as
    // (tile-row block << 12 | column block) of the dense decode, listed per XCD and interleaved - entry k * 8 + x is the k-th
    // active workgroup of XCD x (0xffffffff: padding).  A third of scene 5 is solid wall: those workgroups are never launched.
    const uint32_t *tiles;
};

// Kernel constants with Taichi's typing rules; filled on the host by make_konst().
template <typename T>
struct Konst {
    T dt, dx, re;
    T two_dx;     // 2.0*dx      folded in double   (fs/solver.py:257,260)
    T dx2_fold;   // dx**2       folded in double   (fs/solver.py:311-313)
    T dx3_fold;   // dx**3       folded in double   (fs/solver.py:304-305)
    T dx_sq;      // dx*dx in T                     (fs/differentiation.py:55,60)
    T six_dx;     // 6*dx  in T                     (fs/advection.py:46,58)
    T eight_dt;   // 8*dt  in T                     (fs/pressure_updater.py:37)
    T dtw;        // dt*weight   folded in double   (fs/vorticity_confinement.py:42)
    T om, om1;    // omega, 1.0-omega folded        (fs/pressure_updater.py:112)
    // When dx is an exact power of two (res = 1024, 4096, 8192 ...) every dx-derived divisor below is one too and
    // x / d == x * (1/d) bit for bit (exact scaling); p2 switches those divisions to multiplications.
    int p2;
    T inv_dx, inv_two_dx, inv_dx_sq, inv_dx2_fold, inv_dx3_fold;
    // RN64(1 / d) of every loop-invariant divisor d (as rounded to T), for the f64-multiply division below (f32 fields only)
    double r_dx, r_two_dx, r_dx_sq, r_dx2_fold, r_dx3_fold, r_six_dx, r_eight_dt, r_re;
};

// ---- division by loop-invariant divisors --------------------------------------------------------------------------------
// HIP's IEEE f32 division expands to 11 instructions (div_scale x2, rcp, 5 fma, div_fmas, div_fixup); the kernels divide by a handful
// of launch constants (dx, 2 dx, dx^2, dx^3, 6 dx, 8 dt, Re) up to 20 times per cell.  Two exact shortcuts, chosen per launch through
// the kernels' DM template parameter:
//   bit 0 (DM_P2):  the dx-derived divisors are powers of two (res 1024, 4096, 8192 ...) -> x * (1/d) is exact
//   bit 2 (DM_F64): every other loop-invariant divisor through ONE f64 multiplication (f32 fields only), see f64div
// DM_IEEE = neither: the plain division (f64 fields, and f32 divisors that admit a tie).
// (Round 2's reciprocal-FMA sequence with its range guard, redo path and exhaustive per-divisor check - bit 1 - is gone: the f64
//  multiplication is shorter, needs no guard and measured faster everywhere, MAC update included: 176 against 208 / 194 us.)
constexpr int DM_IEEE = 0, DM_P2 = 1, DM_F64 = 4;
// The f32 quotient x / d through ONE f64 multiplication: with R = RN64(1 / d), RN32(RN64(x R)) == RN32(x / d) for every f32 x unless x / d
// is EXACTLY a rounding boundary (tie) of the f32 grid: x R misses the real quotient by less than 2^-52 relative (R's rounding + the
// product's), while a quotient of two 24-bit numbers that is not a tie stays at least 2^-49 relative away from one - the argument that
// makes double rounding innocuous for division once the wide format has 2 p + 2 bits (Figueroa).  3 instructions (v_cvt_f64_f32,
// v_mul_f64, v_cvt_f32_f64) against 11.
// Exact ties.  A NORMAL quotient is never one (a 25-bit tie times the divisor does not fit a 24-bit dividend).  Among the
// denormals the grid is coarser, and x / d = (m + 1/2) 2^-149 happens for f32 x exactly when d = D 2^e with D odd and e >= 1 - an EVEN
// INTEGER (x = D (2 m + 1) 2^(e - 150) must be a multiple of 2^-149).  RN64(x R) can then sit one f64 ulp beside the tie, on the side
// round-to-even would not have taken (d = 1e5, x = 9.108440018111311e-40).  Of this library's loop-invariant divisors only the Reynolds
// number can be an even integer in practice (1e6, 1000, 1e8 ...); dx, 2 dx, dx^2, dx^3, 6 dx are below one, 8 dt is unless dt >= 1/4.  So:
//   * tie_free(d) is decided per divisor on the host (fs_host.h): the plain three-instruction form is used only for divisors that admit
//     no tie; a launch whose dx- / dt-derived divisors fail the test falls back to the IEEE division (never seen outside the tests);
//   * the division by Re (rdiv below) always takes the guarded form: a result that came out denormal is recomputed by the IEEE division.  That catches every wrong answer: the wrong and the right result of a tie are neighbours, and of such a pair one
//     is always a denormal (the tie between the largest denormal and 2^-126 resolves to 2^-126, the one between 0 and 2^-149 to 0 - the
//     wrong answer is the denormal).  One v_cmp_class and a branch no healthy wave takes, on ONE division per cell and component.
// Checked: tests/test_f64div.py (numpy: all significands of several binades, random bit patterns, and every tie dividend of even-integer
// divisors against the guarded form) and on the device per divisor (fs_selftest_f64div runs the form the library would use for it).
__device__ __forceinline__ float f64div(float x, double rd) { return (float)((double)x * rd); }
__device__ __forceinline__ double f64div(double x, double) { return x; }                    // f64 fields: never selected
#ifndef FS_RDIV_FIX
#define FS_RDIV_FIX 2
#endif
__device__ __forceinline__ float f64div_guarded(float x, float d, double rd)
{
#if FS_RDIV_FIX == 1      // (A/B) branch-free: one Newton step makes the f64 quotient exact for exact cases - two f64 FMAs on every division
    const double xd = (double)x;
    double p = xd * rd;
    const double r = __builtin_fma(-p, (double)d, xd);
    p = __builtin_fma(r, rd, p);
    return (float)p;
#elif FS_RDIV_FIX == 2
    float q = (float)((double)x * rd);
    if (__builtin_expect(__builtin_amdgcn_class(q, 0x90), 0)) q = x / d;       // 0x90: -denormal | +denormal
    return q;
#else                     // (A/B) round 3's unguarded form
    (void)d;
    return (float)((double)x * rd);
#endif
}
__device__ __forceinline__ double f64div_guarded(double x, double, double) { return x; }
// x / d for a dx-derived divisor / for any other loop-invariant divisor / for the Reynolds number (always guarded: one division per cell and component)
template <int DM, typename T>
__device__ __forceinline__ T xdiv(T x, T d, T inv_d, double rd) { return (DM & DM_P2) ? x * inv_d : ((DM & DM_F64) ? f64div(x, rd) : x / d); }
template <int DM, typename T>
__device__ __forceinline__ T cdiv(T x, T d, double rd) { return (DM & DM_F64) ? f64div(x, rd) : x / d; }
template <int DM, typename T>
__device__ __forceinline__ T rdiv(T x, T d, double rd) { return (DM & DM_F64) ? f64div_guarded(x, d, rd) : x / d; }


// ---- packed f32: the two cells of a lane as ONE operand --------------------------------------------------------------------
// gfx950 issues v_pk_mul_f32 / v_pk_add_f32 (two IEEE f32 operations on an aligned register pair, each rounded like its scalar form) in
// the slot of one scalar operation.  The tile kernels hold 2 consecutive x cells per lane and evaluate the same expression tree for
// both: written on v2f the mul / add / sub chains pack; selects, DPP shifts and the f64-multiply divisions stay per half.  No
// contraction (-ffp-contract=off), no reassociation: the bits are those of the scalar form (tests compare them at tolerance 0).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
template <typename V> struct VecOf;
template <> struct VecOf<float> { using S = float; using D = double; };
template <> struct VecOf<double> { using S = double; using D = double; };
template <> struct VecOf<v2f> { using S = float; using D = v2d; };
__device__ __forceinline__ double to_dbl(float x) { return (double)x; }
__device__ __forceinline__ double to_dbl(double x) { return x; }
__device__ __forceinline__ v2d to_dbl(v2f x) { return __builtin_convertvector(x, v2d); }
// x < 0 ? a : b  per element (sign(0) = +1, fs/differentiation.py:12-14)
__device__ __forceinline__ float sel_neg(float x, float a, float b) { return x < 0.0f ? a : b; }
__device__ __forceinline__ double sel_neg(double x, double a, double b) { return x < 0.0 ? a : b; }
__device__ __forceinline__ v2f sel_neg(v2f x, v2f a, v2f b) { v2f r; r.x = x.x < 0.0f ? a.x : b.x; r.y = x.y < 0.0f ? a.y : b.y; return r; }
__device__ __forceinline__ v2f f64div(v2f x, v2d rd) { return __builtin_convertvector(__builtin_convertvector(x, v2d) * rd, v2f); }
template <int DM> __device__ __forceinline__ v2f xdiv(v2f x, v2f d, v2f inv_d, v2d rd) { return (DM & DM_P2) ? x * inv_d : ((DM & DM_F64) ? f64div(x, rd) : x / d); }
template <int DM> __device__ __forceinline__ v2f xdiv(v2f x, float d, float inv_d, double rd)
{ return (DM & DM_P2) ? x * inv_d : ((DM & DM_F64) ? f64div(x, (v2d)rd) : x / d); }
template <int DM> __device__ __forceinline__ v2f cdiv(v2f x, float d, double rd) { return (DM & DM_F64) ? f64div(x, (v2d)rd) : x / d; }

template <typename T> __device__ __forceinline__ T tmin(T a, T b);
template <typename T> __device__ __forceinline__ T tmax(T a, T b);
template <> __device__ __forceinline__ float tmin<float>(float a, float b) { return fminf(a, b); }
template <> __device__ __forceinline__ float tmax<float>(float a, float b) { return fmaxf(a, b); }
template <> __device__ __forceinline__ double tmin<double>(double a, double b) { return fmin(a, b); }
template <> __device__ __forceinline__ double tmax<double>(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float tsqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double tsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ float tabs(float a) { return fabsf(a); }
__device__ __forceinline__ double tabs(double a) { return fabs(a); }

// ---- "hot" flag of a velocity buffer --------------------------------------------------------------------------------------
// limit_field (fs/solver.py:38-43) rewrites the cells whose speed exceeds 10 - in a healthy run none, yet the pass reads the whole
// velocity field every step (268 MB at res 4096).  Every kernel that writes a 2-channel field raises the buffer's flag word when
// it stores a value with x*x + y*y > 99 (the same expression limit_field takes the square root of, so speed <= sqrt(99) = 9.9499 is
// certain otherwise); the flag is never cleared.  fs_limit_field exits at once while the flag is down and the limit's square is above
// 99.01 (the margin covers the rounding of limit * limit): the result is the same for every input, the common case costs a 3 us launch
// instead of 47 us.  Uploads scan what they bring in, ghost-row exchanges check what they unpack.  NaN never raises the flag -
// limit_field ignores NaN as well (NaN > limit is false).  (Round 4: 99 instead of 64.  Kernels that see ONE component per wave - the
// fused K3+K4 pass - must raise at x*x > 99 / 2, i.e. |x| > 7.04 instead of 5.66: the headline scene passes 5.66 after ~3000 steps
// without ever needing the limiter, and a raised flag costs the full pass, 54 us = 8 % of its step, from then on.)
// Word [3] (round 4): the fused K3+K4 pass sees ONE component per wave and can only test x*x > 99 / 2 - a flow with |u| > 7.04 and a speed
// below 9.95 would keep the limit pass running for nothing (the headline scene, 6 000 steps in: 57 us = 8 % of its step).  Its FLUID-cell
// raises therefore go to word [3] of the buffer, which the kernels that rewrite every fluid cell of a buffer with both components in sight
// (K2, K5+K6: exact test, word [0]) clear when they cover the whole single-GPU grid.  Every gate reads [3] as well: a buffer that goes from
// K3+K4 straight to limit_field (no vorticity confinement) is judged as conservatively as before.
constexpr float FS_HOT_SQ = 99.0f;
constexpr float FS_HOT_GATE_SQ = 99.01f;      // a limit is gated by the flag when limit * limit exceeds this
template <typename T> __device__ __forceinline__ bool hot2(T x, T y) { return x * x + y * y > (T)FS_HOT_SQ; }
template <typename T> __device__ __forceinline__ bool hot1(T x) { return x * x > (T)(0.5f * FS_HOT_SQ); }      // one component alone
__device__ __forceinline__ void raise_hot(unsigned *hot, bool h) { if (h) atomicOr(hot, 1u); }                  // rare: no wave logic

// x / d with the exact-reciprocal shortcut (see Konst::p2)
// P2 is a COMPILE-TIME switch: a run-time branch per division splits the kernel into basic blocks and
// serialises its loads (measured: K2 192 -> 302 us).
template <bool P2, typename T>
__device__ __forceinline__ T qdiv(T x, T d, T inv_d) { return P2 ? x * inv_d : x / d; }

__device__ __forceinline__ int clampx(const Grid &g, int i) { return i < 0 ? 0 : (i > g.X - 1 ? g.X - 1 : i); }
__device__ __forceinline__ int clampy(const Grid &g, int j) { return j < g.jlo ? g.jlo : (j > g.jhi ? g.jhi : j); }

template <int C, typename T>
__device__ __forceinline__ size_t idx(const Grid &g, int c, int i, int j) { return ((size_t)j * C + c) * g.P + i; }

// direct read / write of an in-range cell
template <int C, typename T>
__device__ __forceinline__ T at(const T *f, const Grid &g, int c, int i, int j) { return f[idx<C, T>(g, c, i, j)]; }

// fs/differentiation.py:4-9  sample(): clamp-to-edge read
template <int C, typename T>
__device__ __forceinline__ T smp(const T *f, const Grid &g, int c, int i, int j)
{ return f[idx<C, T>(g, c, clampx(g, i), clampy(g, j))]; }

__device__ __forceinline__ uint8_t mask_at(const Grid &g, int i, int j) { return g.mask[(size_t)j * g.Pm + i]; }

// fs/differentiation.py:41-50  central differences
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T diff_x(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>((T)0.5 * (smp<C>(f, g, c, i + 1, j) - smp<C>(f, g, c, i - 1, j)), k.dx, k.inv_dx); }
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T diff_y(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>((T)0.5 * (smp<C>(f, g, c, i, j + 1) - smp<C>(f, g, c, i, j - 1)), k.dx, k.inv_dx); }
// fs/differentiation.py:53-60  second differences
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T diff2_x(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>((smp<C>(f, g, c, i + 1, j) - (T)2.0 * smp<C>(f, g, c, i, j)) + smp<C>(f, g, c, i - 1, j), k.dx_sq, k.inv_dx_sq); }
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T diff2_y(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>((smp<C>(f, g, c, i, j + 1) - (T)2.0 * smp<C>(f, g, c, i, j)) + smp<C>(f, g, c, i, j - 1), k.dx_sq, k.inv_dx_sq); }
// fs/differentiation.py:17-26  forward differences
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T fdiff_x(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>(smp<C>(f, g, c, i + 1, j) - smp<C>(f, g, c, i, j), k.dx, k.inv_dx); }
template <int C, bool P2 = false, typename T>
__device__ __forceinline__ T fdiff_y(const T *f, const Grid &g, const Konst<T> &k, int c, int i, int j)
{ return qdiv<P2>(smp<C>(f, g, c, i, j + 1) - smp<C>(f, g, c, i, j), k.dx, k.inv_dx); }

}  // namespace fs
