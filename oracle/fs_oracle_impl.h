/*
 * fs_oracle_impl.h - type-generic body of the CPU oracle (included twice by fs_oracle.c:
 * REAL=float / SUF=f32 and REAL=double / SUF=f64).
 *
 * TEST INFRASTRUCTURE.  A plain-C restatement of the reference algorithm
 * (takah29/2d-fluid-simulator, package fs/) used only as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never linked into the product.
 *
 * Layout = the reference's own: field[i, j] with i = x in [0, X), j = y in [0, Y), j contiguous,
 * vector fields AoS (fs/double_buffer.py:7-11), i.e. element (i, j, c) lives at ((i*Y + j)*C + c).
 * Every loop nest visits cells in (i-major, j-minor) order; the two boundary-condition kernels rely
 * on that order for the reference's in-kernel read/write hazards.
 *
 * Numeric typing follows Taichi's rules: Python-float attributes (dt, dx, re, omega)
 * are constants rounded to REAL; pure-Python sub-expressions (2.0*dx, dx**2, dx**3, dt*weight, 1.0-omega)
 * fold in double and are rounded once; `dx**2`, `6*dx`, `8*dt` on float-annotated @ti.func arguments are
 * REAL arithmetic.  No FMA contraction (build with -ffp-contract=off), IEEE division / sqrt.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef struct {
    int X, Y;
    REAL dt, dx, re;   /* float(self.dt) etc. as kernel constants                         */
    REAL two_dx;       /* 2.0 * self.dx            folded in double  (solver.py:257,260)   */
    REAL dx2_fold;     /* self.dx**2               folded in double  (solver.py:311-313)   */
    REAL dx3_fold;     /* self.dx**3               folded in double  (solver.py:304-305)   */
    REAL dx_sq;        /* dx**2 on a REAL argument (differentiation.py:55,60)              */
    REAL six_dx;       /* 6 * dx on a REAL argument (advection.py:46,58)                   */
    REAL eight_dt;     /* 8 * dt on a REAL argument (pressure_updater.py:37)               */
} FN(consts);

static FN(consts) FN(mk)(int X, int Y, double dt, double dx, double re)
{
    FN(consts) k;
    k.X = X; k.Y = Y;
    k.dt = (REAL)dt; k.dx = (REAL)dx; k.re = (REAL)re;
    k.two_dx = (REAL)(2.0 * dx);
    k.dx2_fold = (REAL)pow(dx, 2.0);
    k.dx3_fold = (REAL)pow(dx, 3.0);
    k.dx_sq = k.dx * k.dx;
    k.six_dx = (REAL)6 * k.dx;
    k.eight_dt = (REAL)8 * k.dt;
    return k;
}

/* differentiation.py:4-9  sample(): clamp-to-edge read */
static inline REAL FN(smp)(const REAL *f, int C, int X, int Y, int i, int j, int c)
{
    i = i < 0 ? 0 : (i > X - 1 ? X - 1 : i);
    j = j < 0 ? 0 : (j > Y - 1 ? Y - 1 : j);
    return f[((size_t)i * Y + j) * C + c];
}
#define S(f, C, i, j, c) FN(smp)(f, C, X, Y, i, j, c)
#define AT(f, C, i, j, c) f[((size_t)(i) * Y + (j)) * (C) + (c)]

/* differentiation.py:41-50 */
static inline REAL FN(diff_x)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y; return ((REAL)0.5 * (S(f, C, i + 1, j, c) - S(f, C, i - 1, j, c))) / k->dx; }
static inline REAL FN(diff_y)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y; return ((REAL)0.5 * (S(f, C, i, j + 1, c) - S(f, C, i, j - 1, c))) / k->dx; }
/* differentiation.py:53-60 */
static inline REAL FN(diff2_x)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y;
  return ((S(f, C, i + 1, j, c) - (REAL)2.0 * S(f, C, i, j, c)) + S(f, C, i - 1, j, c)) / k->dx_sq; }
static inline REAL FN(diff2_y)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y;
  return ((S(f, C, i, j + 1, c) - (REAL)2.0 * S(f, C, i, j, c)) + S(f, C, i, j - 1, c)) / k->dx_sq; }
/* differentiation.py:17-26 */
static inline REAL FN(fdiff_x)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y; return (S(f, C, i + 1, j, c) - S(f, C, i, j, c)) / k->dx; }
static inline REAL FN(fdiff_y)(const FN(consts) *k, const REAL *f, int C, int i, int j, int c)
{ int X = k->X, Y = k->Y; return (S(f, C, i, j + 1, c) - S(f, C, i, j, c)) / k->dx; }

/* advection.py:12-24  advect_upwind (component c of phi) */
static inline REAL FN(adv_upwind)(const FN(consts) *k, const REAL *vc, const REAL *phi, int C, int i, int j, int c)
{
    int Y = k->Y;
    REAL ux = AT(vc, 2, i, j, 0), uy = AT(vc, 2, i, j, 1);
    int kk = ux < (REAL)0.0 ? i : i - 1;
    REAL a = ux * FN(fdiff_x)(k, phi, C, kk, j, c);
    kk = uy < (REAL)0.0 ? j : j - 1;
    REAL b = uy * FN(fdiff_y)(k, phi, C, i, kk, c);
    return a + b;
}

/* advection.py:27-60  advect_kk_scheme: coefficient vector dotted left to right */
static inline REAL FN(adv_kk)(const FN(consts) *k, const REAL *vc, const REAL *phi, int C, int i, int j, int c)
{
    static const REAL neg[5] = {-2, 10, -9, 2, -1};   /* coef            (u < 0) */
    static const REAL pos[5] = {1, -2, 9, -10, 2};    /* -coef[::-1]     (u >= 0) */
    int X = k->X, Y = k->Y;
    REAL ux = AT(vc, 2, i, j, 0), uy = AT(vc, 2, i, j, 1);
    const REAL *w = ux < (REAL)0 ? neg : pos;
    REAL acc = S(phi, C, i + 2, j, c) * w[0];
    acc = acc + S(phi, C, i + 1, j, c) * w[1];
    acc = acc + S(phi, C, i, j, c) * w[2];
    acc = acc + S(phi, C, i - 1, j, c) * w[3];
    acc = acc + S(phi, C, i - 2, j, c) * w[4];
    REAL a = acc / k->six_dx;
    w = uy < (REAL)0 ? neg : pos;
    acc = S(phi, C, i, j + 2, c) * w[0];
    acc = acc + S(phi, C, i, j + 1, c) * w[1];
    acc = acc + S(phi, C, i, j, c) * w[2];
    acc = acc + S(phi, C, i, j - 1, c) * w[3];
    acc = acc + S(phi, C, i, j - 2, c) * w[4];
    REAL b = acc / k->six_dx;
    return ux * a + uy * b;
}

static inline REAL FN(adv)(int scheme, const FN(consts) *k, const REAL *vc, const REAL *phi, int C, int i, int j, int c)
{ return scheme == 0 ? FN(adv_upwind)(k, vc, phi, C, i, j, c) : FN(adv_kk)(k, vc, phi, C, i, j, c); }

/* ---------------------------------------------------------------------------------------------
 * K1  BoundaryCondition.set_velocity_boundary_condition   (boundary_condition.py:16-39)
 * in place, serial (i, j) order: mirror scatter into the 2nd wall layer / inflow const / outflow floor
 * ------------------------------------------------------------------------------------------- */
#define M(i, j) mask[(size_t)(i) * Y + (j)]
static inline void FN(velocity_bc_cell)(int X, int Y, const uint8_t *mask, const REAL *bc_const, REAL *v, int i, int j)
{
    uint8_t m = M(i, j);
    if (m == 1 && 1 <= i && i < X - 1 && 1 <= j && j < Y - 1) {
        if (M(i - 1, j) == 0 && M(i, j - 1) == 1 && M(i, j + 1) == 1) {
            for (int c = 0; c < 2; ++c) AT(v, 2, i + 1, j, c) = -S(v, 2, i - 1, j, c);
        } else if (M(i + 1, j) == 0 && M(i, j - 1) == 1 && M(i, j + 1) == 1) {
            for (int c = 0; c < 2; ++c) AT(v, 2, i - 1, j, c) = -S(v, 2, i + 1, j, c);
        } else if (M(i, j - 1) == 0 && M(i - 1, j) == 1 && M(i + 1, j) == 1) {
            for (int c = 0; c < 2; ++c) AT(v, 2, i, j + 1, c) = -S(v, 2, i, j - 1, c);
        } else if (M(i, j + 1) == 0 && M(i - 1, j) == 1 && M(i + 1, j) == 1) {
            for (int c = 0; c < 2; ++c) AT(v, 2, i, j - 1, c) = -S(v, 2, i, j + 1, c);
        }
    } else if (m == 2) {
        AT(v, 2, i, j, 0) = AT(bc_const, 2, i, j, 0);
        AT(v, 2, i, j, 1) = AT(bc_const, 2, i, j, 1);
    } else if (m == 3) {
        REAL l = S(v, 2, i - 1, j, 0);
        AT(v, 2, i, j, 0) = FMAX(l, (REAL)0.05);
    }
}

void FN(oracle_velocity_bc)(int X, int Y, const uint8_t *mask, const REAL *bc_const, REAL *v)
{
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j) FN(velocity_bc_cell)(X, Y, mask, bc_const, v, i, j);
}

/* The same kernel over a precomputed list of the cells that can act at all (oracle_bc_cells: serial (i, j) order is kept, so
 * the result is the serial one) - O(perimeter) instead of a serial walk over every cell; used for the large-grid CPU baseline. */
void FN(oracle_velocity_bc_list)(int X, int Y, const uint8_t *mask, const REAL *bc_const, REAL *v, const int64_t *cells, int n)
{
    for (int q = 0; q < n; ++q) FN(velocity_bc_cell)(X, Y, mask, bc_const, v, (int)(cells[q] / Y), (int)(cells[q] % Y));
}

/* ---------------------------------------------------------------------------------------------
 * K7  BoundaryCondition.set_pressure_boundary_condition   (boundary_condition.py:41-65)
 * Unguarded mask reads outside the grid (H3) see "wall"; they can never select a case that
 * changes a value.
 * ------------------------------------------------------------------------------------------- */
static inline int FN(mask_at)(const uint8_t *mask, int X, int Y, int i, int j)
{ return (i < 0 || i >= X || j < 0 || j >= Y) ? 1 : mask[(size_t)i * Y + j]; }

#define MM(i, j) FN(mask_at)(mask, X, Y, i, j)
static inline void FN(pressure_bc_cell)(int X, int Y, const uint8_t *mask, REAL *p, int i, int j)
{
    uint8_t m = M(i, j);
    if (m == 1) {
        if (MM(i - 1, j) == 0 && MM(i, j - 1) == 1 && MM(i, j + 1) == 1) AT(p, 1, i, j, 0) = S(p, 1, i - 1, j, 0);
        else if (MM(i + 1, j) == 0 && MM(i, j - 1) == 1 && MM(i, j + 1) == 1) AT(p, 1, i, j, 0) = S(p, 1, i + 1, j, 0);
        else if (MM(i, j - 1) == 0 && MM(i - 1, j) == 1 && MM(i + 1, j) == 1) AT(p, 1, i, j, 0) = S(p, 1, i, j - 1, 0);
        else if (MM(i, j + 1) == 0 && MM(i - 1, j) == 1 && MM(i + 1, j) == 1) AT(p, 1, i, j, 0) = S(p, 1, i, j + 1, 0);
        else if (MM(i - 1, j) == 0 && MM(i, j + 1) == 0) AT(p, 1, i, j, 0) = (S(p, 1, i - 1, j, 0) + S(p, 1, i, j + 1, 0)) / (REAL)2.0;
        else if (MM(i + 1, j) == 0 && MM(i, j + 1) == 0) AT(p, 1, i, j, 0) = (S(p, 1, i + 1, j, 0) + S(p, 1, i, j + 1, 0)) / (REAL)2.0;
        else if (MM(i - 1, j) == 0 && MM(i, j - 1) == 0) AT(p, 1, i, j, 0) = (S(p, 1, i - 1, j, 0) + S(p, 1, i, j - 1, 0)) / (REAL)2.0;
        else if (MM(i + 1, j) == 0 && MM(i, j - 1) == 0) AT(p, 1, i, j, 0) = (S(p, 1, i + 1, j, 0) + S(p, 1, i, j - 1, 0)) / (REAL)2.0;
    } else if (m == 2) {
        AT(p, 1, i, j, 0) = S(p, 1, i + 1, j, 0);
    } else if (m == 3) {
        AT(p, 1, i, j, 0) = (REAL)0.0;
    }
}

void FN(oracle_pressure_bc)(int X, int Y, const uint8_t *mask, REAL *p)
{
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j) FN(pressure_bc_cell)(X, Y, mask, p, i, j);
}

void FN(oracle_pressure_bc_list)(int X, int Y, const uint8_t *mask, REAL *p, const int64_t *cells, int n)
{
    for (int q = 0; q < n; ++q) FN(pressure_bc_cell)(X, Y, mask, p, (int)(cells[q] / Y), (int)(cells[q] % Y));
}
#undef MM

/* K10  DyeBoundaryCondition.set_dye_boundary_condition   (boundary_condition.py:94-99) */
void FN(oracle_dye_bc)(int X, int Y, const uint8_t *mask, const REAL *bc_dye, REAL *dye)
{
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 2)
                for (int c = 0; c < 3; ++c) AT(dye, 3, i, j, c) = AT(bc_dye, 3, i, j, c);
}

/* ---------------------------------------------------------------------------------------------
 * K2'  MacSolver._update_velocities   (solver.py:94-107)   scheme: 0 upwind, 1 Kawamura-Kuwahara
 * ------------------------------------------------------------------------------------------- */
void FN(oracle_mac_update)(int X, int Y, double dt, double dx, double re, int scheme,
                           const uint8_t *mask, REAL *vn, const REAL *vc, const REAL *pc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, re); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 0) {
                REAL gp[2] = { FN(diff_x)(k, pc, 1, i, j, 0), FN(diff_y)(k, pc, 1, i, j, 0) };
                for (int c = 0; c < 2; ++c) {
                    REAL a = FN(adv)(scheme, k, vc, vc, 2, i, j, c);
                    REAL lap = (FN(diff2_x)(k, vc, 2, i, j, c) + FN(diff2_y)(k, vc, 2, i, j, c)) / k->re;
                    AT(vn, 2, i, j, c) = AT(vc, 2, i, j, c) + k->dt * (((-a) - gp[c]) + lap);
                }
            }
}

/* K11  DyeMacSolver._update_dye   (solver.py:157-161) */
void FN(oracle_mac_dye)(int X, int Y, double dt, double dx, double re, int scheme,
                        const uint8_t *mask, REAL *dn, const REAL *dc, const REAL *vc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, re); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 0)
                for (int c = 0; c < 3; ++c)
                    AT(dn, 3, i, j, c) = AT(dc, 3, i, j, c) - k->dt * FN(adv)(scheme, k, vc, dc, 3, i, j, c);
}

/* K0  CipMacSolver._set_grad   (solver.py:207-211), all cells */
void FN(oracle_cip_set_grad)(int X, int Y, double dx, int C, REAL *fx, REAL *fy, const REAL *f)
{
    FN(consts) kk = FN(mk)(X, Y, 1.0, dx, 1.0); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            for (int c = 0; c < C; ++c) {
                AT(fx, C, i, j, c) = FN(diff_x)(k, f, C, i, j, c);
                AT(fy, C, i, j, c) = FN(diff_y)(k, f, C, i, j, c);
            }
}

/* K2  CipMacSolver._non_advection_phase   (solver.py:229-240, 263-265), not-wall cells */
void FN(oracle_cip_nonadv)(int X, int Y, double dt, double dx, double re,
                           const uint8_t *mask, REAL *fn, const REAL *fc, const REAL *pc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, re); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) != 1) {
                REAL gp[2] = { FN(diff_x)(k, pc, 1, i, j, 0), FN(diff_y)(k, pc, 1, i, j, 0) };
                for (int c = 0; c < 2; ++c) {
                    REAL dif = (FN(diff2_x)(k, fc, 2, i, j, c) + FN(diff2_y)(k, fc, 2, i, j, c)) / k->re;
                    REAL g = (-gp[c]) + dif;
                    AT(fn, 2, i, j, c) = AT(fc, 2, i, j, c) + g * k->dt;
                }
            }
}

/* K12  DyeCipMacSolver._non_advection_phase_dye   (solver.py:378-383) */
void FN(oracle_cip_nonadv_dye)(int X, int Y, double dt, double dx, double re,
                               const uint8_t *mask, REAL *dn, const REAL *dc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, re); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) != 1)
                for (int c = 0; c < 3; ++c) {
                    REAL dif = (FN(diff2_x)(k, dc, 3, i, j, c) + FN(diff2_y)(k, dc, 3, i, j, c)) / k->re;
                    AT(dn, 3, i, j, c) = AT(dc, 3, i, j, c) + dif * k->dt;
                }
}

/* K3  _non_advection_phase_grad   (solver.py:242-261).  The reference indexes fn/fc directly
 * (no sample()); at i = 0 / X-1 that is an out-of-bounds read (H2).  Policy here: clamp. */
void FN(oracle_cip_nonadv_grad)(int X, int Y, double dx, int C, const uint8_t *mask,
                                REAL *fxn, REAL *fyn, const REAL *fxc, const REAL *fyc,
                                const REAL *fc, const REAL *fn)
{
    FN(consts) kk = FN(mk)(X, Y, 1.0, dx, 1.0); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) != 1)
                for (int c = 0; c < C; ++c) {
                    AT(fxn, C, i, j, c) = AT(fxc, C, i, j, c)
                        + (((S(fn, C, i + 1, j, c) - S(fc, C, i + 1, j, c)) - S(fn, C, i - 1, j, c)) + S(fc, C, i - 1, j, c)) / k->two_dx;
                    AT(fyn, C, i, j, c) = AT(fyc, C, i, j, c)
                        + (((S(fn, C, i, j + 1, c) - S(fc, C, i, j + 1, c)) - S(fn, C, i, j - 1, c)) + S(fc, C, i, j - 1, c)) / k->two_dx;
                }
}

/* K4  _advection_phase / _cip_advect   (solver.py:267-332), fluid cells, C channels advected by v */
void FN(oracle_cip_advect)(int X, int Y, double dt, double dx, int C, const uint8_t *mask,
                           REAL *fn, REAL *fxn, REAL *fyn, const REAL *fc, const REAL *fxc, const REAL *fyc,
                           const REAL *v)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, 1.0); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 0) {
                REAL vx_ = AT(v, 2, i, j, 0), vy_ = AT(v, 2, i, j, 1);
                int i_s = vx_ < (REAL)0.0 ? -1 : 1;          /* differentiation.py:12-14 sign(0) = +1 */
                int j_s = vy_ < (REAL)0.0 ? -1 : 1;
                int i_m = i - i_s, j_m = j - j_s;
                REAL is = (REAL)i_s, js = (REAL)j_s;
                REAL i_s_denom = (REAL)(i_s) * k->dx3_fold;  /* i_s * self.dx**3 : exact sign flip */
                REAL j_s_denom = (REAL)(j_s) * k->dx3_fold;
                REAL is_dx = (REAL)(i_s) * k->dx;            /* i_s * self.dx */
                REAL X_ = (-vx_) * k->dt;
                REAL Y_ = (-vy_) * k->dt;
                REAL ddx[2] = { FN(diff_x)(k, v, 2, i, j, 0), FN(diff_x)(k, v, 2, i, j, 1) };
                REAL ddy[2] = { FN(diff_y)(k, v, 2, i, j, 0), FN(diff_y)(k, v, 2, i, j, 1) };
                for (int c = 0; c < C; ++c) {
                    REAL f00 = AT(fc, C, i, j, c), f0m = S(fc, C, i, j_m, c), fm0 = S(fc, C, i_m, j, c), fmm = S(fc, C, i_m, j_m, c);
                    REAL fx00 = AT(fxc, C, i, j, c), fxm0 = S(fxc, C, i_m, j, c), fx0m = S(fxc, C, i, j_m, c);
                    REAL fy00 = AT(fyc, C, i, j, c), fy0m = S(fyc, C, i, j_m, c), fym0 = S(fyc, C, i_m, j, c);
                    REAL tmp1 = ((f00 - f0m) - fm0) + fmm;
                    REAL tmp2 = fm0 - f00;
                    REAL tmp3 = f0m - f00;
                    REAL a = ((is * (fxm0 + fx00)) * k->dx - (REAL)2.0 * (-tmp2)) / i_s_denom;
                    REAL b = ((js * (fy0m + fy00)) * k->dx - (REAL)2.0 * (-tmp3)) / j_s_denom;
                    REAL cc = ((-tmp1) - (is * (fx0m - fx00)) * k->dx) / j_s_denom;
                    REAL d = ((-tmp1) - (js * (fym0 - fy00)) * k->dx) / i_s_denom;
                    REAL e = ((REAL)3.0 * tmp2 + (is * (fxm0 + (REAL)2.0 * fx00)) * k->dx) / k->dx2_fold;
                    REAL f = ((REAL)3.0 * tmp3 + (js * (fy0m + (REAL)2.0 * fy00)) * k->dx) / k->dx2_fold;
                    REAL g = ((-(fym0 - fy00)) + cc * k->dx2_fold) / is_dx;
                    AT(fn, C, i, j, c) =
                        (((((a * X_ + cc * Y_) + e) * X_ + g * Y_) + fx00) * X_
                         + (((b * Y_ + d * X_) + f) * Y_ + fy00) * Y_)
                        + f00;
                    REAL Fx = ((((REAL)3.0 * a) * X_ + ((REAL)2.0 * cc) * Y_) + (REAL)2.0 * e) * X_ + (d * Y_ + g) * Y_ + fx00;
                    REAL Fy = ((((REAL)3.0 * b) * Y_ + ((REAL)2.0 * d) * X_) + (REAL)2.0 * f) * Y_ + (cc * X_ + g) * X_ + fy00;
                    AT(fxn, C, i, j, c) = Fx - (k->dt * (Fx * ddx[0] + Fy * ddx[1])) / (REAL)2.0;
                    AT(fyn, C, i, j, c) = Fy - (k->dt * (Fx * ddy[0] + Fy * ddy[1])) / (REAL)2.0;
                }
            }
}

/* K5  VorticityConfinement._calc_vorticity   (vorticity_confinement.py:27-32) */
void FN(oracle_vort_calc)(int X, int Y, double dx, const uint8_t *mask, REAL *vort, REAL *vort_abs, const REAL *vc)
{
    FN(consts) kk = FN(mk)(X, Y, 1.0, dx, 1.0); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 0) {
                REAL w = FN(diff_x)(k, vc, 2, i, j, 1) - FN(diff_y)(k, vc, 2, i, j, 0);
                AT(vort, 1, i, j, 0) = w;
                AT(vort_abs, 1, i, j, 0) = FABS(w);
            }
}

/* K6  _add_vorticity + _vorticity_vec   (vorticity_confinement.py:34-55).
 * H4: |grad| == 0 gives 0/0 = NaN, which the NaN-ignoring min/max turn into +0.1. */
void FN(oracle_vort_add)(int X, int Y, double dt, double dx, double weight, const uint8_t *mask,
                         REAL *vn, const REAL *vc, const REAL *vort, const REAL *vort_abs)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, 1.0); const FN(consts) *k = &kk;
    const REAL dtw = (REAL)(dt * weight);        /* self.dt * self.weight folded in double */
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) == 0) {
                REAL gx = FN(diff_x)(k, vort_abs, 1, i, j, 0), gy = FN(diff_y)(k, vort_abs, 1, i, j, 0);
                REAL nrm = SQRT(gx * gx + gy * gy);
                gx = gx / nrm; gy = gy / nrm;
                REAL w = AT(vort, 1, i, j, 0);
                REAL f0 = gy * w, f1 = (-gx) * w;
                f0 = FMAX(FMIN(f0, (REAL)0.1), (REAL)-0.1);
                f1 = FMAX(FMIN(f1, (REAL)0.1), (REAL)-0.1);
                AT(vn, 2, i, j, 0) = AT(vc, 2, i, j, 0) + dtw * f0;
                AT(vn, 2, i, j, 1) = AT(vc, 2, i, j, 1) + dtw * f1;
            }
}

/* F1  predict_p   (pressure_updater.py:23-38) - literal operation order */
static inline REAL FN(predict_p)(const FN(consts) *k, const REAL *pc, const REAL *vc, int i, int j)
{
    int X = k->X, Y = k->Y;
    REAL sxx = S(vc, 2, i + 1, j, 0) - S(vc, 2, i - 1, j, 0);
    REAL sxy = S(vc, 2, i + 1, j, 1) - S(vc, 2, i - 1, j, 1);
    REAL syx = S(vc, 2, i, j + 1, 0) - S(vc, 2, i, j - 1, 0);
    REAL syy = S(vc, 2, i, j + 1, 1) - S(vc, 2, i, j - 1, 1);
    REAL t1 = (REAL)0.25 * (((S(pc, 1, i + 1, j, 0) + S(pc, 1, i - 1, j, 0)) + S(pc, 1, i, j + 1, 0)) + S(pc, 1, i, j - 1, 0));
    REAL t2 = ((sxx * sxx + syy * syy) + (syx * sxy)) / (REAL)8.0;
    REAL t3 = (k->dx * (sxx + syy)) / k->eight_dt;
    return (t1 + t2) - t3;
}

/* K8J  JacobiPressureUpdater._update   (pressure_updater.py:62-66), not-wall cells */
void FN(oracle_jacobi_sweep)(int X, int Y, double dt, double dx, const uint8_t *mask,
                             REAL *pn, const REAL *pc, const REAL *vc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, 1.0); const FN(consts) *k = &kk;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (M(i, j) != 1) AT(pn, 1, i, j, 0) = FN(predict_p)(k, pc, vc, i, j);
}

/* K8R  RedBlackSorPressureUpdater._update_pressures_odd/_even + _pn_ij (pressure_updater.py:98-114).
 * parity 1 = odd pass, 0 = even pass; pn may alias pc (the even pass runs in place on p.next). */
void FN(oracle_rbsor_half)(int X, int Y, double dt, double dx, double omega, int parity, const uint8_t *mask,
                           REAL *pn, const REAL *pc, const REAL *vc)
{
    FN(consts) kk = FN(mk)(X, Y, dt, dx, 1.0); const FN(consts) *k = &kk;
    const REAL om1 = (REAL)(1.0 - omega), om = (REAL)omega;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            if (((i + j) & 1) == parity && M(i, j) == 0)
                AT(pn, 1, i, j, 0) = om1 * AT(pc, 1, i, j, 0) + om * FN(predict_p)(k, pc, vc, i, j);
}

/* K9  limit_field   (solver.py:38-43) */
void FN(oracle_limit_field)(int X, int Y, double limit, REAL *v)
{
    const REAL lim = (REAL)limit;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j) {
            REAL x = AT(v, 2, i, j, 0), y = AT(v, 2, i, j, 1);
            REAL nrm = SQRT(x * x + y * y);
            if (nrm > lim) {
                AT(v, 2, i, j, 0) = lim * (x / nrm);
                AT(v, 2, i, j, 1) = lim * (y / nrm);
            }
        }
}

/* K13  clamp_field   (solver.py:46-49) */
void FN(oracle_clamp_field)(int X, int Y, int C, double low, double high, REAL *f)
{
    const REAL lo = (REAL)low, hi = (REAL)high;
    #pragma omp parallel for schedule(static)
    for (int i = 0; i < X; ++i)
        for (int j = 0; j < Y; ++j)
            for (int c = 0; c < C; ++c) AT(f, C, i, j, c) = FMIN(FMAX(AT(f, C, i, j, c), lo), hi);
}

#undef M
#undef S
#undef AT
#undef FN
#undef CAT
#undef CAT_
