/*
 * fs_hip.h - C-ABI of libfs_hip.so: the MI355X (gfx950) implementation of the
 * FluidSimulator.step() hot path of takah29/2d-fluid-simulator.
 *
 * The reference has no FFI: its hot path is a set of Taichi kernels behind plain Python classes
 * (SURVEY.md 8b).  This header is the boundary a maintainer would bind instead of Taichi: one entry
 * point per reference kernel, taking opaque device-field handles, so that the Python classes keep
 * the reference's own orchestration (and its DoubleBuffer swap semantics).  Each declaration cites
 * the reference code it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - every function returns 0 on success, a negative fs_status on failure; fs_last_error() gives text.
 *   - all launches are asynchronous on the context's HIP stream; fs_sync()/downloads synchronise.
 *   - host arrays use the reference layout: shape (X, rows[, C]), y contiguous, channels innermost
 *     (what Taichi's to_numpy()/from_numpy() exchange: fs/fluid_simulator.py:34-36,
 *     fs/boundary_condition.py:78-85).  Device layout is private (row-major in x).
 *   - dtype: 0 = f32 (the reference's only precision, fs/double_buffer.py:7-11), 1 = f64.
 *   - a context owns one slab of the grid: global rows [y0, y0 + ny_local) plus `halo` ghost rows on
 *     each side (single GPU: y0 = 0, ny_local = ny).  "local row" r maps to global y = y0 - halo + r.
 *   - kernels take a local row range [row_begin, row_end) to compute on; reads reach up to 2 rows
 *     outside it (clamped at the global domain edge like fs/differentiation.py:4-9 sample()).
 */
#ifndef FS_HIP_H
#define FS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 9

typedef struct fs_ctx fs_ctx;
typedef struct fs_field fs_field;

enum fs_status {
    FS_OK = 0,
    FS_ERR_ARG = -1,      /* bad argument / shape / dtype mismatch        */
    FS_ERR_HIP = -2,      /* HIP runtime error (text in fs_last_error)    */
    FS_ERR_STATE = -3,    /* call order (e.g. kernel before mask upload)  */
    FS_ERR_COMM = -4,     /* RCCL error / communicator not initialised    */
    FS_ERR_UNSUPPORTED = -5
};

enum fs_scheme { FS_UPWIND = 0, FS_KK = 1 };   /* fs/advection.py:12-24 / :27-60 */

int fs_abi_version(void);
const char *fs_last_error(void);
int fs_device_count(int *count);
/* Compact launch lists of this context: how many exist (one hipMalloc + stream synchronisation each, at the first launch of a geometry / slab row
 * range), and how many launches wanted one they could not build (inside a hipGraph capture / tape recording, or beyond 512 lists) and ran dense. */
int fs_tile_list_stats(const fs_ctx *ctx, int *built, int *misses);

/* ---- context -------------------------------------------------------------------------------- */
/* replaces ti.init(...) + Taichi's field allocator (main.py:65-69). */
int fs_create(fs_ctx **out, int device, int nx, int ny, int dtype, int y0, int ny_local, int halo);
int fs_destroy(fs_ctx *ctx);
int fs_sync(fs_ctx *ctx);
int fs_ctx_info(const fs_ctx *ctx, int *nx, int *ny, int *dtype, int *y0, int *ny_local, int *halo, int *pitch);

/* ---- scene (BoundaryCondition.to_field, fs/boundary_condition.py:78-85, :101-112) ----------- */
/* Arrays are GLOBAL: mask (X, Y) u8; bc_const (X, Y, 2); bc_dye (X, Y, 3) of the ctx dtype.      */
int fs_upload_mask(fs_ctx *ctx, const uint8_t *mask_xy);
int fs_upload_bc_const(fs_ctx *ctx, const void *bc_xy2);
int fs_upload_bc_dye(fs_ctx *ctx, const void *bc_xy3);
/* Stencil radius in rows of the velocity / pressure boundary kernels on this mask: how far (chains of the
 * reference's in-kernel read-after-write hazards included) a rewritten cell depends on pre-kernel data.
 * 2 and 1 for regular scenes; slab runs need halo >= these.                                            */
int fs_bc_radius(const fs_ctx *ctx, int *velocity_rows, int *pressure_rows);

/* ---- fields (ti.field / ti.Vector.field + DoubleBuffer members, fs/double_buffer.py:4-18) ---- */
int fs_field_alloc(fs_ctx *ctx, int nchan, fs_field **out);           /* zero-initialised */
int fs_field_free(fs_field *f);
int fs_field_fill(fs_field *f, double value);                          /* field.fill()     */
int fs_field_nchan(const fs_field *f);
/* from_numpy / to_numpy on a window of local rows: host shape (X, nrows, C). */
int fs_field_upload(fs_field *f, const void *host_xrc, int row_begin, int nrows);
int fs_field_download(const fs_field *f, void *host_xrc, int row_begin, int nrows);
int fs_field_copy(fs_field *dst, const fs_field *src);
/* raw device pointer + geometry, for zero-copy interop (hipMemcpy peers, profilers). */
/* Is the velocity buffer's "may hold a speed above 9.95" flag up?  (Synchronises; lets the host take limit_field as its own launch again in a run that has gone hot.) */
int fs_field_hot(const fs_field *f, int *hot);
int fs_field_devptr(const fs_field *f, void **ptr, size_t *bytes);

/* ---- boundary-condition kernels -------------------------------------------------------------- */
/* BoundaryCondition.set_velocity_boundary_condition   fs/boundary_condition.py:16-39  (in place) */
int fs_velocity_bc(fs_ctx *ctx, fs_field *v, int row_begin, int row_end);
/* limit_field of step n + set_velocity_boundary_condition of step n+1 in ONE launch (new: fs/solver.py:38-43 then fs/boundary_condition.py:16-39;
 * the reference issues them as the last kernel of one step and the first of the next).  Behind the velocity buffer's flag the limit pass
 * does nothing in a healthy run, yet its launch is a fifth of a small-grid step; the Python shell defers it (the field carries a pending
 * limit until anything else looks at it) and this entry point runs gate + limit (rare) + boundary op list.  Same bits as fs_limit_field
 * over [limit_begin, limit_end) followed by fs_velocity_bc over [row_begin, row_end).  fs_velocity_bc_limit_ok: available for this context.
 * `parity` (0 / 1): consecutive calls on ONE field alternate it - a launch's op list raises the flag word of its parity and its gate reads
 * the other one, so that every workgroup of a launch takes the same decision (a call that repeats its predecessor's parity would see what
 * that predecessor's op list raised one call late).  fs_field_copy / fs_field_fill / fs_limit_field read all words: any order there. */
int fs_velocity_bc_limit_ok(const fs_ctx *ctx, int *ok);
int fs_velocity_bc_limit(fs_ctx *ctx, double limit, fs_field *v, int parity, int limit_begin, int limit_end, int row_begin, int row_end);
/* BoundaryCondition.set_pressure_boundary_condition   fs/boundary_condition.py:41-65  (in place) */
int fs_pressure_bc(fs_ctx *ctx, fs_field *p, int row_begin, int row_end);
/* DyeBoundaryCondition.set_dye_boundary_condition     fs/boundary_condition.py:94-99  (in place) */
int fs_dye_bc(fs_ctx *ctx, fs_field *dye, int row_begin, int row_end);
/* limit_field(v) at the end of the flow step + set_dye_boundary_condition in ONE launch (the dye solvers: fs/solver.py:148-155, 385-392 call
 * them back to back).  Same bits as fs_limit_field(v) over [limit_begin, limit_end) followed by fs_dye_bc(dye) over [row_begin, row_end). */
int fs_dye_bc_limit_ok(const fs_ctx *ctx, int *ok);
int fs_dye_bc_limit(fs_ctx *ctx, double limit, fs_field *v, fs_field *dye, int limit_begin, int limit_end, int row_begin, int row_end);

/* ---- velocity / dye transport ----------------------------------------------------------------- */
/* MacSolver._update_velocities        fs/solver.py:94-107   (fluid cells; scheme = fs_scheme)   */
int fs_mac_update(fs_ctx *ctx, int scheme, double dt, double dx, double re,
                  fs_field *vn, const fs_field *vc, const fs_field *pc, int row_begin, int row_end);
/* DyeMacSolver._update_dye            fs/solver.py:157-161                                      */
int fs_mac_dye(fs_ctx *ctx, int scheme, double dt, double dx,
               fs_field *dn, const fs_field *dc, const fs_field *vc, int row_begin, int row_end);
/* CipMacSolver._set_grad              fs/solver.py:207-211  (all cells)                         */
int fs_cip_set_grad(fs_ctx *ctx, double dx, fs_field *fx, fs_field *fy, const fs_field *f,
                    int row_begin, int row_end);
/* CipMacSolver._non_advection_phase   fs/solver.py:229-240  (not-wall cells)                    */
int fs_cip_nonadv(fs_ctx *ctx, double dt, double dx, double re,
                  fs_field *fn, const fs_field *fc, const fs_field *pc, int row_begin, int row_end);
/* DyeCipMacSolver._non_advection_phase_dye  fs/solver.py:378-383                                */
int fs_cip_nonadv_dye(fs_ctx *ctx, double dt, double dx, double re,
                      fs_field *dn, const fs_field *dc, int row_begin, int row_end);
/* CipMacSolver._non_advection_phase_grad    fs/solver.py:242-261  (C = 2 or 3 channels)         */
int fs_cip_nonadv_grad(fs_ctx *ctx, double dx, fs_field *fxn, fs_field *fyn,
                       const fs_field *fxc, const fs_field *fyc, const fs_field *fc, const fs_field *fn,
                       int row_begin, int row_end);
/* CipMacSolver._advection_phase / _cip_advect  fs/solver.py:267-332  (v = advecting velocity)   */
int fs_cip_advect(fs_ctx *ctx, double dt, double dx, fs_field *fn, fs_field *fxn, fs_field *fyn,
                  const fs_field *fc, const fs_field *fxc, const fs_field *fyc, const fs_field *v,
                  int row_begin, int row_end);

/* _non_advection_phase_grad + _advection_phase of the VELOCITY field fused into one pass (build-side optimisation, same
 * bits for everything observable): the intermediate gradients never go through HBM.  fn = velocity after
 * _non_advection_phase, fc = velocity before it, gxc/gyc = gradients before; outputs: v_out (advected value on fluid cells, fc
 * carried elsewhere - what the reference's in-place update of fc's buffer leaves), gx_out / gy_out (not-wall cells).  v_out must
 * be a third buffer distinct from fn and fc; the caller rotates buffers (see fs/solver.py).  Tiles without a fluid cell carry only
 * the cells some kernel writes (not-wall cells and the targets of the velocity boundary kernel): every other cell is equal in fc and
 * v_out unless somebody uploaded into one of them - then ONE call with full != 0 carries every cell.                           */
int fs_cip_grad_advect(fs_ctx *ctx, double dt, double dx, fs_field *v_out, fs_field *gx_out, fs_field *gy_out,
                       const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc,
                       int full, int row_begin, int row_end);
/* CipMacSolver._update_velocities (fs/solver.py:213-227) as ONE call: _non_advection_phase (:229-240; fn <- fc, pc on the not-wall cells)
 * followed by the fused pass above.  Same results in v_out / gx_out / gy_out and in every cell of fn that anything reads before it is
 * rewritten.  On f32 grids of even width (wherever the compact launch lists exist) the post-K2 velocity is evaluated in registers on the way (csrc/fs_k234.h: one launch over every
 * tile - FS_FUSE_K2=1: one over the tiles that see nothing but fluid, one over the others) and stored only on the not-wall cells that are not
 * fluid (inflow / outflow): the fluid cells of fn keep their old content - which the reference's own sequence overwrites (K2 of the next step,
 * or this step's vorticity confinement) before reading it.  fs_cip_step_ok: whether calls take that form (f32, FS_FUSE_K2 != 0, the size; one GPU:
 * calls over the whole grid; a slab context: any row range - K2 is then evaluated for the rows within 2 of the range from rows within 3 of it,
 * which the caller keeps valid, where the two calls would read what an earlier K2 launch left in fn); otherwise the call is exactly
 * fs_cip_nonadv + fs_cip_grad_advect (on a slab: fs_cip_nonadv over the range widened by 2 rows).                                    */
int fs_cip_step_ok(const fs_ctx *ctx, int *ok);
/* The same for the dye: DyeCipMacSolver._update_dye (fs/solver.py:385-401) as ONE call = fs_cip_nonadv_dye(fn <- fc) + fs_cip_grad_advect_dye, with
 * K12 evaluated in registers under the conditions of fs_cip_step_ok (csrc/fs_k234.h k_cip_dye). */
int fs_cip_step_dye(fs_ctx *ctx, double dt, double dx, double re, fs_field *d_out, fs_field *gx_out, fs_field *gy_out, fs_field *fn,
                    const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v, int clamp01, int full, int row_begin, int row_end);
/* diagnostic: the tiles (tile_rows x tile_cells cells each) of the two classes of a whole-grid fs_cip_step launch - all-fluid tiles, the others
 * (`band`: always 0 since ABI 8's three-part form is gone); 0 0 0 where the call takes the two-call form. */
int fs_cip_step_tiles(fs_ctx *ctx, int *plain, int *boundary, int *band, int *tile_rows, int *tile_cells);
int fs_cip_step(fs_ctx *ctx, double dt, double dx, double re, fs_field *v_out, fs_field *gx_out, fs_field *gy_out, fs_field *fn,
                const fs_field *fc, const fs_field *pc, const fs_field *gxc, const fs_field *gyc, int full, int row_begin, int row_end);
/* The same for the dye (C = 3, advected by the velocity field v of the finished flow step; fs/solver.py:378-401 _update_dye without its
 * first kernel): K3 (_non_advection_phase_grad, :242-261) + K4 (_cip_advect, :267-332), d_out = a third dye buffer the caller rotates.
 * clamp01 != 0 folds clamp_field(dye, 0, 1) (:46-49) into the store of the advected cells; full as above.  f32 only.                   */
int fs_cip_grad_advect_dye(fs_ctx *ctx, double dt, double dx, fs_field *d_out, fs_field *gx_out, fs_field *gy_out,
                           const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v,
                           int clamp01, int full, int row_begin, int row_end);

/* ---- vorticity confinement -------------------------------------------------------------------- */
/* VorticityConfinement._calc_vorticity    fs/vorticity_confinement.py:27-32                     */
int fs_vort_calc(fs_ctx *ctx, double dx, fs_field *vort, fs_field *vort_abs, const fs_field *vc,
                 int row_begin, int row_end);
/* VorticityConfinement._add_vorticity     fs/vorticity_confinement.py:34-55                     */
int fs_vort_add(fs_ctx *ctx, double dt, double dx, double weight, fs_field *vn, const fs_field *vc,
                const fs_field *vort, const fs_field *vort_abs, int row_begin, int row_end);

/* _calc_vorticity + _add_vorticity fused into one pass (build-side optimisation, same bits): the vorticity never
 * goes through HBM.  vort / vort_abs may be NULL; when given they are also written (fluid cells), like K5 does. */
int fs_vort_confine(fs_ctx *ctx, double dt, double dx, double weight, fs_field *vn, const fs_field *vc,
                    fs_field *vort, fs_field *vort_abs, int row_begin, int row_end);

/* ---- pressure Poisson relaxation (predict_p, fs/pressure_updater.py:23-38) -------------------- */
/* JacobiPressureUpdater._update           fs/pressure_updater.py:62-66   (not-wall cells)       */
int fs_jacobi_sweep(fs_ctx *ctx, double dt, double dx, fs_field *pn, const fs_field *pc,
                    const fs_field *vc, int row_begin, int row_end);
/* RedBlackSorPressureUpdater._update_pressures_odd/_even  fs/pressure_updater.py:98-114.
 * parity 1 = odd cells ((i + j) % 2 == 1), 0 = even; pn may be the same field as pc (even pass).  */
int fs_rbsor_halfsweep(fs_ctx *ctx, double dt, double dx, double omega, int parity,
                       fs_field *pn, const fs_field *pc, const fs_field *vc, int row_begin, int row_end);
/* One whole red-black iteration (odd pass pc -> pn, then even pass in place on pn; fs/pressure_updater.py:92-96)
 * as a single fused kernel (build-side optimisation, same bits, 21 instead of 34 B/cell).  pn != pc.   */
int fs_rbsor_iteration(fs_ctx *ctx, double dt, double dx, double omega, fs_field *pn, const fs_field *pc,
                       const fs_field *vc, int row_begin, int row_end);
/* Source-term precompute (build-side optimisation; the source of predict_p depends only on v and is
 * constant over the sweeps of one step).  src has 2 channels: (s2, s3) with predict_p = (0.25*sum + s2) - s3,
 * i.e. the reference's operation order is kept and results stay bit-identical to the v-reading kernels.  src is defined on every
 * cell a sweep reads it at (not-wall cells and everything within a few cells of one); regions of nothing but wall may be left untouched. */
int fs_poisson_source(fs_ctx *ctx, double dt, double dx, fs_field *src, const fs_field *vc,
                      int row_begin, int row_end);
int fs_jacobi_sweep_src(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src,
                        int row_begin, int row_end);
/* Jacobi sweep on the source pair with the pressure boundary condition evaluated on the fly from the RAW output of the previous
 * sweep (build-side optimisation for long Jacobi runs, same bits): replaces fs_pressure_bc + fs_jacobi_sweep_src for all but the
 * last two sweeps of a JacobiPressureUpdater.update (fs/pressure_updater.py:56-66).  fs_lazy_bc_ok: does this mask admit it
 * (every K7 source a not-wall cell, no computed cell in the first / last row)?                                                */
int fs_lazy_bc_ok(const fs_ctx *ctx, int *ok);
int fs_jacobi_sweep_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end);
/* Diagnostic: the classification the two kernels above and below work from, one byte per (wave column of 248 cells, local row),
 * [wave column][row]; bit 0: a computed cell of the row has a K7 target among its 4 neighbours, 1: a wall / target within 2 columns,
 * 3: a target that reads another row within 4 columns, 2: such a target in a wall one cell thick, or a wall cell whose content is
 * history, 4 / 5: the row goes through the general path of fs_jacobi_pair_lazy with mode bit 1 = 0 / 1 (general_rows[0 / 1] = how many
 * such rows hold computed cells; an int[2]).  out may be NULL to query the sizes.                                                    */
int fs_lazy_flags(fs_ctx *ctx, uint8_t *out, int capacity, int *wave_columns, int *rows, int *general_rows);
/* TWO such sweeps in one pass: pn <- sweep(sweep(pc)), the first sweep's rows staying in registers.  mode bit 0 ("swapped"): the two
 * buffers of the reference's rotation differ in the wall cells nothing ever writes, and not-wall cells beside them read them - 0 when pc
 * is the physical buffer the reference holds this pass's input iterate in (the 1st, 3rd ... pass of a pc -> pn -> pc sequence; the
 * intermediate iterate's wall cells are then pn's), 1 when it is the other way round (2nd, 4th ... pass).  mode bit 1: the tile path also
 * applies the recipes that read the row below / above (pays on masks with staircase outlines, costs a few percent elsewhere; same bits).
 * Both buffers are read 4 rows beyond the written range, src 2 rows.                                                                  */
int fs_jacobi_pair_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int mode, int row_begin, int row_end);
/* Diagnostic.  f32 runs divide by their loop-invariant divisors (dx, 2 dx, dx^2, dx^3, 6 dx, 8 dt, Re - the divisions of
 * fs/differentiation.py:41-60, fs/solver.py:257-260, 304-313, fs/advection.py:46-58, fs/pressure_updater.py:37) as
 * (float)((double)x * (1.0 / d)), which equals the IEEE quotient x / d for every f32 x (csrc/fs_device.h f64div).  This checks the
 * identity on the device for one divisor: ~2^28 dividends (all significands of nine binades incl. the denormals, both signs, and
 * arbitrary bit patterns); *mismatches must come back 0.                                                                        */
int fs_selftest_f64div(fs_ctx *ctx, double divisor, int *mismatches);
/* TWO red-black iterations - RedBlackSorPressureUpdater.update with n_iter = 2, fs/pressure_updater.py:86-96, which is what
 * fs/fluid_simulator.py:76-78 wires into every create() - and the two set_pressure_boundary_condition passes between them
 * (fs/boundary_condition.py:41-65) in ONE pass over HBM (build-side optimisation, same bits): (pc_out, pn_out) receive what the
 * reference leaves in (p.current, p.next) after two iterations that start from (pc, pn).  Out of place: four distinct fields; the caller
 * rotates the pairs.  Only fluid cells and boundary-condition targets are stored (every other cell holds the same value in all four
 * buffers as long as nobody uploads into them); full != 0 stores every cell of the row range - the carry pass after an upload.  pc and
 * vc are read 4 rows beyond the written range, pn 3.  fs_rbsor_pair_ok: f32, X % 4 == 0, and a mask without one-cell-thin walls between
 * fluid regions and without fluid in the first / last row (csrc/fs_rbpair.h); otherwise FS_ERR_UNSUPPORTED - use fs_rbsor_iteration. */
int fs_rbsor_pair_ok(const fs_ctx *ctx, int *ok);
int fs_rbsor_pair(fs_ctx *ctx, double dt, double dx, double omega, fs_field *pc_out, fs_field *pn_out, const fs_field *pc,
                  const fs_field *pn, const fs_field *vc, int full, int row_begin, int row_end);
/* FOUR such sweeps in one pass (csrc/fs_jquad.h): pn[not wall] <- sweep(sweep(sweep(sweep(pc)))), the boundary condition evaluated in
 * front of each; for long Jacobi runs whose data stay cache resident and whose cost is launches and latency.  Writes the not-wall cells
 * of pn only: the caller guarantees that the wall cells no kernel writes are equal in pn and pc (nobody uploaded into one of them).
 * pc is read 4 rows beyond the written range, src 3.  fs_jacobi_quad_ok: f32, fs_lazy_bc_ok, and no recipe that reads the far side of
 * its target as seen from a cell whose value is used; otherwise FS_ERR_UNSUPPORTED - use fs_jacobi_pair_lazy / fs_jacobi_sweep_lazy.  */
int fs_jacobi_quad_ok(const fs_ctx *ctx, int *ok);
int fs_jacobi_quad_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end);
/* The LAST two rounds (K7, sweep, swap; K7, sweep, swap - fs/pressure_updater.py:56-66) of such a run in one pass: pc holds the raw iterate
 * n-2; pc_out (a third buffer) receives what the reference leaves in p.current (iterate n on the not-wall cells, K7(iterate n-2) on the wall
 * cells with a recipe), pn what it leaves in p.next (iterate n-1 with K7 applied).  Cells no kernel writes are not stored: they must be
 * equal in the three buffers.  Same conditions as fs_jacobi_quad_lazy; pc is read 2 rows beyond the written range, src 1. */
int fs_jacobi_finish(fs_ctx *ctx, fs_field *pc_out, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end);
int fs_rbsor_halfsweep_src(fs_ctx *ctx, double omega, int parity, fs_field *pn, const fs_field *pc,
                           const fs_field *src, int row_begin, int row_end);
/* Residual diagnostic (new; the reference never measures convergence): sum over owned not-wall cells of
 * (predict_p(p) - p)^2 and the cell count, accumulated in double.  Does not alter any field.       */
int fs_poisson_residual(fs_ctx *ctx, double dt, double dx, const fs_field *p, const fs_field *vc,
                        double *sum_sq, double *count);

/* ---- pointwise -------------------------------------------------------------------------------- */
/* limit_field   fs/solver.py:38-43 ;  clamp_field   fs/solver.py:46-49                           */
int fs_limit_field(fs_ctx *ctx, double limit, fs_field *v, int row_begin, int row_end);
int fs_clamp_field(fs_ctx *ctx, double low, double high, fs_field *f, int row_begin, int row_end);

/* Dye transport with the final clamp_field(dye, 0, 1) folded in (build-side optimisation, same bits): CIP advection of the
 * 3-channel dye that stores clamp(value, 0, 1) on fluid cells, and the clamp of the inflow cells (the only other cells the
 * step leaves outside [0, 1]).  Together they replace fs_cip_advect + fs_clamp_field over the whole grid.                    */
int fs_cip_advect_dye_clamped(fs_ctx *ctx, double dt, double dx, fs_field *fn, fs_field *fxn, fs_field *fyn,
                              const fs_field *fc, const fs_field *fxc, const fs_field *fyc, const fs_field *v,
                              int row_begin, int row_end);
int fs_clamp_inflow(fs_ctx *ctx, double low, double high, fs_field *dye, int row_begin, int row_end);

/* ---- visualisation (the image buffers main.py:93-107 shows; all cells, wall cells take the wall colour) ---------------- */
/* FluidSimulator._to_norm        fs/fluid_simulator.py:38-44   0.2 * visualize_norm(v) + 0.002 * visualize_pressure(p)    */
int fs_vis_norm(fs_ctx *ctx, fs_field *rgb, const fs_field *v, const fs_field *p, int row_begin, int row_end);
/* FluidSimulator._to_pressure    fs/fluid_simulator.py:46-51   0.04 * visualize_pressure(p)   (fs/visualization.py:14-16) */
int fs_vis_pressure(fs_ctx *ctx, fs_field *rgb, const fs_field *p, int row_begin, int row_end);
/* FluidSimulator._to_vorticity   fs/fluid_simulator.py:53-58   0.005 * visualize_vorticity(v) (fs/visualization.py:19-22) */
int fs_vis_vorticity(fs_ctx *ctx, double dx, fs_field *rgb, const fs_field *v, int row_begin, int row_end);
/* DyeFluidSimulator._to_dye      fs/fluid_simulator.py:121-126                                                            */
int fs_vis_dye(fs_ctx *ctx, fs_field *rgb, const fs_field *dye, int row_begin, int row_end);

/* ---- multi-GPU: y-slab halo exchange over RCCL (new; the reference is single-device) ---------- */
#define FS_UNIQUE_ID_BYTES 128
/* librccl loads and exports what the exchange needs (*ok; no GPU call): the pre-flight check of an N > 1 job, before any rank can block in
 * ncclCommInitRank.  *ok = 0: the reason is in fs_last_error(). */
int fs_comm_available(int *ok);
int fs_comm_unique_id(void *out_128_bytes);
int fs_comm_init(fs_ctx *ctx, int rank, int nranks, const void *unique_id_128_bytes);
int fs_comm_destroy(fs_ctx *ctx);
/* Fill `depth` ghost rows on each side from the slab neighbours (ncclSend/ncclRecv pairs).  All RCCL calls of a context
 * run on its own communication stream; these two are begin + wait (see below).                                          */
int fs_halo_exchange(fs_ctx *ctx, fs_field *f, int depth);
/* Same for several fields in ONE grouped RCCL call: their ghost-row blocks travel as one packed message per neighbour.   */
int fs_halo_exchange_multi(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth);
/* Split form for overlap: begin() queues the exchange on the communication stream after everything already queued on the
 * compute stream; until wait() the caller may launch kernels that neither read ghost rows nor write the `depth` outermost
 * owned rows of these fields (the interior rows of the kernel that needed the exchange).  wait() orders the compute
 * stream after the exchange.  One exchange in flight per context.                                                        */
int fs_halo_exchange_begin(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth);
/* As begin(), but field k's ghost rows are known to be correct to depth valid_rows[k] already: only depth offsets
 * [valid_rows[k], depth) travel (same numbers on both sides of a slab boundary; the host tracker has them).          */
int fs_halo_exchange_begin_partial(fs_ctx *ctx, fs_field *const *fields, const int *valid_rows, int nfields, int depth);
int fs_halo_exchange_wait(fs_ctx *ctx);
/* Optional, before begin(): the exchange will depend on the compute stream as of NOW - kernels launched between mark() and
 * begin() (same restrictions as above) are already running while the host still issues the exchange.                     */
int fs_halo_exchange_mark(fs_ctx *ctx);
/* Exchanges in line on the compute stream (0, default) or on the communication stream (1: overlappable; FS_OVERLAP=1 at fs_comm_init).  Same
 * bits either way; a tape recorded under one setting is replayed under it. */
int fs_comm_set_overlap(fs_ctx *ctx, int on);
/* Loop-back self-test on a 1-rank communicator: the rank is its own lower and upper neighbour, so afterwards
 * lower ghost rows == first owned rows and upper ghost rows == last owned rows (single-GPU check of the RCCL leg).
 * fs_comm_loopback(ctx, 1) makes every later exchange of a 1-rank communicator behave that way.                          */
int fs_halo_exchange_self(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth);
int fs_comm_loopback(fs_ctx *ctx, int on);
int fs_allreduce_sum(fs_ctx *ctx, double *values, int n);

/* ---- launch-overhead removal: capture the launches issued between begin/end into a hipGraph ---- */
int fs_graph_begin(fs_ctx *ctx);
int fs_graph_end(fs_ctx *ctx, int *graph_id);
int fs_graph_launch(fs_ctx *ctx, int graph_id, int times);
int fs_graph_free(fs_ctx *ctx, int graph_id);

/* ---- command tapes: replay of a recorded launch sequence INCLUDING the RCCL ghost-row exchanges (slab runs) ---------------
 * Between begin and end every kernel entry point and fs_halo_exchange_begin* / _mark / _wait appends a closure holding its
 * arguments (execute = 1: and runs as usual; execute = 0: records only).  fs_tape_replay re-issues the closures `times` times
 * from a C++ loop - the fused driver of SURVEY.md 8b for N > 1, where a hipGraph cannot carry the exchange.  The caller
 * guarantees that the recorded sequence is valid to repeat (fs/runtime.py records a whole period of the buffer rotation and
 * of its ghost-row bookkeeping).                                                                                          */
int fs_tape_begin(fs_ctx *ctx, int execute);
int fs_tape_end(fs_ctx *ctx, int *tape_id);
int fs_tape_length(fs_ctx *ctx, int tape_id, int *nops);
int fs_tape_replay(fs_ctx *ctx, int tape_id, int times);
int fs_tape_free(fs_ctx *ctx, int tape_id);

/* Measurement hygiene (new; no reference counterpart): what THIS GPU streams at - float4 read of one buffer and float4 copy between two
 * buffers of `bytes` each, about budget_ms of GPU time per leg, HIP events on the context's stream.  bench.py prints the two rates and
 * `frac_of_box_copy` next to every roofline fraction, so that a slow box is not mistaken for a regression. */
int fs_box_rates(fs_ctx *ctx, size_t bytes, double budget_ms, double *read_GBps, double *copy_GBps);
/* ... and the rate at which one SIMD issues independent f32 multiplies / adds at 4 waves per SIMD, in 1e9 wave-instructions per second
 * (no memory traffic): what the issue-bound K3+K4 pass is priced against (bench.py roofline.valu_issue). */
int fs_box_valu_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd);
/* The same chains on packed operands (v_pk_mul_f32 / v_pk_add_f32: two f32 operations per lane and instruction) - what the packed bodies of
 * the transport kernels (csrc/fs_k34n.h, fs_k234.h) issue.  Wave-instructions per second and SIMD. */
int fs_box_valu_pk_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd);
/* ... and a float4 copy with 176 f32 multiplies / adds per 16 bytes on the way (5.5 lane-operations per byte moved: the instruction density of the K3+K4 pass): memory system and SIMDs loaded
 * together, in GB/s of read + written bytes. */
int fs_box_mixed_rate(fs_ctx *ctx, size_t bytes, double budget_ms, double *GBps);

/* ---- per-kernel timing with HIP events on the ctx stream (bench.py roofline leg) --------------- */
int fs_prof_enable(fs_ctx *ctx, int on);          /* record an event pair around every launch      */
/* One HIP-event pair on the context's stream around whatever is queued between the two calls (launch boundaries included); fs_span_end waits
 * for it and returns the milliseconds.  Independent of the per-launch profile. */
int fs_span_begin(fs_ctx *ctx);
int fs_span_end(fs_ctx *ctx, double *ms);
int fs_prof_reset(fs_ctx *ctx);
int fs_prof_count(fs_ctx *ctx, int *n);           /* number of distinct kernels seen (syncs)       */
int fs_prof_get(fs_ctx *ctx, int idx, char *name, int name_cap, int *launches, double *total_ms);
/* The __global__ functions launched under profile name `name` since fs_prof_enable, demangled ("fs::k_jacobi_ov2<4, 4>"), one per line in
 * `out` (may be null), their number in *n: what a rocprofv3 --kernel-trace of the same run lists.  bench.py names the kernels it prices
 * from here instead of from string literals. */
int fs_prof_kernels(fs_ctx *ctx, const char *name, char *out, int capacity, int *n);

#ifdef __cplusplus
}
#endif
#endif /* FS_HIP_H */
