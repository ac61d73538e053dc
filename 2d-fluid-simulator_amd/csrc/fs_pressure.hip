// fs_pressure.hip - C-ABI entry points of the pressure kernels: Jacobi sweeps (single, lazily bounded pairs / quads / finishing pass), red-black
// SOR (half sweeps, one fused iteration, two iterations per pass), the Poisson source pair and the residual.
#include "fs_launch.h"

using namespace fs;

#define FS_PAIR(RT) FS_KLAUNCH((k_jacobi_pair<RT, SW, HV, T>), grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (const uint8_t *)ctx->d_lazyflags, list, nlist, zoff, (T *)pn->d, (const T *)pc->d, (const T *)src->d)
template <bool SW, bool HV, typename T>
static void launch_pair(fs_ctx *ctx, const OvGrid &og, int rt, int row_begin, int row_end, fs_field *pn, const fs_field *pc, const fs_field *src)
{
    // the general rows ride in front: `zoff` leading z slices of the same launch, one wave per listed row
    const uint32_t *list = ctx->d_pairlist + (HV ? (size_t)ctx->nwx * ctx->rows : 0);
    const int nlist = ctx->n_pairlist[HV ? 1 : 0];
    const int per_slice = (int)(og.grid.x * og.grid.y), blocks = nlist, zoff = (blocks + per_slice - 1) / per_slice;      // one listed row per workgroup
    const dim3 grid(og.grid.x, og.grid.y, og.grid.z + zoff);
    if (rt == 2) FS_PAIR(2); else FS_PAIR(3);
}

template <bool SRC, typename T>
static int launch_jacobi(fs_ctx *ctx, const char *name, const Konst<T> &k, int jb, int je, T *pn, const T *pc, const T *vs)
{
    if constexpr (!SRC && std::is_same<T, float>::value) {
        // the literal f32 sweep on packed lanes of 2 cells, 4-row tiles, per-wave plain hints in the launch list (fs_jquad.h k_jacobi_ov2; round 5:
        // 74.5-75.6 against 81.6 us for the quad form below - 8-row tiles 86, 2-row tiles 84, without the hints 79-80)
        if (ctx->use_pairs) {
            const OvGrid og = ov_grid_lanes(ctx, jb, je, 4, 1, XCD_JACOBI, 3, true, 0, 1);      // (reach 1: the hints)
            const int dm = dm_const(ctx, k);
#define FS_JAC2(DM) FS_KLAUNCH((k_jacobi_ov2<4, DM>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs)
            return launch(ctx, name, [=] { FS_DMC(dm, FS_JAC2); });
        }
    }
    // overlapped-wave register tiles of quads: the source-pair form streams best with 1-row tiles at 8 waves/SIMD (76 vs 79 us), the v-reading form (f64)
    // with 4-row tiles (round 4: 84.7 against 85.9-86.4 us; the tile heights 2 and 3 of rounds 2 - 4 went with their switch in round 6)
    constexpr int RT = SRC ? 1 : 4;
    const OvGrid og = ov_grid(ctx, jb, je, RT, 1, XCD_JACOBI);
    const int dm = SRC ? 0 : dm_const(ctx, k);           // the source-pair form divides nothing
#define FS_JAC(DM) FS_KLAUNCH((k_jacobi_ov<SRC, RT, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs)
    return launch(ctx, name, [=] { FS_DMC(dm, FS_JAC); });
}

extern "C" {

int fs_jacobi_sweep(fs_ctx *ctx, double dt, double dx, fs_field *pn, const fs_field *pc, const fs_field *vc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(vc, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        // (odd res - X = 2 res not a multiple of 4: the f32 sweep on packed lanes of 2 cells needs an even width only, fs_jquad.h k_jacobi_ov2)
        if (ctx->use_march || (ctx->use_pairs && std::is_same<T, float>::value))
            return launch_jacobi<false, T>(ctx, "jacobi_sweep", k, row_begin, row_end, (T *)pn->d, (const T *)pc->d, (const T *)vc->d);
        FS_LAUNCH_CELLS("jacobi_sweep", (k_jacobi<false, T>), ctx->grid(), k, row_begin, (T *)pn->d, (const T *)pc->d, (const T *)vc->d)
    })
}

int fs_jacobi_sweep_src(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, 1.0, 1.0);
        if (ctx->use_march) return launch_jacobi<true, T>(ctx, "jacobi_sweep_src", k, row_begin, row_end, (T *)pn->d, (const T *)pc->d, (const T *)src->d);
        FS_LAUNCH_CELLS("jacobi_sweep_src", (k_jacobi<true, T>), ctx->grid(), k, row_begin, (T *)pn->d, (const T *)pc->d, (const T *)src->d)
    })
}

int fs_lazy_bc_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = ctx->mask_set && ctx->lazy_ok && ctx->use_march ? 1 : 0;
    return FS_OK;
}

// diagnostic: the per wave-tile-row flags of the lazy / two-sweep kernels (fs_march.h k_lazy_flags, k_pair_list), [wave column][local row]
int fs_lazy_flags(fs_ctx *ctx, uint8_t *out, int capacity, int *wave_columns, int *rows, int *general_rows)
{
    FS_REQUIRE(ctx && wave_columns && rows && general_rows, "null argument");
    FS_REQUIRE(ctx->mask_set && ctx->d_lazyflags, "no mask uploaded");
    *wave_columns = ctx->nwx; *rows = ctx->rows; general_rows[0] = ctx->n_pairlist[0]; general_rows[1] = ctx->n_pairlist[1];
    if (out) {
        FS_REQUIRE(capacity >= ctx->nwx * ctx->rows, "buffer too small");
        FS_HIP(hipMemcpyAsync(out, ctx->d_lazyflags, (size_t)ctx->nwx * ctx->rows, hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(hipStreamSynchronize(ctx->stream));
    }
    return FS_OK;
}

int fs_jacobi_sweep_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_ROWS();
    if (!(ctx->lazy_ok && ctx->use_march)) { set_error("this mask does not admit the lazy pressure boundary condition (fs_lazy_bc_ok)"); return FS_ERR_UNSUPPORTED; }
    const OvGrid og = ov_grid(ctx, row_begin, row_end, 1, 1, XCD_JACOBI);
    FS_DISPATCH(ctx, {
        return launch(ctx, "jacobi_sweep_lazy", [=] {
            FS_KLAUNCH((k_jacobi_lazy<T>), og.grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end,
                               (const uint8_t *)ctx->d_bcmap, (const uint8_t *)ctx->d_lazyflags, (T *)pn->d, (const T *)pc->d, (const T *)src->d);
        });
    })
}

// two lazily-bounded sweeps in one pass (fs_march.h k_jacobi_pair): pn <- sweep(sweep(pc)); pn's wall cells are read (the intermediate
// buffer of the two-buffer rotation is pn itself)
int fs_jacobi_pair_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int mode, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_REQUIRE(mode >= 0 && mode <= 3, "mode: bit 0 = swapped buffers, bit 1 = vertical recipes in the tile path");
    FS_ROWS();
    if (!(ctx->lazy_ok && ctx->use_march)) { set_error("this mask does not admit the lazy pressure boundary condition (fs_lazy_bc_ok)"); return FS_ERR_UNSUPPORTED; }
    const int rt = (mode & 2) ? 2 : 3;      // rows per tile: 3 is within 2 % of the best of 2 / 3 / 4 from res 1024 to 4096 (the third tile path at 3 rows: 97 VGPRs, one wave per SIMD less)
    const OvGrid og = ov_grid(ctx, row_begin, row_end, rt, 1, XCD_JACOBI, false);      // (dense: its general rows ride in leading z slices)
    FS_DISPATCH(ctx, {
        return launch(ctx, "jacobi_pair_lazy", [=] {
            switch (mode) {
            case 0: launch_pair<false, false, T>(ctx, og, rt, row_begin, row_end, pn, pc, src); break;
            case 1: launch_pair<true, false, T>(ctx, og, rt, row_begin, row_end, pn, pc, src); break;
            case 2: launch_pair<false, true, T>(ctx, og, rt, row_begin, row_end, pn, pc, src); break;
            default: launch_pair<true, true, T>(ctx, og, rt, row_begin, row_end, pn, pc, src); break;
            }
        });
    })
}

static inline dim3 rb_grid(const fs_ctx *c, int jb, int je) { return dim3(((c->X + 1) / 2 + 255) / 256, je - jb, 1); }

int fs_rbsor_halfsweep(fs_ctx *ctx, double dt, double dx, double omega, int parity, fs_field *pn, const fs_field *pc,
                       const fs_field *vc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(parity == 0 || parity == 1, "parity must be 0 or 1");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(vc, 2);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0, 0.0, omega);
        return launch(ctx, parity ? "rbsor_odd" : "rbsor_even", [=] {
            FS_KLAUNCH((k_rbsor<false, T>), rb_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k,
                               row_begin, parity, (T *)pn->d, (const T *)pc->d, (const T *)vc->d);
        });
    })
}

int fs_rbsor_iteration(fs_ctx *ctx, double dt, double dx, double omega, fs_field *pn, const fs_field *pc, const fs_field *vc,
                       int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(vc, 2);
    FS_REQUIRE(pn != pc, "the fused iteration needs distinct p.next / p.current");
    FS_ROWS();
    if (!ctx->use_pairs) {
        int rc = fs_rbsor_halfsweep(ctx, dt, dx, omega, 1, pn, pc, vc, std::max(row_begin - 1, 0), std::min(row_end + 1, ctx->rows));
        if (rc) return rc;
        return fs_rbsor_halfsweep(ctx, dt, dx, omega, 0, pn, pn, vc, row_begin, row_end);
    }
    // lanes of 2 cells, 4-row tiles (fs_k34n.h k_rbsor_iter_n): 119 -> 115 us at bc5 res 4096 against the 3-row quad tiles it replaces, f64 (bc3 res
    // 4096) 318 -> 289; 2 / 6 rows: 129 / 115
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 4, 1, XCD_RBSOR, 3, false);
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0, 0.0, omega);
#define FS_RBN4(DM) FS_KLAUNCH((k_rbsor_iter_n<2, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (T *)pn->d, (const T *)pc->d, (const T *)vc->d)
        return launch(ctx, "rbsor_iteration", [=] { FS_DMC(dm_const(ctx, k), FS_RBN4); });
    })
}

// four lazily-bounded Jacobi sweeps in one pass (fs_jquad.h): pn[not wall] <- sweep^4(pc); both buffers hold raw sweep output
int fs_jacobi_quad_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = ctx->mask_set && ctx->jq_ok && ctx->use_march && ctx->dtype == 0 ? 1 : 0;
    return FS_OK;
}

int fs_jacobi_quad_lazy(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_ROWS();
    if (!(ctx->jq_ok && ctx->use_march && ctx->dtype == 0)) { set_error("this mask / precision does not admit the four-sweep Jacobi pass (fs_jacobi_quad_ok)"); return FS_ERR_UNSUPPORTED; }
    using T = float;
    // lanes of 2 cells (116 VGPRs = 4 waves per SIMD at 4 rows; quads: 182 = 2 waves, 44.9 against 34.3 us per pass at bc2 res 1600)
    constexpr int rt = 4;
#define FS_JQ(RT, PATH) FS_KLAUNCH((k_jacobi_quad<2, RT, PATH, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pn->d, (const T *)pc->d, (const T *)src->d)
    // ONE launch of one-wave workgroups whose list entry says which body the tile takes, masked tiles first (round 6; fs_launch.h ov_grid_lanes: the hinted
    // launches of the pressure families) - bc5 res 4096 122.5 -> 108.7 us per pass against rounds 4 - 5's two compact launches over the all-fluid and the other
    // workgroups (81.4 + 49.8; 137.5 dense), bc2 res 1600 30.2 -> 23.9 against the 4-wave workgroups with per-wave hints
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, true, 0, 4);      // (reach 4: the hints)
    return launch(ctx, "jacobi_quad_lazy", [=] {
        FS_JQ(4, 2);
    });
}

// the last two rounds of a lazily-bounded Jacobi run in one pass (fs_jquad.h k_jacobi_finish): from pc = raw iterate n-2,
//   pc_out <- iterate n (not-wall cells) + K7(iterate n-2) (wall cells with a recipe);  pn <- iterate n-1 as K7 leaves it
int fs_jacobi_finish(fs_ctx *ctx, fs_field *pc_out, fs_field *pn, const fs_field *pc, const fs_field *src, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pc_out, 1); FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pc_out != pc && pc_out != pn && pn != pc, "the finishing pass needs three distinct pressure fields");
    FS_ROWS();
    if (!(ctx->jq_ok && ctx->use_march && ctx->dtype == 0)) { set_error("this mask / precision does not admit the multi-sweep Jacobi passes (fs_jacobi_quad_ok)"); return FS_ERR_UNSUPPORTED; }
    using T = float;
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 4, 1, XCD_RBSOR, 2, true, 0, 2);      // (per-wave plain hints: two sweeps reach 2 rows)
    return launch(ctx, "jacobi_finish", [=] {
        FS_KLAUNCH((k_jacobi_finish<2, 4, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end,
                           (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn->d, (const T *)pc->d, (const T *)src->d);
    });
}

int fs_rbsor_pair_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = ctx->mask_set && ctx->rb_pair_ok && ctx->use_pairs ? 1 : 0;      // (f32 and, since round 4, f64)
    return FS_OK;
}

// two red-black iterations + both pressure boundary passes in one pass (fs_rbpair.h): (pc_out, pn_out) <- the state two iterations of
// fs/pressure_updater.py:86-96 leave in (p.current, p.next) when they start from (pc, pn)
int fs_rbsor_pair(fs_ctx *ctx, double dt, double dx, double omega, fs_field *pc_out, fs_field *pn_out, const fs_field *pc, const fs_field *pn,
                  const fs_field *vc, int full, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pc_out, 1); FS_FIELD(pn_out, 1); FS_FIELD(pc, 1); FS_FIELD(pn, 1); FS_FIELD(vc, 2);
    FS_REQUIRE(pc_out != pn_out && pc_out != pc && pc_out != pn && pn_out != pc && pn_out != pn && pc != pn, "the two-iteration pass needs four distinct pressure fields");
    FS_ROWS();
    if (!(ctx->rb_pair_ok && ctx->use_pairs)) {
        set_error("this mask does not admit the two-iteration red-black pass (fs_rbsor_pair_ok)");
        return FS_ERR_UNSUPPORTED;
    }
    const Grid gg = ctx->grid();
    const int par0 = (gg.ybase + row_begin) & 1;
    if (ctx->dtype == 1) {
        // f64 (round 4; BASELINE configs[4]'s truth leg): the same body on double2 lanes.  A lane's window costs twice the registers, so the
        // tiles are 2 rows high (230 VGPRs with both paths = 2 waves per SIMD; the plain part on 4-row tiles: 220) - against 2 x (K7 + single
        // iteration) at 137 VGPRs that is still one pass over p and v instead of two.
        using T = double;
        auto k = make_konst<T>(ctx, dt, dx, 1.0, 0.0, omega);
#define FS_RBPD_K(RT, PAR, PATH, FULL) FS_KLAUNCH((k_rbsor_pair<2, RT, PAR, 0, PATH, FULL, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn_out->d, (const T *)pc->d, (const T *)pn->d, (const T *)vc->d)
#define FS_RBPD(RT, PATH, FULL) do { if (par0) FS_RBPD_K(RT, 1, PATH, FULL); else FS_RBPD_K(RT, 0, PATH, FULL); } while (0)
        if (!full && (ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 23)))) {
            // (round 6: ONE launch of one-wave workgroups over both kinds of tile, as the f32 pass has it, loses here - 514-527 against 424 us at bc3 res 4096:
            //  the double2 bodies hold 220-256 VGPRs, and the masked one then sets the occupancy of the all-fluid tiles too)
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 4, 1, XCD_RBSOR, 2, true, 1, 4, 1);
            const OvGrid ogb = ov_grid_lanes(ctx, row_begin, row_end, 2, 1, XCD_RBSOR, 2, true, 2, 4, 1, 4);
            if (og.g.tiles && ogb.g.tiles) {
                int rc = launch(ctx, "rbsor_pair", [=] { FS_RBPD(4, 3, false); });
                if (rc) return rc;
                { const OvGrid og = ogb; return launch(ctx, "rbsor_pair_bnd", [=] { FS_RBPD(2, 2, false); }); }
            }
        }
        const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 2, 1, XCD_RBSOR, 2, !full, 0, 4);
        return launch(ctx, "rbsor_pair", [=] { if (full) FS_RBPD(2, 2, true); else FS_RBPD(2, 2, false); });
    }
    using T = float;
    auto k = make_konst<T>(ctx, dt, dx, 1.0, 0.0, omega);
    const int dm = dm_const(ctx, k);
    // lanes of 2 cells (8-byte loads: 126 - 156 VGPRs where quads need 223 - 248), RT = 4 rows per tile (6: 210-254 us, window registers).  The carrying
    // pass after an upload (full) is rare: one configuration.
    // (grids below 1 M cells: 2-row tiles - fewer waves than SIMDs there, the pass takes as long as ONE wave's chain of loads and stages:
    //  res 200 12.1 -> 9.2 us per launch, BASELINE configs[0] 53.3 -> 62.8 k steps/s; res 1600: 4 rows, 5602 against 5435 steps/s)
    const int rt = full || !small_tiles(ctx) ? 4 : 2;
#define FS_RBP_K(RT, PAR, DM, PATH, FULL) FS_KLAUNCH((k_rbsor_pair<2, RT, PAR, DM, PATH, FULL, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn_out->d, (const T *)pc->d, (const T *)pn->d, (const T *)vc->d)
#define FS_RBP_PAR(RT, DM, PATH, FULL) do { if (par0) FS_RBP_K(RT, 1, DM, PATH, FULL); else FS_RBP_K(RT, 0, DM, PATH, FULL); } while (0)
#define FS_RBP_DM(RT, PATH) do { if (dm & DM_F64) FS_RBP_PAR(RT, 4, PATH, false); else FS_RBP_PAR(RT, 0, PATH, false); } while (0)
    // Compact launch in two parts where the lists exist (single GPU, whole grid): the workgroups that see nothing but fluid within reach run
    // the plain path as its own kernel (PATH 3: no mask loads, 126 VGPRs = 4 waves per SIMD), the others the kernel with both paths.
    // Round 6: ONE launch over both kinds of tile (fs_rbpair.h k_rbsor_pair_all: all-fluid 16-row tiles as two stacked waves, the others as two 4-row
    // tiles with masks, boundary entries first) from 1 M cells - bc5 res 4096: 161 -> 139.5 us; bc2 res 1600 45.8 -> 36.4, res 1200 26.2 -> 20.2, res 1024
    // 23.0 -> 19.0, res 800 20.0 -> 16.7; res 512 14.5 -> 14.9 and res 400 12.6 -> 14.0 (the 2-row tiles of small grids stay there).  Round 5's two
    // launches over the two kinds of tile (k_rbsor_pair_stack, then the masked kernel: from 8 M cells) went with it.
    if (!full && (ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 20)))) {
        const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 8, 1, XCD_RBSOR, 2, true, 3, 4, 1, 16, ctx->halo != 0);
        if (og.g.tiles) {
#define FS_RBA_K(PAR, DM) FS_KLAUNCH((k_rbsor_pair_all<2, PAR, DM, T>), og.grid, dim3(128), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn_out->d, (const T *)pc->d, (const T *)pn->d, (const T *)vc->d)
#define FS_RBA_PAR(DM) do { if (par0) FS_RBA_K(1, DM); else FS_RBA_K(0, DM); } while (0)
            return launch(ctx, "rbsor_pair", [=] { if (dm & DM_F64) FS_RBA_PAR(4); else FS_RBA_PAR(0); });
        }
    }
    // (one launch: the list's entries carry a per-wave "plain" hint - a wave that sees nothing but fluid within 4 rows skips its mask loads)
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, !full, 0, 4);
    return launch(ctx, "rbsor_pair", [=] {
        if (full) FS_RBP_PAR(4, 0, 2, true);
        else if (rt == 2) FS_RBP_DM(2, 2);
        else FS_RBP_DM(4, 2);
    });
}

int fs_rbsor_halfsweep_src(fs_ctx *ctx, double omega, int parity, fs_field *pn, const fs_field *pc, const fs_field *src,
                           int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(parity == 0 || parity == 1, "parity must be 0 or 1");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, 1.0, 1.0, 0.0, omega);
        return launch(ctx, parity ? "rbsor_odd_src" : "rbsor_even_src", [=] {
            FS_KLAUNCH((k_rbsor<true, T>), rb_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k,
                               row_begin, parity, (T *)pn->d, (const T *)pc->d, (const T *)src->d);
        });
    })
}

int fs_poisson_source(fs_ctx *ctx, double dt, double dx, fs_field *src, const fs_field *vc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(src, 2); FS_FIELD(vc, 2);
    FS_REQUIRE(src != vc, "src must not alias vc");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        if (ctx->use_pairs) {
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 4, 1, XCD_JACOBI, 3);      // (deep-wall workgroups skipped: nobody reads the source there)
#define FS_PSN(DM) FS_KLAUNCH((k_poisson_source_n<2, 4, DM, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)src->d, (const T *)vc->d)
            return launch(ctx, "poisson_source", [=] { FS_DMC(dm_const(ctx, k), FS_PSN); });
        }
        FS_LAUNCH_CELLS("poisson_source", (k_poisson_source<T>), ctx->grid(), k, row_begin, (T *)src->d, (const T *)vc->d)
    })
}

int fs_poisson_residual(fs_ctx *ctx, double dt, double dx, const fs_field *p, const fs_field *vc, double *sum_sq, double *count)
{
    FS_REQUIRE(ctx && sum_sq && count, "null argument");
    FS_FIELD(p, 1); FS_FIELD(vc, 2);
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "residual during graph capture / tape recording");
    if (!ctx->mask_set) { set_error("mask not uploaded"); return FS_ERR_STATE; }
    const int row_begin = ctx->halo, row_end = ctx->halo + ctx->nyl;
    const dim3 grid((ctx->X + 255) / 256, (row_end - row_begin + RES_ROWS - 1) / RES_ROWS);
    const size_t nblocks = (size_t)grid.x * grid.y;
    if (nblocks > ctx->partial_cap) {
        if (ctx->d_partial) { FS_HIP(hipStreamSynchronize(ctx->stream)); FS_HIP(hipFree(ctx->d_partial)); ctx->d_partial = nullptr; ctx->partial_cap = 0; }
        FS_HIP(hipMalloc(&ctx->d_partial, nblocks * 2 * sizeof(double)));
        ctx->partial_cap = nblocks;
    }
    int rc;
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        rc = launch(ctx, "poisson_residual", [=] {
            FS_KLAUNCH((k_residual<T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, row_end,
                               (const T *)p->d, (const T *)vc->d, ctx->d_partial);
            FS_KLAUNCH((k_residual_final<double>), dim3(1), dim3(1024), 0, ctx->stream, (const double *)ctx->d_partial, (int)nblocks, ctx->d_acc);
        });
    })
    if (rc) return rc;
    double h[2];
    FS_HIP(hipMemcpyAsync(h, ctx->d_acc, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    *sum_sq = h[0];
    *count = h[1];
    return FS_OK;
}

}  // extern "C"
