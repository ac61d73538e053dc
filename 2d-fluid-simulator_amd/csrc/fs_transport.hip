// fs_transport.hip - C-ABI entry points of the transport kernels: K2' (upwind / KK update), K0, K2, K3, K4 and the fused K3+K4 pass of the CIP
// solvers, K10 - K13 of the dye, K5 / K6 / the fused vorticity confinement.
#include "fs_launch.h"
#include "fs_k234.h"

using namespace fs;

// K3 + K4 in one pass (fs_k34n.h), velocity (C = 2, v = nullptr) and dye (C = 3): lane width / tile rows by grid size
// (2 cells per lane x 4 rows; 4 cells per lane: 2 rows), compact two-part launch on large single-GPU grids
template <int C, bool CLAMP>
static int launch_k34(fs_ctx *ctx, const char *name, const char *name_bnd, double dt, double dx, fs_field *f_out, fs_field *gx_out, fs_field *gy_out,
                      const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v, int full, int jb, int je)
{
    using T = float;
    auto k = make_konst<T>(ctx, dt, dx, 1.0);
    const int dm = dm_dx(ctx, k);
    // geometry by grid size (K3+K4 of the velocity, us):   2 cells x 4 rows   4 x 2   2 x 2
    //   >= 8 M cells (two-part launch): bc5 res 4096 / bc5 res 2048 / bc2 res 3000                   307 / 103 / 258    325 / 106 / 267   - / 113 / 282
    //   2 - 8 M cells: bc2 res 1600 / bc5 res 1024 (the boundary kernel of 2 x 4 holds 4 waves per SIMD)  90.7 / 26.5   81.6 / 27.8   90.3 / 26.4
    //   smaller: bc2 res 800 / res 400 (workgroups of half the size)                                  28.2 / 14.5        26.6 / 13.7       24.8 / 12.9
    const size_t cells = (size_t)ctx->X * ctx->rows;      // (this context's slab)
    // (round 5, packed bodies: the dye's three channels on pairs everywhere - bc2 res 1600: 141 against 184 us on quads; the velocity's two stay on
    //  quads from 2 M to 8 M cells: 79 against 92 (2 x 2) / 101 (2 x 4) us)
    const int N = ctx->X % 4 != 0 || C == 3 ? 2 : (cells >= ((size_t)1 << 23) || cells < ((size_t)1 << 21) ? 2 : 4);
    // (below 1 M cells: 1-row tiles for the dye's three channels - a launch is one wave's chain there, fs_ctx::small_tiles; res 400: 17.6 against
    //  17.1 k steps/s with the dye; the velocity's pass stays on 2 rows: 29.0 against 28.1 k)
    const int RT = N == 4 ? 2 : (cells >= ((size_t)1 << 23) ? 4 : (small_tiles(ctx) && !full && C == 3 ? 1 : 2)), geo = N == 2 ? 3 : 4;
#define FS_K34(NN, R, DM, PL) FS_KLAUNCH((k_cip_grad_advect_n<C, NN, R, DM, PL, CLAMP, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, \
        (T *)f_out->d, (T *)gx_out->d, (T *)gy_out->d, (const T *)fn->d, (const T *)fc->d, (const T *)gxc->d, (const T *)gyc->d, v ? (const T *)v->d : (const T *)nullptr, \
        f_out->hot, (const uint8_t *)ctx->d_bcmap, full)
#define FS_K34_24(DM) FS_K34(2, 4, DM, false)
#define FS_K34_24P(DM) FS_K34(2, 4, DM, true)
#define FS_K34_22(DM) FS_K34(2, 2, DM, false)
#define FS_K34_22P(DM) FS_K34(2, 2, DM, true)
#define FS_K34_42(DM) FS_K34(4, 2, DM, false)
#define FS_K34_42P(DM) FS_K34(4, 2, DM, true)
#define FS_K34_21(DM) FS_K34(2, 1, DM, false)
#define FS_K34_ANY(SUF) do { if (N == 4) FS_DMX(dm, FS_K34_42##SUF); else if (RT == 4) FS_DMX(dm, FS_K34_24##SUF); else FS_DMX(dm, FS_K34_22##SUF); } while (0)
    // Compact launch in two parts on large single-GPU grids (as fs_rbsor_pair): the workgroups that see nothing but fluid within
    // reach run without mask loads, selects and conditional stores (PLAIN), the others the general tile
    if (!full && RT != 1 && (ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 23)))) {
        const OvGrid og = ov_grid_lanes(ctx, jb, je, RT, C, XCD_ADVECT, geo, true, 1, 2, 1);
        const OvGrid ogb = ov_grid_lanes(ctx, jb, je, RT, C, XCD_ADVECT, geo, true, 2, 2, 1);
        if (og.g.tiles && ogb.g.tiles) {
            int rc = launch(ctx, name, [=] { FS_K34_ANY(P); });
            if (rc) return rc;
            { const OvGrid og = ogb; return launch(ctx, name_bnd, [=] { FS_K34_ANY(); }); }
        }
    }
    // (the carrying pass visits every tile.  The per-wave plain hint of fs_rbsor_pair was tried here too: 78.8-79.2 against 77.4-78.8 us at bc2 res 1600 -
    //  the kernel then holds four tile bodies instead of two)
    const OvGrid og = ov_grid_lanes(ctx, jb, je, RT, C, XCD_ADVECT, geo, !full);
    return launch(ctx, name, [=] { if (N == 2 && RT == 1) FS_DMX(dm, FS_K34_21); else FS_K34_ANY(); });
}

extern "C" {

int fs_mac_update(fs_ctx *ctx, int scheme, double dt, double dx, double re, fs_field *vn, const fs_field *vc,
                  const fs_field *pc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(scheme == FS_UPWIND || scheme == FS_KK, "unknown advection scheme");
    FS_FIELD(vn, 2); FS_FIELD(vc, 2); FS_FIELD(pc, 1);
    FS_REQUIRE(vn != vc, "vn must not alias vc");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, re);
        if (ctx->use_pairs) {
            // lanes of 2 cells (fs_k34n.h k_mac_update_n), tiles of 4 rows on large f32 grids (KK at bc3 res 4096: 178 -> 162 us against the one-row quad
            // form it replaces; f64: 424 -> 306 with 2-row tiles), 2 rows on small grids (more workgroups) and for f64 (registers)
            // (round 6, packed body: 6-row tiles 144.7-151.1 against 154.2 us, 8-row tiles 157.8-158.9 (111 VGPRs = 4 waves): left at 4)
            const int rt = sizeof(T) == 4 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 20) ? 4 : 2;
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_NONADV, 3, true, 0, 2);      // (reach: per-wave plain hints in the list)
            return launch(ctx, scheme == FS_UPWIND ? "mac_update_upwind" : "mac_update_kk", [=] {
#define FS_K2MN(SS, RR, PP) FS_KLAUNCH((k_mac_update_n<SS, 2, RR, PP, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
            (T *)vn->d, (const T *)vc->d, (const T *)pc->d, vn->hot)
#define FS_K2MN_UP4(DM) FS_K2MN(0, 4, DM)
#define FS_K2MN_KK4(DM) FS_K2MN(1, 4, DM)
#define FS_K2MN_UP2(DM) FS_K2MN(0, 2, DM)
#define FS_K2MN_KK2(DM) FS_K2MN(1, 2, DM)
                if (rt == 2) { if (scheme == FS_UPWIND) FS_DMA(dm_all(ctx, k), FS_K2MN_UP2); else FS_DMA(dm_all(ctx, k), FS_K2MN_KK2); }
                else { if (scheme == FS_UPWIND) FS_DMA(dm_all(ctx, k), FS_K2MN_UP4); else FS_DMA(dm_all(ctx, k), FS_K2MN_KK4); }
            });
        }
        if (scheme == FS_UPWIND) { FS_LAUNCH_CELLS("mac_update_upwind", (k_mac_update<0, T>), ctx->grid(), k, row_begin, (T *)vn->d, (const T *)vc->d, (const T *)pc->d, vn->hot) }
        else { FS_LAUNCH_CELLS("mac_update_kk", (k_mac_update<1, T>), ctx->grid(), k, row_begin, (T *)vn->d, (const T *)vc->d, (const T *)pc->d, vn->hot) }
    })
}

int fs_mac_dye(fs_ctx *ctx, int scheme, double dt, double dx, fs_field *dn, const fs_field *dc, const fs_field *vc,
               int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(scheme == FS_UPWIND || scheme == FS_KK, "unknown advection scheme");
    FS_FIELD(dn, 3); FS_FIELD(dc, 3); FS_FIELD(vc, 2);
    FS_REQUIRE(dn != dc, "dn must not alias dc");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        if (scheme == FS_UPWIND) { FS_LAUNCH_CELLS("mac_dye_upwind", (k_mac_dye<0, T>), ctx->grid(), k, row_begin, (T *)dn->d, (const T *)dc->d, (const T *)vc->d) }
        else { FS_LAUNCH_CELLS("mac_dye_kk", (k_mac_dye<1, T>), ctx->grid(), k, row_begin, (T *)dn->d, (const T *)dc->d, (const T *)vc->d) }
    })
}

int fs_cip_set_grad(fs_ctx *ctx, double dx, fs_field *fx, fs_field *fy, const fs_field *f, int row_begin, int row_end)
{
    FS_REQUIRE(ctx && f, "null argument");
    const int C = f->C;
    FS_REQUIRE(C == 2 || C == 3, "set_grad needs a 2- or 3-channel field");
    FS_FIELD(fx, C); FS_FIELD(fy, C); FS_FIELD(f, C);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, dx, 1.0);
        if (C == 2) { FS_LAUNCH_CELLS("cip_set_grad", (k_cip_set_grad<2, T>), ctx->grid(), k, row_begin, (T *)fx->d, (T *)fy->d, (const T *)f->d) }
        else { FS_LAUNCH_CELLS("cip_set_grad_c3", (k_cip_set_grad<3, T>), ctx->grid(), k, row_begin, (T *)fx->d, (T *)fy->d, (const T *)f->d) }
    })
}

int fs_cip_nonadv(fs_ctx *ctx, double dt, double dx, double re, fs_field *fn, const fs_field *fc, const fs_field *pc,
                  int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(fn, 2); FS_FIELD(fc, 2); FS_FIELD(pc, 1);
    FS_REQUIRE(fn != fc, "fn must not alias fc");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, re);
        if (ctx->use_pairs) {
            // lanes of 2 cells, tiles of 4 rows (fs_k34n.h k_cip_nonadv_n), compact launch: 116 -> 102 us at bc5 res 4096 against the one-row quad form
            // it replaces (2 rows: 112, 8 rows: 106-110)
            // (small grids - fewer waves than SIMDs, a launch takes as long as one wave's chain: 2-row tiles, fs_ctx::small_tiles)
            const bool small = small_tiles(ctx);
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_NONADV, 3, true, 0, 1);      // (reach 1: per-wave plain hints)
            const int clear3 = whole_grid(ctx, row_begin, row_end);      // (fs_device.h "hot" word [3])
            return launch(ctx, "cip_nonadv", [=] {
#define FS_K2N4(DM) FS_KLAUNCH((k_cip_nonadv_n<2, 4, DM, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot, clear3)
#define FS_K2N2(DM) FS_KLAUNCH((k_cip_nonadv_n<2, 2, DM, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot, clear3)
                if (small) FS_DMA(dm_all(ctx, k), FS_K2N2); else FS_DMA(dm_all(ctx, k), FS_K2N4);
            });
        }
        if (k.p2) { FS_LAUNCH_CELLS("cip_nonadv", (k_cip_nonadv<true, T>), ctx->grid(), k, row_begin, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot) }
        FS_LAUNCH_CELLS("cip_nonadv", (k_cip_nonadv<false, T>), ctx->grid(), k, row_begin, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot)
    })
}

int fs_cip_nonadv_dye(fs_ctx *ctx, double dt, double dx, double re, fs_field *dn, const fs_field *dc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(dn, 3); FS_FIELD(dc, 3);
    FS_REQUIRE(dn != dc, "dn must not alias dc");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, re);
        if (ctx->use_pairs) {
            // lanes of 2 cells, 4-row tiles (fs_k34n.h k_cip_nonadv_dye_n), compact launch: 141 -> 122-130 us at bc5 res 4096 against the one-row quad form
            const bool small = small_tiles(ctx);       // (2-row tiles, see fs_cip_nonadv)
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_NONADV, 3, true, 0, 1);
            return launch(ctx, "cip_nonadv_dye", [=] {
#define FS_K12N(DM) FS_KLAUNCH((k_cip_nonadv_dye_n<2, 4, DM, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)dn->d, (const T *)dc->d)
#define FS_K12N2(DM) FS_KLAUNCH((k_cip_nonadv_dye_n<2, 2, DM, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)dn->d, (const T *)dc->d)
                if (small) FS_DMA(dm_all(ctx, k), FS_K12N2); else FS_DMA(dm_all(ctx, k), FS_K12N);
            });
        }
        FS_LAUNCH_CELLS("cip_nonadv_dye", (k_cip_nonadv_dye<T>), ctx->grid(), k, row_begin, (T *)dn->d, (const T *)dc->d)
    })
}

#define FS_K3Q(CC, NC, PP) FS_KLAUNCH((k_cip_nonadv_grad_quad<CC, NC, PP, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
            (T *)fxn->d, (T *)fyn->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)fc->d, (const T *)fn->d)
#define FS_K3(CC, PP, NAME) { FS_LAUNCH_CELLS(NAME, (k_cip_nonadv_grad<CC, PP, T>), ctx->grid(), k, row_begin, (T *)fxn->d, (T *)fyn->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)fc->d, (const T *)fn->d) }
int fs_cip_nonadv_grad(fs_ctx *ctx, double dx, fs_field *fxn, fs_field *fyn, const fs_field *fxc, const fs_field *fyc,
                       const fs_field *fc, const fs_field *fn, int row_begin, int row_end)
{
    FS_REQUIRE(ctx && fc, "null argument");
    const int C = fc->C;
    FS_REQUIRE(C == 2 || C == 3, "nonadv_grad needs 2- or 3-channel fields");
    FS_FIELD(fxn, C); FS_FIELD(fyn, C); FS_FIELD(fxc, C); FS_FIELD(fyc, C); FS_FIELD(fc, C); FS_FIELD(fn, C);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, dx, 1.0);
        if (ctx->use_march) {
            const OvGrid og = ov_grid(ctx, row_begin, row_end, 1, C == 2 ? 1 : 3, XCD_GRAD);
            return launch(ctx, C == 2 ? "cip_nonadv_grad" : "cip_nonadv_grad_c3", [=] {
#define FS_K3Q_V(DM) FS_K3Q(2, 2, DM)
#define FS_K3Q_D(DM) FS_K3Q(3, 1, DM)
                if (C == 2) FS_DMX(dm_dx(ctx, k), FS_K3Q_V); else FS_DMX(dm_dx(ctx, k), FS_K3Q_D);
            });
        }
        if (C == 2 && k.p2) FS_K3(2, true, "cip_nonadv_grad")
        else if (C == 2) FS_K3(2, false, "cip_nonadv_grad")
        else if (k.p2) FS_K3(3, true, "cip_nonadv_grad_c3")
        else FS_K3(3, false, "cip_nonadv_grad_c3")
    })
}

#define FS_K4Q(CC, NC, SELF, PP) FS_KLAUNCH((k_cip_advect_quad<CC, NC, SELF, PP, false, T>), qgrid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
#define FS_K4D(PP) FS_KLAUNCH((k_cip_advect_dye<PP, false, T>), qgrid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
#define FS_K4N(CC, PP) FS_KLAUNCH((k_cip_advect<CC, PP, T>), cells_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, \
        (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
int fs_cip_advect(fs_ctx *ctx, double dt, double dx, fs_field *fn, fs_field *fxn, fs_field *fyn, const fs_field *fc,
                  const fs_field *fxc, const fs_field *fyc, const fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx && fc, "null argument");
    const int C = fc->C;
    FS_REQUIRE(C == 2 || C == 3, "cip_advect needs 2- or 3-channel fields");
    FS_FIELD(fn, C); FS_FIELD(fxn, C); FS_FIELD(fyn, C); FS_FIELD(fc, C); FS_FIELD(fxc, C); FS_FIELD(fyc, C); FS_FIELD(v, 2);
    FS_REQUIRE(fn != fc && fxn != fxc && fyn != fyc, "outputs must not alias inputs");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        const bool self = (v == fc);
        const OvGrid og = ov_grid(ctx, row_begin, row_end, 1, (C == 2 && !self) ? 2 : 1, XCD_ADVECT);   // C == 3: one pass over the channels
        const dim3 qgrid = og.grid;
        return launch(ctx, C == 2 ? "cip_advect" : "cip_advect_c3", [=] {
            if (ctx->use_march) {
#define FS_K4Q_SELF(DM) FS_K4Q(2, 2, true, DM)
#define FS_K4Q_OTHER(DM) FS_K4Q(2, 1, false, DM)
                if (C == 2 && self) FS_DMX(dm_dx(ctx, k), FS_K4Q_SELF);
                else if (C == 2) FS_DMX(dm_dx(ctx, k), FS_K4Q_OTHER);
                else FS_DMX(dm_dx(ctx, k), FS_K4D);
            } else {
                if (C == 2) { if (k.p2) FS_K4N(2, true); else FS_K4N(2, false); }
                else { if (k.p2) FS_K4N(3, true); else FS_K4N(3, false); }
            }
        });
    })
}

int fs_cip_advect_dye_clamped(fs_ctx *ctx, double dt, double dx, fs_field *fn, fs_field *fxn, fs_field *fyn, const fs_field *fc,
                              const fs_field *fxc, const fs_field *fyc, const fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(fn, 3); FS_FIELD(fxn, 3); FS_FIELD(fyn, 3); FS_FIELD(fc, 3); FS_FIELD(fxc, 3); FS_FIELD(fyc, 3); FS_FIELD(v, 2);
    FS_REQUIRE(fn != fc && fxn != fxc && fyn != fyc, "outputs must not alias inputs");
    FS_REQUIRE(ctx->use_march, "needs X % 4 == 0 (use fs_cip_advect + fs_clamp_field)");
    FS_ROWS();
    const OvGrid og = ov_grid(ctx, row_begin, row_end, 1, 1, XCD_ADVECT);
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        return launch(ctx, "cip_advect_c3_clamped", [=] {
#define FS_K4DC(DM) FS_KLAUNCH((k_cip_advect_dye<DM, true, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                                         (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
            FS_DMX(dm_dx(ctx, k), FS_K4DC);
        });
    })
}

int fs_clamp_inflow(fs_ctx *ctx, double low, double high, fs_field *dye, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(dye, 3);
    FS_ROWS();
    if (!ctx->d_bc_dye) { set_error("bc_dye not uploaded"); return FS_ERR_STATE; }
    if (ctx->ops_dye.lanes() == 0) return FS_OK;
    FS_DISPATCH(ctx, {
        return launch(ctx, "clamp_inflow", [=] {
            FS_KLAUNCH(k_clamp_inflow<T>, dim3((ctx->ops_dye.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_dye.view(), row_begin, row_end, (T)low, (T)high, (T *)dye->d);
        });
    })
}

int fs_cip_grad_advect(fs_ctx *ctx, double dt, double dx, fs_field *v_out, fs_field *gx_out, fs_field *gy_out,
                       const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc, int full, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v_out, 2); FS_FIELD(gx_out, 2); FS_FIELD(gy_out, 2); FS_FIELD(fn, 2); FS_FIELD(fc, 2); FS_FIELD(gxc, 2); FS_FIELD(gyc, 2);
    FS_REQUIRE(v_out != fn && v_out != fc && gx_out != gxc && gy_out != gyc && fn != fc, "outputs must not alias inputs");
    FS_REQUIRE(ctx->use_pairs, "the fused gradient+advection pass needs an even X (use the two-kernel form)");
    FS_ROWS();
    if (ctx->dtype != 0) { set_error("the fused gradient+advection pass exists for f32 (f64: the two-kernel form)"); return FS_ERR_UNSUPPORTED; }
    return launch_k34<2, false>(ctx, "cip_grad_advect_rt", "cip_grad_advect_rt_bnd", dt, dx, v_out, gx_out, gy_out, fn, fc, gxc, gyc, nullptr, full, row_begin, row_end);
}

static bool cip_step_multi_part(const fs_ctx *ctx)
{
    // Wherever the launch lists exist (late round 5, K2 / K12 in registers on every tile in ONE launch - 2-row tiles below FS_SMALL_CELLS: bc2 res 400 28.6 -> 29.4 k
    // steps/s, with dye 18.0 -> 19.2 k; bc5 res 512 27.2 -> 29.0 k, with dye 16.6 -> 18.1 k; bc2 res 800 16.1 -> 17.1 k; 4-row tiles: bc5 res 1024 13.9 -> 14.9 k, res
    // 1200 10.35 -> 11.0 k, res 1600 6 150 -> 6 700).  As two launches the form paid from 2.6 M cells; with K2 as a third launch over the boundary tiles' rows and
    // the general K3 + K4 kernel there (the round's first form) from 8 M.
    const bool big = ctx->rbpair_split == 2 || ctx->rbpair_split == 1;
    return ctx->mask_set && ctx->fuse_k2 != 0 && big && ctx->dtype == 0 && ctx->use_pairs && !ctx->h_act2.empty() && (ctx->tile_list_mask & XCD_ADVECT);
}
int fs_cip_step_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = cip_step_multi_part(ctx) ? 1 : 0;
    return FS_OK;
}

// diagnostic (bench.py: the algorithmic bytes of each part): how many tiles of tile_rows x tile_cells cells the two classes of a whole-grid fs_cip_step
// launch hold - the all-fluid tiles, the others (`band`: 0 since the form with K2 as a kernel over the boundary tiles' rows is gone; 0 0 0: the two calls)
int fs_cip_step_tiles(fs_ctx *ctx, int *plain, int *boundary, int *band, int *tile_rows, int *tile_cells)
{
    FS_REQUIRE(ctx && plain && boundary && band && tile_rows && tile_cells, "null argument");
    // (the one-launch form runs 2-row tiles below FS_SMALL_CELLS; the per-kind lists counted here are those of the two-launch form: 4-row tiles everywhere)
    *plain = *boundary = *band = 0; *tile_rows = 4; *tile_cells = 120;
    if (!cip_step_multi_part(ctx) || ctx->capturing || ctx->tape_rec) return FS_OK;
    if (ctx->fuse_k2 >= 2 && small_tiles(ctx)) *tile_rows = 2;
    const OvGrid ogp = ov_grid_lanes(ctx, 0, ctx->rows, *tile_rows, 1, XCD_ADVECT, 2, true, 1, 2, 1);
    const OvGrid ogb = ov_grid_lanes(ctx, 0, ctx->rows, *tile_rows, 1, XCD_ADVECT, 2, true, 2, 2, 1);
    for (const auto &kv : ctx->tile_lists) {
        if (!kv.second.d) continue;
        if (kv.second.d == ogp.g.tiles) *plain = kv.second.count;
        if (kv.second.d == ogb.g.tiles) *boundary = kv.second.count;
    }
    return FS_OK;
}

// K2 + K3 + K4 of the velocity (fs/solver.py:213-227) as ONE call: fs_cip_nonadv(fn <- fc, pc) followed by fs_cip_grad_advect(v_out, gx_out,
// gy_out <- fn, fc, gxc, gyc) - with the one difference that the fluid cells of fn that nothing reads before the next kernel rewrites them
// are NOT stored where the form below applies (f32, even X, wherever the launch lists exist - cip_step_multi_part above: every grid size;
// fs_k234.h): every tile evaluates K2 in registers on the way to K3 + K4 -
// ONE launch over the list of all tiles, whose entries say which body a tile takes (k_cip_step_all: all fluid within reach / masks, K2 stored
// on inflow / outflow cells).  FS_FUSE_K2=1: the two bodies as two launches over the two classes (k_cip_step_plain, k_cip_step_bnd: the form
// of bench.py's per-part roofline).  fs_cip_step_ok: the static conditions (the kernel names of a profile say what ran).
int fs_cip_step(fs_ctx *ctx, double dt, double dx, double re, fs_field *v_out, fs_field *gx_out, fs_field *gy_out, fs_field *fn,
                const fs_field *fc, const fs_field *pc, const fs_field *gxc, const fs_field *gyc, int full, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v_out, 2); FS_FIELD(gx_out, 2); FS_FIELD(gy_out, 2); FS_FIELD(fn, 2); FS_FIELD(fc, 2); FS_FIELD(pc, 1); FS_FIELD(gxc, 2); FS_FIELD(gyc, 2);
    FS_REQUIRE(v_out != fn && v_out != fc && gx_out != gxc && gy_out != gyc && fn != fc, "outputs must not alias inputs");
    FS_REQUIRE(ctx->use_pairs, "the fused gradient+advection pass needs an even X (use the two-kernel form)");
    FS_ROWS();
    if (ctx->dtype != 0) { set_error("the fused gradient+advection pass exists for f32 (f64: the two-kernel form)"); return FS_ERR_UNSUPPORTED; }
    using T = float;
    constexpr int RT = 4;
    // (a slab - halo != 0: any row range; K2 is then evaluated for the rows within 2 of the range from rows within 3 of it, which the caller keeps
    //  valid - fs/runtime.py cip_step - where the two calls would read what an earlier K2 launch left in fn.  Single GPU: the whole grid.)
    const bool slab = ctx->halo != 0;
    // (K2 in registers reads rows row_begin - 3 .. row_end + 2 of fc and pc through clampy(), which clamps to the DOMAIN's edge rows, not to the
    //  slab buffer: a range closer than 3 rows to the buffer's end that is not the domain's edge takes the two-call form below, whose K2 launch has its own range)
    const bool rows_ok = !slab || ((row_begin >= 3 || ctx->y0 - ctx->halo + row_begin <= 0) && (row_end + 3 <= ctx->rows || ctx->y0 - ctx->halo + row_end >= ctx->Y));
    if (cip_step_multi_part(ctx) && !full && rows_ok && (slab || (row_begin == 0 && row_end == ctx->rows))) {
        auto k = make_konst<T>(ctx, dt, dx, re);
        const int dm = dm_all(ctx, k);
#define FS_K234(KERNEL, DM) FS_KLAUNCH((KERNEL<RT, DM>), og.grid, dim3(128), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)v_out->d, (T *)gx_out->d, (T *)gy_out->d, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, (const T *)gxc->d, (const T *)gyc->d, v_out->hot, fn->hot)
#define FS_K234A(DM) FS_K234(k_cip_step_all, DM)
#define FS_K234P(DM) FS_K234(k_cip_step_plain, DM)
#define FS_K234B(DM) FS_K234(k_cip_step_bnd, DM)
        if (ctx->fuse_k2 >= 2) {
            // one launch over both kinds of tile: the class 0 list, whose entries carry the per-tile "all fluid within reach" hint
            // (grids below FS_SMALL_CELLS: 2-row tiles - a launch there lasts as long as one wave's chain, fs_launch.h small_tiles)
            if (small_tiles(ctx)) {
                const OvGrid oga = ov_grid_lanes(ctx, row_begin, row_end, 2, 1, XCD_ADVECT, 2, true, 0, 2, 1, 0, slab);
#define FS_K234A2(DM) FS_KLAUNCH((k_cip_step_all<2, DM>), og.grid, dim3(128), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)v_out->d, (T *)gx_out->d, (T *)gy_out->d, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, (const T *)gxc->d, (const T *)gyc->d, v_out->hot, fn->hot)
                if (oga.g.tiles) return launch(ctx, "cip_step", [=] { const OvGrid og = oga; FS_DMA(dm, FS_K234A2); });
            }
            const OvGrid oga = ov_grid_lanes(ctx, row_begin, row_end, RT, 1, XCD_ADVECT, 2, true, 0, 2, 1, 0, slab);
            if (oga.g.tiles) return launch(ctx, "cip_step", [=] { const OvGrid og = oga; FS_DMA(dm, FS_K234A); });
        }
        // two launches: the all-fluid tiles, the others (one entry per tile, two waves each)
        const OvGrid ogp = ov_grid_lanes(ctx, row_begin, row_end, RT, 1, XCD_ADVECT, 2, true, 1, 2, 1, 0, slab);
        const OvGrid ogb = ov_grid_lanes(ctx, row_begin, row_end, RT, 1, XCD_ADVECT, 2, true, 2, 2, 1, 0, slab);
        if (ogp.g.tiles && ogb.g.tiles) {
            int rc = launch(ctx, "cip_step", [=] { const OvGrid og = ogp; FS_DMA(dm, FS_K234P); });
            if (rc) return rc;
            return launch(ctx, "cip_step_bnd", [=] { const OvGrid og = ogb; FS_DMA(dm, FS_K234B); });
        }
    }
    // (a slab: K2 also on the 2 rows either side that K3 + K4 read - the call's contract there, see above)
    int rc = fs_cip_nonadv(ctx, dt, dx, re, fn, fc, pc, slab ? std::max(row_begin - 2, 0) : row_begin, slab ? std::min(row_end + 2, ctx->rows) : row_end);
    if (rc) return rc;
    return launch_k34<2, false>(ctx, "cip_grad_advect_rt", "cip_grad_advect_rt_bnd", dt, dx, v_out, gx_out, gy_out, fn, fc, gxc, gyc, nullptr, full, row_begin, row_end);
}

// the dye: d_out <- advect(fn with the gradients K3 derives from fc -> fn) by v
int fs_cip_grad_advect_dye(fs_ctx *ctx, double dt, double dx, fs_field *d_out, fs_field *gx_out, fs_field *gy_out,
                           const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v,
                           int clamp01, int full, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(d_out, 3); FS_FIELD(gx_out, 3); FS_FIELD(gy_out, 3); FS_FIELD(fn, 3); FS_FIELD(fc, 3); FS_FIELD(gxc, 3); FS_FIELD(gyc, 3); FS_FIELD(v, 2);
    FS_REQUIRE(d_out != fn && d_out != fc && gx_out != gxc && gy_out != gyc && fn != fc, "outputs must not alias inputs");
    FS_REQUIRE(ctx->use_pairs, "the fused gradient+advection pass needs an even X (use the two-kernel form)");
    FS_ROWS();
    if (row_begin >= row_end) return FS_OK;
    if (ctx->dtype != 0) { set_error("the fused dye pass exists for f32 (f64: the two-kernel form)"); return FS_ERR_UNSUPPORTED; }
    if (clamp01) return launch_k34<3, true>(ctx, "cip_grad_advect_dye", "cip_grad_advect_dye_bnd", dt, dx, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, full, row_begin, row_end);
    return launch_k34<3, false>(ctx, "cip_grad_advect_dye", "cip_grad_advect_dye_bnd", dt, dx, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, full, row_begin, row_end);
}

// K12 + K3 + K4 of the dye (fs/solver.py:385-401 _update_dye) as ONE call: fs_cip_nonadv_dye(fn <- fc) followed by fs_cip_grad_advect_dye - in the form of
// fs_cip_step where that applies (same conditions, fs_cip_step_ok): K12 in registers on every tile (fs_k234.h k_cip_dye: one launch over all tiles, or -
// FS_FUSE_K2=1 - one per class).  The fluid cells of fn that nothing reads before the next K12 rewrites them are then not stored.
int fs_cip_step_dye(fs_ctx *ctx, double dt, double dx, double re, fs_field *d_out, fs_field *gx_out, fs_field *gy_out, fs_field *fn,
                    const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v, int clamp01, int full, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(d_out, 3); FS_FIELD(gx_out, 3); FS_FIELD(gy_out, 3); FS_FIELD(fn, 3); FS_FIELD(fc, 3); FS_FIELD(gxc, 3); FS_FIELD(gyc, 3); FS_FIELD(v, 2);
    FS_REQUIRE(d_out != fn && d_out != fc && gx_out != gxc && gy_out != gyc && fn != fc, "outputs must not alias inputs");
    FS_REQUIRE(ctx->use_pairs, "the fused gradient+advection pass needs an even X (use the two-kernel form)");
    FS_ROWS();
    if (ctx->dtype != 0) { set_error("the fused dye pass exists for f32 (f64: the two-kernel form)"); return FS_ERR_UNSUPPORTED; }
    using T = float;
    constexpr int RT = 4;
    const bool slab = ctx->halo != 0;      // (as fs_cip_step, its row condition included)
    const bool rows_ok = !slab || ((row_begin >= 3 || ctx->y0 - ctx->halo + row_begin <= 0) && (row_end + 3 <= ctx->rows || ctx->y0 - ctx->halo + row_end >= ctx->Y));
    if (cip_step_multi_part(ctx) && !full && rows_ok && (slab || (row_begin == 0 && row_end == ctx->rows))) {
        auto k = make_konst<T>(ctx, dt, dx, re);
        const int dm = dm_all(ctx, k);
        // k_cip_dye<RT, DM, CLAMP, KIND> over a list: KIND 1 - the all-fluid tiles, 2 - the others, 0 - both (class 0 list with the per-tile hint)
#define FS_KD(DM, CL, KIND) FS_KLAUNCH((k_cip_dye<RT, DM, CL, KIND>), og.grid, dim3(64), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)d_out->d, (T *)gx_out->d, (T *)gy_out->d, (T *)fn->d, (const T *)fc->d, (const T *)gxc->d, (const T *)gyc->d, (const T *)v->d)
#define FS_KD_C0(DM) FS_KD(DM, true, 0)
#define FS_KD_N0(DM) FS_KD(DM, false, 0)
#define FS_KD_C1(DM) FS_KD(DM, true, 1)
#define FS_KD_N1(DM) FS_KD(DM, false, 1)
#define FS_KD_C2(DM) FS_KD(DM, true, 2)
#define FS_KD_N2(DM) FS_KD(DM, false, 2)
        if (ctx->fuse_k2 >= 2) {
            if (small_tiles(ctx)) {      // (2-row tiles, as fs_cip_step)
                const OvGrid oga = ov_grid_lanes(ctx, row_begin, row_end, 2, 3, XCD_ADVECT, 2, true, 0, 2, 1, 0, slab);
#define FS_KD2(DM, CL) FS_KLAUNCH((k_cip_dye<2, DM, CL, 0>), og.grid, dim3(64), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)d_out->d, (T *)gx_out->d, (T *)gy_out->d, (T *)fn->d, (const T *)fc->d, (const T *)gxc->d, (const T *)gyc->d, (const T *)v->d)
#define FS_KD2_C(DM) FS_KD2(DM, true)
#define FS_KD2_N(DM) FS_KD2(DM, false)
                if (oga.g.tiles) return launch(ctx, "cip_step_dye", [=] { const OvGrid og = oga; if (clamp01) FS_DMA(dm, FS_KD2_C); else FS_DMA(dm, FS_KD2_N); });
            }
            const OvGrid oga = ov_grid_lanes(ctx, row_begin, row_end, RT, 3, XCD_ADVECT, 2, true, 0, 2, 1, 0, slab);
            if (oga.g.tiles) return launch(ctx, "cip_step_dye", [=] { const OvGrid og = oga; if (clamp01) FS_DMA(dm, FS_KD_C0); else FS_DMA(dm, FS_KD_N0); });
        }
        const OvGrid ogp = ov_grid_lanes(ctx, row_begin, row_end, RT, 3, XCD_ADVECT, 2, true, 1, 2, 1, 0, slab);
        const OvGrid ogb = ov_grid_lanes(ctx, row_begin, row_end, RT, 3, XCD_ADVECT, 2, true, 2, 2, 1, 0, slab);
        if (ogp.g.tiles && ogb.g.tiles) {
            int rc = launch(ctx, "cip_step_dye", [=] { const OvGrid og = ogp; if (clamp01) FS_DMA(dm, FS_KD_C1); else FS_DMA(dm, FS_KD_N1); });
            if (rc) return rc;
            return launch(ctx, "cip_step_dye_bnd", [=] { const OvGrid og = ogb; if (clamp01) FS_DMA(dm, FS_KD_C2); else FS_DMA(dm, FS_KD_N2); });
        }
    }
    int rc = fs_cip_nonadv_dye(ctx, dt, dx, re, fn, fc, slab ? std::max(row_begin - 2, 0) : row_begin, slab ? std::min(row_end + 2, ctx->rows) : row_end);
    if (rc) return rc;
    return fs_cip_grad_advect_dye(ctx, dt, dx, d_out, gx_out, gy_out, fn, fc, gxc, gyc, v, clamp01, full, row_begin, row_end);
}

// ---- vorticity confinement -------------------------------------------------------------------------------
int fs_vort_calc(fs_ctx *ctx, double dx, fs_field *vort, fs_field *vort_abs, const fs_field *vc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(vort, 1); FS_FIELD(vort_abs, 1); FS_FIELD(vc, 2);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, dx, 1.0);
        FS_LAUNCH_CELLS("vort_calc", (k_vort_calc<T>), ctx->grid(), k, row_begin, (T *)vort->d, (T *)vort_abs->d, (const T *)vc->d)
    })
}

int fs_vort_add(fs_ctx *ctx, double dt, double dx, double weight, fs_field *vn, const fs_field *vc, const fs_field *vort,
                const fs_field *vort_abs, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(vn, 2); FS_FIELD(vc, 2); FS_FIELD(vort, 1); FS_FIELD(vort_abs, 1);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0, weight);
        FS_LAUNCH_CELLS("vort_add", (k_vort_add<T>), ctx->grid(), k, row_begin, (T *)vn->d, (const T *)vc->d, (const T *)vort->d, (const T *)vort_abs->d, vn->hot)
    })
}

int fs_vort_confine(fs_ctx *ctx, double dt, double dx, double weight, fs_field *vn, const fs_field *vc, fs_field *vort,
                    fs_field *vort_abs, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(vn, 2); FS_FIELD(vc, 2);
    FS_REQUIRE(vn != vc, "vn must not alias vc");
    FS_REQUIRE((vort == nullptr) == (vort_abs == nullptr), "pass both vort and vort_abs or neither");
    if (vort) { FS_FIELD(vort, 1); FS_FIELD(vort_abs, 1); }
    FS_ROWS();
    if (!ctx->use_pairs) {   // odd width: the unfused pair
        if (!vort) { set_error("fused vorticity confinement needs an even X or explicit vort fields"); return FS_ERR_UNSUPPORTED; }
        int rc = fs_vort_calc(ctx, dx, vort, vort_abs, vc, std::max(row_begin - 1, 0), std::min(row_end + 1, ctx->rows));
        if (rc) return rc;
        return fs_vort_add(ctx, dt, dx, weight, vn, vc, vort, vort_abs, row_begin, row_end);
    }
    // lanes of 2 cells (fs_k34n.h k_vort_n), 4-row tiles, compact launch: 100 -> 95 us at bc5 res 4096 against the quad form it replaces (6 / 8 rows:
    // 102 / 103; f64 at bc3 res 4096: 251 -> 224)
    const bool small = small_tiles(ctx) && !vort;      // (small grids: 2-row tiles, see fs_cip_nonadv)
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_VORT, 3, true, 0, 2);      // (reach 2: per-wave plain hints in the list)
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0, weight);
        const int dm = dm_dx(ctx, k);
        T *w = vort ? (T *)vort->d : nullptr; T *wa = vort_abs ? (T *)vort_abs->d : nullptr;
        const int clear3 = whole_grid(ctx, row_begin, row_end);      // (fs_device.h "hot" word [3])
#define FS_VORTN(DM, ST) FS_KLAUNCH((k_vort_n<2, 4, DM, ST, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)vn->d, (const T *)vc->d, w, wa, vn->hot, clear3)
#define FS_VORTN_S(DM) FS_VORTN(DM, true)
#define FS_VORTN_N(DM) FS_VORTN(DM, false)
#define FS_VORTN_2(DM) FS_KLAUNCH((k_vort_n<2, 2, DM, false, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)vn->d, (const T *)vc->d, w, wa, vn->hot, clear3)
        return launch(ctx, "vort_confine", [=] { if (vort) FS_DMX(dm, FS_VORTN_S); else if (small) FS_DMX(dm, FS_VORTN_2); else FS_DMX(dm, FS_VORTN_N); });
    })
}

}  // extern "C"
