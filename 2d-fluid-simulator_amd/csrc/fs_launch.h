// fs_launch.h - host-side launch plumbing shared by the translation units behind the C-ABI (fs_core.hip: contexts, fields, scene upload,
// boundary kernels, graphs / tapes / profiling; fs_transport.hip: K2 - K6, K10 - K13; fs_pressure.hip: K7 - K8, the Poisson residual):
// the launch wrapper (profiling events, tape recording), XCD-band launch geometry with compact tile lists, division-mode dispatch,
// argument checks.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fs_host.h"

namespace fs {

hipEvent_t prof_event(fs_ctx *c);      // fs_core.hip


// Every kernel launch of the library goes through here.  The callable captures its arguments BY VALUE: while a tape is being
// recorded (fs_tape_begin) a copy is kept and re-issued by fs_tape_replay without going back through the caller.
template <typename F>
inline int launch(fs_ctx *c, const char *name, F &&f)
{
    if (c->tape_rec) {
        c->tape_rec->ops.emplace_back([f]() -> int {
            f();
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? FS_OK : hip_fail(e, "tape replay", __FILE__, __LINE__);
        });
        if (!c->tape_execute) return FS_OK;
    }
    const bool prof = c->prof_on && !c->capturing;
    ProfRec rec{};
    if (prof) {
        auto it = c->prof_ids.find(name);
        if (it == c->prof_ids.end()) {
            it = c->prof_ids.emplace(name, (int)c->prof_names.size()).first;
            c->prof_names.push_back(name);
            c->prof_launches.push_back(0);
            c->prof_ms.push_back(0.0);
            c->prof_kernels.emplace_back();
        }
        rec.name_id = it->second;
        rec.start = prof_event(c);
        rec.stop = prof_event(c);
        (void)hipEventRecord(rec.start, c->stream);
    }
    kernel_notes.n = 0;
    f();
    if (prof) {
        (void)hipEventRecord(rec.stop, c->stream);
        c->prof_recs.push_back(rec);
        auto &ks = c->prof_kernels[rec.name_id];
        for (int i = 0; i < kernel_notes.n; ++i)
            if (std::find(ks.begin(), ks.end(), kernel_notes.fn[i]) == ks.end()) ks.push_back(kernel_notes.fn[i]);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name, __FILE__, __LINE__);
    return FS_OK;
}

// Grids below 2 M cells have few waves per SIMD: a launch takes as long as ONE wave's chain of loads, stages and stores, and
// tiles of half the height halve that chain (round 4, tools/r4_chain.py; env FS_SMALL_CELLS=0: the big grids' tile heights everywhere)
static inline bool small_tiles(const fs_ctx *c) { return (size_t)c->X * c->Y < c->small_cells; }

// a launch over every row of a single-GPU grid (what may clear a buffer's "hot" word [3], fs_device.h)
static inline int whole_grid(const fs_ctx *c, int jb, int je) { return c->halo == 0 && jb == 0 && je == c->rows ? 1 : 0; }

static inline dim3 cells_grid(const fs_ctx *c, int jb, int je) { return dim3((c->X + 255) / 256, je - jb, 1); }

// overlapped-wave tile kernels: nbx blocks of 4 waves x 62 quads across, nby tile rows, XCD-band 1-D launch
struct OvGrid { int nbx, nby; dim3 grid; Grid g; int threads = 256; };    // threads: 64 x waves per workgroup
enum { XCD_RBSOR = 1, XCD_VORT = 2, XCD_ADVECT = 4, XCD_NONADV = 8, XCD_GRAD = 16, XCD_JACOBI = 32 };
// Compact list of the workgroups of a dense XCD-band launch that have anything to do (Grid::tiles), built once per geometry from the
// host-side activity maps of the scene.  lanes = cells per lane (4: wave columns of 248 cells, 2: of 120), rt = rows per tile.
// cls: 0 = every workgroup with work; 1 / 2 = those whose tiles see nothing but fluid within `reach` rows and the halo lanes ("plain":
// no mask loads, no boundary views - their own kernel and register budget) / the others
// `lanes` names the wave geometry: 4 = quads, 62 owner lanes (248 cells, 4 halo cells per side); 2 = pairs, 60 owner lanes (120 cells, 4 halo
// cells); 3 = pairs, 62 owner lanes (124 cells, 2 halo cells)
static inline int geo_cells(int lanes) { return lanes == 4 ? 4 : 2; }
static inline int geo_owners(int lanes) { return lanes == 2 ? 60 : 62; }
const fs_ctx::TileList *tile_list(fs_ctx *c, int lanes, int rt, bool stacked, int group, int nbx, int nby, int cls = 0, int reach = 0, int wgw = 4, int jb = 0, int je = -1, int parent_rt = 0);      // fs_core.hip

// XCD-band launch geometry of a tile kernel family (fs_march.h band_coords); `lanes`: cells per lane.  When the launch covers the whole
// single-GPU grid, the workgroups without anything to do are left out (compact list, Grid::tiles).
static inline OvGrid ov_grid_lanes(fs_ctx *c, int jb, int je, int rt, int zgroups, int family, int lanes, bool allow_list = true, int cls = 0, int reach = 0, int wgw = 4, int parent_rt = 0,
                                   bool slab_classes = false)      // slab_classes: plain / boundary lists (cls 1 / 2) for a row range of a slab too (fs_cip_step)
{
    OvGrid o;
    o.g = c->grid();
    // Round 6: the hinted single-launch kernels of the pressure families run as ONE-wave workgroups - a list entry is then one tile, and the entries whose tile
    // takes the masked body stand first in every XCD's list (fs_core.hip tile_list), the all-fluid tiles fill in behind them: the four-sweep Jacobi pass at bc2 res
    // 1600 30.2 -> 23.9 us (configs[1] 2 057 -> 2 472 steps/s), the finishing pass 24.2 -> 21.2, the small grids' red-black pair 12.7 -> 12.0 (res 400) / 12.3 ->
    // 11.6 (configs[0]: 59.6 -> 62.0 k steps/s), the graded literal sweep 73.6 -> 72.1 (three A/B rounds).  Vorticity confinement and K2' measured no gain
    // (85.5 against 84.9; 155.4 against 154.8) and keep their 4-wave workgroups (column-by-column locality).
    if (c->dtype == 0 && (family & (XCD_RBSOR | XCD_JACOBI)) && wgw == 4 && cls == 0 && reach > 0 && allow_list) wgw = 1;      // (f32: the f64 bodies were not measured)
    const int ow = geo_owners(lanes);
    const int nu = c->X / geo_cells(lanes), waves = (nu + ow - 1) / ow, tiles = (je - jb + rt - 1) / rt;
    const bool stacked = (c->stack_mask & family) != 0 && wgw > 1;    // the 4 waves of a workgroup: 4 tile rows of one wave column
    o.threads = 64 * wgw;
    o.nbx = stacked ? waves : (waves + wgw - 1) / wgw;
    o.nby = stacked ? (tiles + wgw - 1) / wgw : tiles;
    {
        // (8 * block columns, rows per XCD group * channel groups, groups per XCD): decoded without a division (fs_march.h band_coords).  Groups of
        // 8 tile rows per XCD measured best for every family (2 / 4 / 16 / 32 / per-family sizes: rounds 2 - 4)
        constexpr int xg = 8;
        const int group = stacked ? std::max(1, xg / wgw) : xg;     // the same number of field rows per XCD group
        const int groups = (o.nby + group - 1) / group;
        const fs_ctx::TileList *tl = allow_list && (c->tile_list_mask & family) && ((jb == 0 && je == c->rows) || (c->halo != 0 && (cls == 0 || slab_classes)))
                                         ? tile_list(c, lanes, rt, stacked, group, o.nbx, o.nby, cls, reach, wgw, jb, je, parent_rt) : nullptr;
        const bool inner = zgroups > 1 && tl;
        if (tl) { o.grid = dim3(8 * tl->per_xcd * zgroups, 1, 1); o.g.tiles = tl->d; }
        else o.grid = dim3(8 * o.nbx, group * zgroups, (groups + 7) / 8);
        o.nby |= (group - 1) << 24;
        if (inner) o.nby |= FS_CG_INNER;
    }
    if (stacked) o.nby |= FS_STACKED;
    return o;
}
static inline OvGrid ov_grid(fs_ctx *c, int jb, int je, int rt, int zgroups, int family, bool allow_list = true)
{ return ov_grid_lanes(c, jb, je, rt, zgroups, family, 4, allow_list); }
template <int N>
static OvGrid ov_grid_n(fs_ctx *c, int jb, int je, int rt) { return ov_grid_lanes(c, jb, je, rt, 1, XCD_RBSOR, N); }

// Division-mode dispatch (fs_device.h DM_*): CALL(DM) is expanded for the modes a kernel family distinguishes.  f32 fields divide by their
// loop-invariant divisors through the f64 multiplication (modes 4 / 5; divisors that admit a tie: IEEE division, modes 0 / 1); power-of-two dx-derived
// divisors by exact multiplication (bit 0).
#define FS_F32_ONLY(dm, bits, CALL, MODE) if constexpr (std::is_same<T, float>::value) { if (((dm) & 7) == (bits)) { CALL(MODE); break; } }
#define FS_DMC(dm, CALL)      /* modes 0 / 4 : no dx-derived divisor                */ \
    do { FS_F32_ONLY(dm, 4, CALL, 4) CALL(0); } while (0)
#define FS_DMX(dm, CALL)      /* modes 0 / 1 / 4 : dx-derived divisors only         */ \
    do { if ((dm) & 1) { CALL(1); break; } FS_F32_ONLY(dm, 4, CALL, 4) CALL(0); } while (0)
#define FS_DMA(dm, CALL)      /* modes 0 / 1 / 4 / 5 : both kinds                   */ \
    do { FS_F32_ONLY(dm, 5, CALL, 5) FS_F32_ONLY(dm, 4, CALL, 4) if ((dm) & 1) { CALL(1); break; } CALL(0); } while (0)

int check_rows(const fs_ctx *c, int jb, int je);                               // fs_core.hip
int check_field(const fs_ctx *c, const fs_field *f, int C, const char *what);
int ensure_stage(fs_ctx *c, size_t bytes);

#define FS_FIELD(f, C)                                             \
    do {                                                           \
        int rc__ = fs::check_field(ctx, f, C, #f);                 \
        if (rc__) return rc__;                                     \
    } while (0)
#define FS_ROWS()                                                  \
    do {                                                           \
        int rc__ = fs::check_rows(ctx, row_begin, row_end);        \
        if (rc__) return rc__;                                     \
        if (!ctx->mask_set) { fs::set_error("mask not uploaded"); return FS_ERR_STATE; } \
        if (row_begin == row_end) return FS_OK;                    \
    } while (0)

// dispatch on ctx dtype: BODY sees `T`
#define FS_DISPATCH(ctx, ...)                                      \
    if ((ctx)->dtype == 0) { using T = float; __VA_ARGS__ }        \
    else { using T = double; __VA_ARGS__ }

#define FS_LAUNCH_CELLS(name, kern, ...)                                                                   \
    return launch(ctx, name, [=] {                                                                         \
        FS_KLAUNCH(kern, cells_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, __VA_ARGS__); \
    });

}  // namespace fs
