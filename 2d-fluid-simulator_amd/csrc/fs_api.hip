// fs_api.hip - C-ABI entry points (include/fs_hip.h): contexts, fields, scene upload, kernel launches.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>
#include <unordered_map>

#include "fs_host.h"

namespace fs {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_err = buf;
    return FS_ERR_HIP;
}

// ---- f64-multiply division (fs_device.h f64div): the identity checked ON THE DEVICE for one divisor -------------------------------
// every significand of x in 9 binades (tiny, denormal quotients, huge), both signs; the dividends x = d (m + 1/2) 2^-149 whose quotient
// is (when the product is an f32 number: exactly) a TIE between two denormals - the case an unguarded f64 product gets wrong; plus 2^24
// arbitrary bit patterns (NaN compared as NaN)
__global__ __launch_bounds__(256) static void k_verify_f64div(float d, double rd, int guarded, unsigned *bad)
{
    const unsigned m = blockIdx.x * 256u + threadIdx.x;          // 2^23 significands; blockIdx.y: the binade / the ties / the random sweeps
    float x;
    if (blockIdx.y < 9) {
        const int e[9] = {0, -60, -100, -126, 60, 100, -20, 20, 127};
        x = __uint_as_float(0x3f800000u | m);
        x = ldexpf(x, e[blockIdx.y]);
        if (blockIdx.y == 3) x = __uint_as_float(m);             // the denormals themselves
    } else if (blockIdx.y == 9) {
        x = (float)(ldexp((double)m + 0.5, -149) * (double)fabsf(d));   // (m + 1/2) ulp_denormal * |d|: exact in f64, an f32 number for many m
    } else {
        unsigned h = (m + 0x9e3779b9u * (blockIdx.y - 9u)) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x = __uint_as_float(h);
    }
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
        const float xs = sgn ? -x : x;
        const float q = guarded ? f64div_guarded(xs, d, rd) : f64div(xs, rd), t = xs / d;
        const bool same = __float_as_uint(q) == __float_as_uint(t) || (q != q && t != t);
        if (!same) atomicAdd(bad, 1u);
    }
}

// ---- launch helper: optional HIP-event pair around every launch (fs_prof_*) --------------------
static hipEvent_t prof_event(fs_ctx *c)
{
    hipEvent_t e;
    if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
    hipEventCreate(&e);
    return e;
}

// Every kernel launch of the library goes through here.  The callable captures its arguments BY VALUE: while a tape is being
// recorded (fs_tape_begin) a copy is kept and re-issued by fs_tape_replay without going back through the caller.
template <typename F>
static int launch(fs_ctx *c, const char *name, F &&f)
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
        }
        rec.name_id = it->second;
        rec.start = prof_event(c);
        rec.stop = prof_event(c);
        (void)hipEventRecord(rec.start, c->stream);
    }
    f();
    if (prof) {
        (void)hipEventRecord(rec.stop, c->stream);
        c->prof_recs.push_back(rec);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name, __FILE__, __LINE__);
    return FS_OK;
}

// the same event pair for work that is not a single kernel launch (fs_comm.hip: the pack -> RCCL -> unpack chain of a ghost-row exchange),
// on the stream that work is queued on
ProfRec prof_span_begin(fs_ctx *c, const char *name, hipStream_t stream)
{
    ProfRec rec{};
    rec.name_id = -1;
    if (!(c->prof_on && !c->capturing)) return rec;
    auto it = c->prof_ids.find(name);
    if (it == c->prof_ids.end()) {
        it = c->prof_ids.emplace(name, (int)c->prof_names.size()).first;
        c->prof_names.push_back(name);
        c->prof_launches.push_back(0);
        c->prof_ms.push_back(0.0);
    }
    rec.name_id = it->second;
    rec.start = prof_event(c);
    rec.stop = prof_event(c);
    (void)hipEventRecord(rec.start, stream);
    return rec;
}
void prof_span_end(fs_ctx *c, const ProfRec &rec, hipStream_t stream)
{
    if (rec.name_id < 0) return;
    (void)hipEventRecord(rec.stop, stream);
    c->prof_recs.push_back(rec);
}

static int prof_drain(fs_ctx *c)
{
    if (c->prof_recs.empty()) return FS_OK;
    FS_HIP(hipStreamSynchronize(c->stream));
    for (auto &r : c->prof_recs) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.start, r.stop);
        c->prof_launches[r.name_id] += 1;
        c->prof_ms[r.name_id] += ms;
        c->prof_pool.push_back(r.start);
        c->prof_pool.push_back(r.stop);
    }
    c->prof_recs.clear();
    return FS_OK;
}

// Grids below 2 M cells have few waves per SIMD: a launch takes as long as ONE wave's chain of loads, stages and stores, and
// tiles of half the height halve that chain (round 4, tools/r4_chain.py; env FS_SMALL_TILES=0: the big grids' tile heights everywhere)
static inline bool small_tiles(const fs_ctx *c) { return c->small_tiles && (size_t)c->X * c->Y < c->small_cells; }

// a launch over every row of a single-GPU grid (what may clear a buffer's "hot" word [3], fs_device.h)
static inline int whole_grid(const fs_ctx *c, int jb, int je) { return c->halo == 0 && jb == 0 && je == c->rows ? 1 : 0; }

static inline dim3 cells_grid(const fs_ctx *c, int jb, int je) { return dim3((c->X + 255) / 256, je - jb, 1); }

// overlapped-wave tile kernels: nbx blocks of 4 waves x 62 quads across, nby tile rows, XCD-band 1-D launch
struct OvGrid { int nbx, nby; dim3 grid; Grid g; int threads = 256; };    // threads: 64 x waves per workgroup
enum { XCD_RBSOR = 1, XCD_VORT = 2, XCD_ADVECT = 4, XCD_NONADV = 8, XCD_GRAD = 16, XCD_JACOBI = 32, XCD_MARCH = 64 };   // XCD_MARCH: the row-marching passes - one strip row per XCD group, workgroups side by side
// Compact list of the workgroups of a dense XCD-band launch that have anything to do (Grid::tiles), built once per geometry from the
// host-side activity maps of the scene.  lanes = cells per lane (4: wave columns of 248 cells, 2: of 120), rt = rows per tile.
// cls: 0 = every workgroup with work; 1 / 2 = those whose tiles see nothing but fluid within `reach` rows and the halo lanes ("plain":
// no mask loads, no boundary views - their own kernel and register budget) / the others
// `lanes` names the wave geometry: 4 = quads, 62 owner lanes (248 cells, 4 halo cells per side); 2 = pairs, 60 owner lanes (120 cells, 4 halo
// cells); 3 = pairs, 62 owner lanes (124 cells, 2 halo cells)
static inline int geo_cells(int lanes) { return lanes == 4 ? 4 : 2; }
static inline int geo_owners(int lanes) { return lanes == 2 ? 60 : (lanes == 5 ? 58 : (lanes == 6 ? 56 : 62)); }   // 5 / 6: pairs with 3 / 4 halo lanes per side (the 6- / 8-sweep marching passes; no compact lists)
static const fs_ctx::TileList *tile_list(fs_ctx *c, int lanes, int rt, bool stacked, int group, int nbx, int nby, int cls = 0, int reach = 0, int wgw = 4, int jb = 0, int je = -1, int parent_rt = 0)
{
    if (c->h_act4.empty() || nbx > 0xfff || nby > 0xffff || lanes > 4 || c->rows > 0xffff) return nullptr;      // (entry: class hints << 28 | by << 12 | bx)
    if (je < 0) je = c->rows;
    const uint32_t key0 = (uint32_t)lanes | ((uint32_t)rt << 4) | ((uint32_t)stacked << 12) | ((uint32_t)group << 16) | ((uint32_t)cls << 24) | ((uint32_t)reach << 26) | ((uint32_t)(wgw & 7) << 29);
    if (parent_rt == rt) parent_rt = 0;
    if (parent_rt && (wgw != 1 || parent_rt % rt != 0 || parent_rt > 64)) return nullptr;      // (a coarser plain tiling is defined for one-wave workgroups)
    const std::pair<uint32_t, uint32_t> key(key0 ^ ((uint32_t)parent_rt << 5), ((uint32_t)jb << 16) | (uint32_t)je);      // slab launches cover varying row ranges: one list per range
    auto it = c->tile_lists.find(key);
    if (it != c->tile_lists.end()) return it->second.d ? &it->second : nullptr;
    if (c->capturing || c->tape_rec) return nullptr;      // (building one synchronises the stream: not inside a capture - the dense grid then)
    const std::vector<uint8_t> &act = lanes == 4 ? c->h_act4 : (lanes == 2 ? c->h_act2 : c->h_act2w);
    const int ow = geo_owners(lanes), waves = (c->X / geo_cells(lanes) + ow - 1) / ow, Y = c->rows;       // (activity maps are indexed by LOCAL row)
    std::vector<uint32_t> per[8];
    bool any_hint = false;
    const int groups = (nby + group - 1) / group;
    // inside a group the workgroups are listed column by column: vertically adjacent workgroups, which re-read each other's halo rows, are
    // neighbours in dispatch order (bc5 res 4096: K3+K4 333 -> 319 us, the red-black pair 195 -> 191; FS_LIST_ROWMAJOR=1: row by row)
    const bool col_major = getenv("FS_LIST_ROWMAJOR") == nullptr;
    for (int xcd = 0; xcd < 8; ++xcd)
        for (int lg = 0; lg * 8 + xcd < groups; ++lg)
            for (int o = 0; o < group * nbx; ++o) {
                const int ly = col_major ? o % group : o / nbx, bx = col_major ? o / group : o % nbx;
                const int by = (lg * 8 + xcd) * group + ly;
                if (by >= nby) continue;
                {
                    // wave columns / rows of this workgroup (4 waves: side by side, or stacked = 4 tile rows of one column)
                    const int wx0 = stacked ? bx : bx * wgw, wx1 = std::min(waves, stacked ? bx + 1 : bx * wgw + wgw);
                    const int j0 = jb + (stacked ? by * wgw : by) * rt, j1 = std::min(je, jb + (stacked ? by * wgw + wgw : by + 1) * rt);
                    bool any = false;
                    for (int wx = wx0; wx < wx1 && !any; ++wx)
                        for (int j = j0; j < j1; ++j)
                            if (act[(size_t)wx * Y + j] & 1) { any = true; break; }
                    if (any && cls) {
                        // plain: no non-fluid cell (bit 1 of the activity byte; halo lanes included) within `reach` rows of the tile - or, for the
                        // boundary list of a launch whose plain part runs on tiles of parent_rt rows, of the parent tile this tile lies in
                        int p0 = j0, p1 = j1;
                        if (parent_rt) { p0 = jb + (j0 - jb) / parent_rt * parent_rt; p1 = std::min(je, p0 + parent_rt); }
                        bool plain = true;
                        for (int wx = wx0; wx < wx1 && plain; ++wx)
                            for (int j = std::max(0, p0 - reach); j < std::min(Y, p1 + reach); ++j)
                                if (act[(size_t)wx * Y + j] & 2) { plain = false; break; }
                        any = plain == (cls == 1);
                    }
                    uint32_t hints = 0u;
                    if (any && !cls && reach > 0 && wgw <= 4) {
                        // per-wave hint for a kernel that holds both paths (unsplit launches): wave w is plain - no non-fluid cell within `reach` rows
                        // of ITS tile, halo lanes included - and may skip its mask loads and the classification (band_coords cls)
                        for (int w = 0; w < wgw; ++w) {
                            const int wx = stacked ? bx : bx * wgw + w;
                            const int t0 = jb + (stacked ? by * wgw + w : by) * rt, t1 = std::min(je, t0 + rt);
                            if (wx >= waves || t0 >= je) continue;
                            bool plain = true;
                            for (int j = std::max(0, t0 - reach); j < std::min(Y, t1 + reach) && plain; ++j)
                                if (act[(size_t)wx * Y + j] & 2) plain = false;
                            if (plain) hints |= 1u << w;
                        }
                        any_hint = any_hint || hints != 0u;
                    }
                    if (any) per[xcd].push_back((hints << 28) | ((uint32_t)by << 12) | (uint32_t)bx);
                }
            }
    size_t K = 0, total = 0;
    for (auto &v : per) { K = std::max(K, v.size()); total += v.size(); }
    fs_ctx::TileList tl;
    if (K > 0 && (cls || any_hint || total < (size_t)nbx * nby)) {        // (nothing to skip, no hint to give: the dense grid needs no list)
        std::vector<uint32_t> h(K * 8, 0xffffffffu);
        for (int xcd = 0; xcd < 8; ++xcd)
            for (size_t k = 0; k < per[xcd].size(); ++k) h[k * 8 + xcd] = per[xcd][k];
        if (hipMalloc(&tl.d, h.size() * sizeof(uint32_t)) == hipSuccess &&
            hipMemcpyAsync(tl.d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess)
            tl.per_xcd = (int)K;
        else { if (tl.d) hipFree(tl.d); tl.d = nullptr; }
    }
    auto &slot = c->tile_lists[key] = tl;
    return slot.d ? &slot : nullptr;
}

static void tile_lists_free(fs_ctx *c)
{
    for (auto &kv : c->tile_lists) if (kv.second.d) hipFree(kv.second.d);
    c->tile_lists.clear();
}

// XCD-band launch geometry of a tile kernel family (fs_march.h band_coords); `lanes`: cells per lane.  When the launch covers the whole
// single-GPU grid, the workgroups without anything to do are left out (compact list, Grid::tiles).
static inline OvGrid ov_grid_lanes(fs_ctx *c, int jb, int je, int rt, int zgroups, int family, int lanes, bool allow_list = true, int cls = 0, int reach = 0, int wgw = 4, int parent_rt = 0)
{
    OvGrid o;
    o.g = c->grid();
    const int ow = geo_owners(lanes);
    const int nu = c->X / geo_cells(lanes), waves = (nu + ow - 1) / ow, tiles = (je - jb + rt - 1) / rt;
    const bool stacked = (c->stack_mask & family) != 0;    // the 4 waves of a workgroup: 4 tile rows of one wave column
    o.threads = 64 * wgw;
    o.nbx = stacked ? waves : (waves + wgw - 1) / wgw;
    o.nby = stacked ? (tiles + wgw - 1) / wgw : tiles;
    if (c->xcd_mask & family) {
        // (8 * block columns, rows per XCD group * channel groups, groups per XCD): decoded without a division (fs_march.h band_coords)
        int xg = c->xcd_group;
        for (int f = 0; f < 7; ++f) if ((family >> f) & 1) xg = c->xcd_group_fam[f] > 0 ? c->xcd_group_fam[f] : xg;
        const int group = stacked ? std::max(1, xg / wgw) : xg;     // the same number of field rows per XCD group
        const int groups = (o.nby + group - 1) / group;
        const fs_ctx::TileList *tl = allow_list && (c->tile_list_mask & family) && ((jb == 0 && je == c->rows) || (c->halo != 0 && cls == 0))
                                         ? tile_list(c, lanes, rt, stacked, group, o.nbx, o.nby, cls, reach, wgw, jb, je, parent_rt) : nullptr;
        const bool inner = zgroups > 1 && (tl || (c->cg_inner_mask & family) != 0);
        if (tl) { o.grid = dim3(8 * tl->per_xcd * zgroups, 1, 1); o.g.tiles = tl->d; }
        else o.grid = inner ? dim3(8 * o.nbx * zgroups, group, (groups + 7) / 8) : dim3(8 * o.nbx, group * zgroups, (groups + 7) / 8);
        o.nby |= (group - 1) << 24;
        if (inner) o.nby |= FS_CG_INNER;
    } else { o.grid = dim3(o.nbx * o.nby, zgroups, 1); o.nbx = -o.nbx; }   // negative nbx = row-major decode
    if (stacked) o.nby |= FS_STACKED;
    return o;
}
static inline OvGrid ov_grid(fs_ctx *c, int jb, int je, int rt, int zgroups, int family, bool allow_list = true)
{ return ov_grid_lanes(c, jb, je, rt, zgroups, family, 4, allow_list); }
template <int N>
static OvGrid ov_grid_n(fs_ctx *c, int jb, int je, int rt) { return ov_grid_lanes(c, jb, je, rt, 1, XCD_RBSOR, N); }

// Division-mode dispatch (fs_device.h DM_*): CALL(DM) is expanded for the modes a kernel family distinguishes.  f32 fields divide by their
// loop-invariant divisors through the f64 multiplication (modes 4 / 5; FS_F64DIV=0: IEEE division, modes 0 / 1); power-of-two dx-derived
// divisors by exact multiplication (bit 0).
#define FS_F32_ONLY(dm, bits, CALL, MODE) if constexpr (std::is_same<T, float>::value) { if (((dm) & 7) == (bits)) { CALL(MODE); break; } }
#define FS_DMC(dm, CALL)      /* modes 0 / 4 : no dx-derived divisor                */ \
    do { FS_F32_ONLY(dm, 4, CALL, 4) CALL(0); } while (0)
#define FS_DMX(dm, CALL)      /* modes 0 / 1 / 4 : dx-derived divisors only         */ \
    do { if ((dm) & 1) { CALL(1); break; } FS_F32_ONLY(dm, 4, CALL, 4) CALL(0); } while (0)
#define FS_DMA(dm, CALL)      /* modes 0 / 1 / 4 / 5 : both kinds                   */ \
    do { FS_F32_ONLY(dm, 5, CALL, 5) FS_F32_ONLY(dm, 4, CALL, 4) if ((dm) & 1) { CALL(1); break; } CALL(0); } while (0)

#define FS_PAIR(RT) hipLaunchKernelGGL((k_jacobi_pair<RT, SW, HV, T>), grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (const uint8_t *)ctx->d_lazyflags, list, nlist, zoff, (T *)pn->d, (const T *)pc->d, (const T *)src->d)
template <bool SW, bool HV, typename T>
static void launch_pair(fs_ctx *ctx, const OvGrid &og, int rt, int row_begin, int row_end, fs_field *pn, const fs_field *pc, const fs_field *src)
{
    // the general rows ride in front: `zoff` leading z slices of the same launch, one wave per listed row
    const uint32_t *list = ctx->d_pairlist + (HV ? (size_t)ctx->nwx * ctx->rows : 0);
    const int nlist = ctx->n_pairlist[HV ? 1 : 0];
    const int per_slice = (int)(og.grid.x * og.grid.y), blocks = nlist, zoff = (blocks + per_slice - 1) / per_slice;      // one listed row per workgroup
    const dim3 grid(og.grid.x, og.grid.y, og.grid.z + zoff);
    if (rt == 1) FS_PAIR(1); else if (rt == 4) FS_PAIR(4); else if (rt == 2) FS_PAIR(2); else FS_PAIR(3);
}

template <bool SRC, typename T>
static int launch_jacobi(fs_ctx *ctx, const char *name, const Konst<T> &k, int jb, int je, T *pn, const T *pc, const T *vs)
{
    // overlapped-wave register tiles of 1 - 4 rows (FS_JACOBI=21 .. 24).  Default (0): the source-pair
    // form streams best with 1-row tiles at 8 waves/SIMD (76 vs 79 us), the v-reading form with 2-row tiles (89 vs 95 us)
    const int v = ctx->jacobi_variant ? ctx->jacobi_variant : (SRC ? 21 : 24);      // (round 4, after the DPP diet: 4-row tiles for the v-reading form: 84.7 against 85.9-86.4 us)
    const int rt = v == 24 ? 4 : (v == 21 ? 1 : (v == 23 ? 3 : 2));
    const OvGrid og = ov_grid(ctx, jb, je, rt, 1, XCD_JACOBI);
    const int dm = SRC ? 0 : dm_const(ctx, k);           // the source-pair form divides nothing
#define FS_JAC(DM) do { \
        if (rt == 2) hipLaunchKernelGGL((k_jacobi_ov<SRC, 2, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs); \
        else if (rt == 3) hipLaunchKernelGGL((k_jacobi_ov<SRC, 3, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs); \
        else if (rt == 4) hipLaunchKernelGGL((k_jacobi_ov<SRC, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs); \
        else hipLaunchKernelGGL((k_jacobi_ov<SRC, 1, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, pn, pc, vs); } while (0)
    return launch(ctx, name, [=] { FS_DMC(dm, FS_JAC); });
}

static int check_rows(const fs_ctx *c, int jb, int je)
{
    if (!(0 <= jb && jb <= je && je <= c->rows)) {
        set_error("row range outside the local slab");
        return FS_ERR_ARG;
    }
    return FS_OK;
}

static int check_field(const fs_ctx *c, const fs_field *f, int C, const char *what)
{
    if (!f || f->ctx != c || f->C != C) {
        set_error(std::string("field argument '") + what + "' is null, from another context, or has the wrong channel count");
        return FS_ERR_ARG;
    }
    return FS_OK;
}

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

static int ensure_stage(fs_ctx *c, size_t bytes)
{
    if (c->stage_bytes >= bytes) return FS_OK;
    if (c->d_stage) { FS_HIP(hipStreamSynchronize(c->stream)); FS_HIP(hipFree(c->d_stage)); c->d_stage = nullptr; c->stage_bytes = 0; }
    FS_HIP(hipMalloc(&c->d_stage, bytes));
    c->stage_bytes = bytes;
    return FS_OK;
}

// ---- boundary-condition op lists (host analysis of the global mask) ----------------------------
struct HostOp { int kind; long long t, s1, s2; };  // cells as global (i*Y + j) ids

struct DSU {
    std::vector<int> p;
    explicit DSU(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0); }
    int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

static void free_ops(BcOpsDev &o)
{
    int **ptrs[] = {&o.comp_begin, &o.comp_rlo, &o.comp_rhi, &o.kind, &o.tgt, &o.s1, &o.s2, &o.row, &o.srow};
    for (auto pp : ptrs) { if (*pp) hipFree(*pp); *pp = nullptr; }
    if (o.simple) hipFree(o.simple);
    if (o.pair) hipFree(o.pair);
    o.simple = o.pair = nullptr;
    o.nsimple = o.npair = o.ncomp = o.nops = 0;
}

// Group the serial-order op list into hazard components and upload it in local cell offsets.
// `rows_in_z`: the simple-op record carries the ROW of source 1 in .z (velocity field: 2 channels, element offsets need the row)
// instead of source 2 (pressure: offsets are element offsets as they are).
static int upload_ops(fs_ctx *c, const std::vector<HostOp> &ops, BcOpsDev &out, int &reach, int min_radius, bool rows_in_z)
{
    free_ops(out);
    const int n = (int)ops.size();
    const int Y = c->Y;
    DSU dsu(n);
    {
        std::unordered_map<long long, int> last_writer;
        std::unordered_map<long long, std::vector<int>> readers;
        last_writer.reserve(n * 2);
        readers.reserve(n * 2);
        for (int o = 0; o < n; ++o) {
            const long long srcs[2] = {ops[o].s1, ops[o].s2};
            for (long long s : srcs) {
                if (s < 0) continue;
                auto w = last_writer.find(s);
                if (w != last_writer.end()) dsu.unite(o, w->second);   // read after write
                readers[s].push_back(o);
            }
            const long long t = ops[o].t;
            auto w = last_writer.find(t);
            if (w != last_writer.end()) dsu.unite(o, w->second);       // write after write
            auto r = readers.find(t);
            if (r != readers.end())
                for (int q : r->second) if (q != o) dsu.unite(o, q);  // write after read
            last_writer[t] = o;
        }
    }
    // components in order of their first op; ops inside a component keep serial order
    std::vector<int> root(n), order(n);
    for (int o = 0; o < n; ++o) root[o] = dsu.find(o);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return root[a] < root[b]; });

    auto local_row = [&](long long cell) { return (int)(cell % Y) - c->y0 + c->halo; };
    auto local_off = [&](long long cell) { return local_row(cell) * c->P + (int)(cell / Y); };

    std::vector<int> h_begin, h_rlo, h_rhi, h_kind, h_tgt, h_s1, h_s2, h_row, h_srow;
    std::vector<int4> h_simple, h_pair;
    auto record = [&](const HostOp &op) {
        const int tr = local_row(op.t);
        int4 r;
        r.x = local_off(op.t);
        r.y = op.s1 >= 0 ? local_off(op.s1) : -1;
        r.z = rows_in_z ? (op.s1 >= 0 ? local_row(op.s1) : 0) : (op.s2 >= 0 ? local_off(op.s2) : -1);
        r.w = op.kind | (tr << 2);
        return r;
    };
    int pos = 0;
    while (pos < n) {
        int end = pos;
        while (end < n && root[order[end]] == root[order[pos]]) ++end;
        int tlo = INT32_MAX, thi = INT32_MIN, clo = INT32_MAX, chi = INT32_MIN;
        // dependency span: rows of PRE-kernel data each rewritten cell depends on (chains through earlier ops of the
        // component included).  The largest |target row - dependency row| is the stencil radius a slab needs.
        std::unordered_map<long long, std::pair<int, int>> dep;
        for (int q = pos; q < end; ++q) {
            const HostOp &op = ops[order[q]];
            int tr = local_row(op.t);
            tlo = std::min(tlo, tr); thi = std::max(thi, tr);
            clo = std::min(clo, tr); chi = std::max(chi, tr);
            int dlo = INT32_MAX, dhi = INT32_MIN;
            const long long srcs[2] = {op.s1, op.s2};
            for (long long s : srcs) if (s >= 0) {
                int sr = local_row(s);
                clo = std::min(clo, sr); chi = std::max(chi, sr);
                auto d = dep.find(s);
                if (d != dep.end()) { dlo = std::min(dlo, d->second.first); dhi = std::max(dhi, d->second.second); }
                else { dlo = std::min(dlo, sr); dhi = std::max(dhi, sr); }
            }
            if (dlo <= dhi) {
                reach = std::max(reach, std::max(tr - dlo, dhi - tr));
                dep[op.t] = {dlo, dhi};
            } else {
                dep[op.t] = {tr, tr};   // constant assignment (inflow value, p = 0)
            }
        }
        // A slab can evaluate an assignment when its target and its sources lie in its local rows and no source is a cell that an
        // earlier, non-evaluable assignment of the chain should have rewritten.  A hazard component that leaves the local rows (thin
        // walls: the mirror of a one-cell wall overwrites a FLUID cell that a later mirror reads) is therefore pruned to its evaluable
        // prefix relations instead of being dropped as a whole - dropping it left ghost rows un-updated that the validity tracker
        // (fs/runtime.py) counts as correct to depth halo - radius.  Cells that end up with an unknowable value ("bad") must lie
        // deeper than that, otherwise this decomposition is refused.
        (void)clo; (void)chi;
        std::unordered_map<long long, bool> bad;
        std::vector<int> kept;
        for (int q = pos; q < end; ++q) {
            const HostOp &op = ops[order[q]];
            const int tr = local_row(op.t);
            bool ok = tr >= 0 && tr < c->rows;
            const long long srcs[2] = {op.s1, op.s2};
            for (long long s : srcs) if (s >= 0) {
                const int sr = local_row(s);
                auto b = bad.find(s);
                if (sr < 0 || sr >= c->rows || (b != bad.end() && b->second)) ok = false;
            }
            if (ok) { kept.push_back(order[q]); bad[op.t] = false; }
            else bad[op.t] = true;
        }
        for (const auto &b : bad) {
            if (!b.second) continue;
            const int r = local_row(b.first);
            if (r < 0 || r >= c->rows) continue;
            const int depth = r < c->halo ? c->halo - r : (r >= c->halo + c->nyl ? r - (c->halo + c->nyl) + 1 : 0);
            if (depth <= c->halo - min_radius) c->bc_incomplete = true;   // an owned row, or a ghost row the tracker may rely on
        }
        if (kept.size() == 1) {        // the common case: one assignment, no hazard -> a flat 16-byte record
            h_simple.push_back(record(ops[kept[0]]));
        } else if (kept.size() == 2 && !getenv("FS_BC_NOPAIRS")) {      // a chain of two: two flat records side by side (BcOps::pair)
            h_pair.push_back(record(ops[kept[0]]));
            h_pair.push_back(record(ops[kept[1]]));
        } else if (!kept.empty()) {
            int klo = INT32_MAX, khi = INT32_MIN;
            for (int o : kept) { const int tr = local_row(ops[o].t); klo = std::min(klo, tr); khi = std::max(khi, tr); }
            h_begin.push_back((int)h_kind.size());
            h_rlo.push_back(klo);
            h_rhi.push_back(khi);
            for (int o : kept) {
                const HostOp &op = ops[o];
                h_kind.push_back(op.kind);
                h_tgt.push_back(local_off(op.t));
                h_s1.push_back(op.s1 >= 0 ? local_off(op.s1) : -1);
                h_s2.push_back(op.s2 >= 0 ? local_off(op.s2) : -1);
                h_row.push_back(local_row(op.t));
                h_srow.push_back(op.s1 >= 0 ? local_row(op.s1) : 0);
            }
        }
        pos = end;
    }
    h_begin.push_back((int)h_kind.size());
    if (getenv("FS_BC_STATS")) {        // (debug: how long the serial chains of this op list are)
        std::map<int, int> hist;
        for (size_t k = 0; k + 1 < h_begin.size(); ++k) hist[h_begin[k + 1] - h_begin[k]]++;
        fprintf(stderr, "fs: bc op list: %zu simple, %zu chains of two, %zu longer chains:", h_simple.size(), h_pair.size() / 2, h_rlo.size());
        for (auto &kv : hist) fprintf(stderr, " %dx len %d", kv.second, kv.first);
        fprintf(stderr, "\n");
    }
    out.ncomp = (int)h_rlo.size();
    out.nops = (int)h_kind.size();
    // Stream-ordered like every other memory operation of a context: its stream is non-blocking, so work on the null stream
    // (hipMemcpy / hipMemset) is NOT ordered against the kernels that follow.  The host vectors outlive the copies (sync below).
    auto up = [&](int *&d, const std::vector<int> &h) -> int {
        FS_HIP(hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(int)));
        if (!h.empty()) FS_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        return FS_OK;
    };
    int rc;
    if ((rc = up(out.comp_begin, h_begin))) return rc;
    if ((rc = up(out.comp_rlo, h_rlo))) return rc;
    if ((rc = up(out.comp_rhi, h_rhi))) return rc;
    if ((rc = up(out.kind, h_kind))) return rc;
    if ((rc = up(out.tgt, h_tgt))) return rc;
    if ((rc = up(out.s1, h_s1))) return rc;
    if ((rc = up(out.s2, h_s2))) return rc;
    if ((rc = up(out.row, h_row))) return rc;
    if ((rc = up(out.srow, h_srow))) return rc;
    out.nsimple = (int)h_simple.size();
    FS_HIP(hipMalloc(&out.simple, std::max<size_t>(h_simple.size(), 1) * sizeof(int4)));
    if (!h_simple.empty()) FS_HIP(hipMemcpyAsync(out.simple, h_simple.data(), h_simple.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream));
    out.npair = (int)h_pair.size() / 2;
    FS_HIP(hipMalloc(&out.pair, std::max<size_t>(h_pair.size(), 1) * sizeof(int4)));
    if (!h_pair.empty()) FS_HIP(hipMemcpyAsync(out.pair, h_pair.data(), h_pair.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream));
    FS_HIP(hipStreamSynchronize(c->stream));
    return FS_OK;
}

// Enumerate the assignments of the three BC kernels in the reference's serial (i-major, j-minor) order.
// Only cells whose row lies within `margin` rows of this slab are examined.
static int build_bc_ops(fs_ctx *c, const uint8_t *mask)
{
    const int X = c->X, Y = c->Y;
    auto M = [&](int i, int j) -> int { return mask[(size_t)i * Y + j]; };
    auto MO = [&](int i, int j) -> int { return (i < 0 || i >= X || j < 0 || j >= Y) ? 1 : mask[(size_t)i * Y + j]; };  // H3: outside = wall
    auto id = [&](int i, int j) -> long long { return (long long)i * Y + j; };
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int margin = c->halo + 4;
    const int jlo = std::max(0, c->y0 - margin), jhi = std::min(Y, c->y0 + c->nyl + margin);

    std::vector<HostOp> vel, prs, dye;
    for (int i = 0; i < X; ++i)
        for (int j = jlo; j < jhi; ++j) {
            const int m = M(i, j);
            if (m == 0) continue;
            if (m == 1) {
                // fs/boundary_condition.py:20-33 (velocity mirror, interior wall cells only)
                if (1 <= i && i < X - 1 && 1 <= j && j < Y - 1) {
                    if (M(i - 1, j) == 0 && M(i, j - 1) == 1 && M(i, j + 1) == 1) vel.push_back({0, id(i + 1, j), id(i - 1, j), -1});
                    else if (M(i + 1, j) == 0 && M(i, j - 1) == 1 && M(i, j + 1) == 1) vel.push_back({0, id(i - 1, j), id(i + 1, j), -1});
                    else if (M(i, j - 1) == 0 && M(i - 1, j) == 1 && M(i + 1, j) == 1) vel.push_back({0, id(i, j + 1), id(i, j - 1), -1});
                    else if (M(i, j + 1) == 0 && M(i - 1, j) == 1 && M(i + 1, j) == 1) vel.push_back({0, id(i, j - 1), id(i, j + 1), -1});
                }
                // fs/boundary_condition.py:45-61 (pressure: 4 face cases then 4 corner cases); sample() clamps
                const int iw = cl(i - 1, 0, X - 1), ie = cl(i + 1, 0, X - 1), js = cl(j - 1, 0, Y - 1), jn = cl(j + 1, 0, Y - 1);
                if (MO(i - 1, j) == 0 && MO(i, j - 1) == 1 && MO(i, j + 1) == 1) prs.push_back({0, id(i, j), id(iw, j), -1});
                else if (MO(i + 1, j) == 0 && MO(i, j - 1) == 1 && MO(i, j + 1) == 1) prs.push_back({0, id(i, j), id(ie, j), -1});
                else if (MO(i, j - 1) == 0 && MO(i - 1, j) == 1 && MO(i + 1, j) == 1) prs.push_back({0, id(i, j), id(i, js), -1});
                else if (MO(i, j + 1) == 0 && MO(i - 1, j) == 1 && MO(i + 1, j) == 1) prs.push_back({0, id(i, j), id(i, jn), -1});
                else if (MO(i - 1, j) == 0 && MO(i, j + 1) == 0) prs.push_back({1, id(i, j), id(iw, j), id(i, jn)});
                else if (MO(i + 1, j) == 0 && MO(i, j + 1) == 0) prs.push_back({1, id(i, j), id(ie, j), id(i, jn)});
                else if (MO(i - 1, j) == 0 && MO(i, j - 1) == 0) prs.push_back({1, id(i, j), id(iw, j), id(i, js)});
                else if (MO(i + 1, j) == 0 && MO(i, j - 1) == 0) prs.push_back({1, id(i, j), id(ie, j), id(i, js)});
            } else if (m == 2) {
                vel.push_back({1, id(i, j), -1, -1});                              // :34-35  v = bc_const
                prs.push_back({0, id(i, j), id(cl(i + 1, 0, X - 1), j), -1});      // :62-63  p = p[i+1, j]
                dye.push_back({0, id(i, j), -1, -1});                              // :97-99  dye = bc_dye
            } else if (m == 3) {
                vel.push_back({2, id(i, j), id(cl(i - 1, 0, X - 1), j), -1});      // :36-39  v.x = max(v[i-1].x, 0.05)
                prs.push_back({2, id(i, j), -1, -1});                              // :64-65  p = 0
            }
        }
    // "lazy" pressure boundary condition (fs_march.h k_jacobi_lazy): the recipe of every K7 assignment as one byte per cell, and the
    // preconditions under which evaluating it from the raw sweep output is exactly what K7 followed by the sweep computes
    {
        std::vector<uint8_t> map((size_t)X * Y, 0);
        bool ok = true;
        auto dir = [&](long long t, long long s) -> int {      // 0: i-1, 1: i+1, 2: j-1, 3: j+1, -1: anything else
            const long long d = s - t;
            return d == -(long long)Y ? 0 : (d == (long long)Y ? 1 : (d == -1 ? 2 : (d == 1 ? 3 : -1)));
        };
        for (const HostOp &op : prs) {
            if (op.kind == 2) { map[op.t] = 1 | (2 << 1); continue; }
            if (op.s1 == op.t) continue;                                   // clamped onto itself: p[t] = p[t]
            const int d1 = dir(op.t, op.s1), d2 = op.kind == 1 ? dir(op.t, op.s2) : 0;
            if (d1 < 0 || d2 < 0) { ok = false; continue; }
            if (mask[op.s1] == 1 || (op.kind == 1 && mask[op.s2] == 1)) ok = false;      // a wall source would be read from the buffer's history
            map[op.t] = (uint8_t)(1 | (op.kind << 1) | (d1 << 3) | (d2 << 5));
        }
        for (int i = 0; i < X && ok; ++i)                                  // no computed cell may sample a clamped y neighbour
            if (M(i, 0) != 1 || M(i, Y - 1) != 1) ok = false;
        for (const HostOp &op : vel) map[op.t] |= 0x80;                     // bit 7: a cell the velocity boundary kernel writes (fs_k34n.h)
        c->lazy_ok = ok && X % 4 == 0;
        // Two red-black iterations per pass (fs_rbpair.h): additionally no recipe may read a source on the far side of its target as
        // seen from a fluid cell (a wall one cell thick between two fluid regions) - the shrinking-window argument of that kernel
        bool thin = false;
        for (const HostOp &op : prs) {
            if (op.kind == 2) continue;
            const long long srcs[2] = {op.s1, op.kind == 1 ? op.s2 : -1};
            for (long long s : srcs) {
                if (s < 0 || s == op.t) continue;
                const int oi = 2 * (int)(op.t / Y) - (int)(s / Y), oj = 2 * (int)(op.t % Y) - (int)(s % Y);
                if (oi >= 0 && oi < X && oj >= 0 && oj < Y && M(oi, oj) == 0) thin = true;
            }
        }
        c->rb_pair_ok = ok && !thin;            // (lanes of 2 cells: any even width - fs_rbsor_pair_ok adds use_pairs)
        // Four Jacobi sweeps per pass (fs_jquad.h): the same condition with every cell whose RAW value is live as a reader - fluid cells and
        // the sources of recipes (an inflow cell of column 1 is computed by the sweep and read, raw, by the recipe of column 0)
        {
            std::vector<uint8_t> live((size_t)X * Y);
            for (size_t q = 0; q < live.size(); ++q) live[q] = mask[q] == 0;
            for (const HostOp &op : prs) { if (op.s1 >= 0) live[op.s1] = 1; if (op.kind == 1 && op.s2 >= 0) live[op.s2] = 1; }
            bool thin_live = false;
            for (const HostOp &op : prs) {
                if (op.kind == 2) continue;
                const long long srcs[2] = {op.s1, op.kind == 1 ? op.s2 : -1};
                for (long long s : srcs) {
                    if (s < 0 || s == op.t) continue;
                    const int oi = 2 * (int)(op.t / Y) - (int)(s / Y), oj = 2 * (int)(op.t % Y) - (int)(s % Y);
                    if (oi >= 0 && oi < X && oj >= 0 && oj < Y && live[(size_t)oi * Y + oj]) thin_live = true;
                }
            }
            c->jq_ok = ok && !thin_live;
        }
        c->h_bcmap.swap(map);            // uploaded by fs_upload_mask (same transpose path as the mask), then dropped
    }
    c->bc_incomplete = false;
    c->bc_radius_vel = c->bc_radius_prs = 0;
    int rc, dummy = 0;
    // min_radius: the least radius the host tracker charges for the kernel (fs/runtime.py: max(2, .) / max(1, .) / 0)
    if ((rc = upload_ops(c, vel, c->ops_vel, c->bc_radius_vel, 2, true))) return rc;
    if ((rc = upload_ops(c, prs, c->ops_prs, c->bc_radius_prs, 1, false))) return rc;
    if ((rc = upload_ops(c, dye, c->ops_dye, dummy, 0, true))) return rc;
    return FS_OK;
}

}  // namespace fs

using namespace fs;

// K3 + K4 in one pass (fs_k34n.h), velocity (C = 2, v = nullptr) and dye (C = 3): lane width / tile rows from FS_K34_N / FS_K34_RT
// (default: 2 cells per lane, 4 rows; 4 cells per lane: 2 rows), compact two-part launch on large single-GPU grids
template <int C, bool CLAMP>
static int launch_k34(fs_ctx *ctx, const char *name, const char *name_bnd, double dt, double dx, fs_field *f_out, fs_field *gx_out, fs_field *gy_out,
                      const fs_field *fn, const fs_field *fc, const fs_field *gxc, const fs_field *gyc, const fs_field *v, int full, int jb, int je)
{
    using T = float;
    auto k = make_konst<T>(ctx, dt, dx, 1.0);
    const int dm = dm_dx(ctx, k);
    // geometry by grid size unless FS_K34_N / FS_K34_RT say otherwise (K3+K4 of the velocity, us):   2 cells x 4 rows   4 x 2   2 x 2
    //   >= 8 M cells (two-part launch): bc5 res 4096 / bc5 res 2048 / bc2 res 3000                   307 / 103 / 258    325 / 106 / 267   - / 113 / 282
    //   2 - 8 M cells: bc2 res 1600 / bc5 res 1024 (the boundary kernel of 2 x 4 holds 4 waves per SIMD)  90.7 / 26.5   81.6 / 27.8   90.3 / 26.4
    //   smaller: bc2 res 800 / res 400 (workgroups of half the size)                                  28.2 / 14.5        26.6 / 13.7       24.8 / 12.9
    const size_t cells = (size_t)ctx->X * ctx->rows;      // (this context's slab)
    const int N = ctx->X % 4 != 0 ? 2 : (ctx->k34_n ? ctx->k34_n : (cells >= ((size_t)1 << 23) || cells < ((size_t)1 << 21) ? 2 : 4));
    // (below 1 M cells: 1-row tiles for the dye's three channels - a launch is one wave's chain there, fs_ctx::small_tiles; res 400: 17.6 against
    //  17.1 k steps/s with the dye; the velocity's pass stays on 2 rows: 29.0 against 28.1 k)
    const int RT = N == 4 ? 2 : (ctx->k34_rt ? ctx->k34_rt : (cells >= ((size_t)1 << 23) ? 4 : (small_tiles(ctx) && !full && C == 3 ? 1 : 2))), geo = N == 2 ? 3 : 4;
#define FS_K34(NN, R, DM, PL) hipLaunchKernelGGL((k_cip_grad_advect_n<C, NN, R, DM, PL, CLAMP, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, jb, je, \
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
        const OvGrid og = ov_grid_lanes(ctx, jb, je, RT, C, XCD_ADVECT, geo, true, 1, 2, ctx->split_wgw);
        const OvGrid ogb = ov_grid_lanes(ctx, jb, je, RT, C, XCD_ADVECT, geo, true, 2, 2, ctx->split_wgw);
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

int fs_abi_version(void) { return FS_ABI_VERSION; }
const char *fs_last_error(void) { return g_err.c_str(); }

int fs_device_count(int *count)
{
    FS_REQUIRE(count, "count is null");
    FS_HIP(hipGetDeviceCount(count));
    return FS_OK;
}

int fs_create(fs_ctx **out, int device, int nx, int ny, int dtype, int y0, int ny_local, int halo)
{
    FS_REQUIRE(out, "out is null");
    FS_REQUIRE(nx >= 4 && ny >= 4, "grid must be at least 4x4");
    FS_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (f64)");
    FS_REQUIRE(halo >= 0 && ny_local >= 1 && y0 >= 0 && y0 + ny_local <= ny, "bad slab (y0, ny_local, halo)");
    FS_REQUIRE((long long)(ny_local + 2 * halo) * (((long long)nx + 63) / 64 * 64) < (1LL << 31), "slab too large for 32-bit cell offsets");
    FS_HIP(hipSetDevice(device));
    fs_ctx *c = new fs_ctx();
    c->device = device; c->X = nx; c->Y = ny; c->dtype = dtype; c->y0 = y0; c->nyl = ny_local; c->halo = halo;
    c->rows = ny_local + 2 * halo;
    c->P = (nx + 63) / 64 * 64;
    c->Pm = (nx + 63) / 64 * 64;
    c->esize = dtype == 0 ? 4 : 8;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    e = hipMalloc(&c->d_mask, (size_t)c->rows * c->Pm);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_mask, 1, (size_t)c->rows * c->Pm, c->stream);   // never the null stream: see upload_ops
    if (e == hipSuccess) e = hipMalloc(&c->d_acc, 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&c->d_sync, 16 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_sync, 0, 16 * sizeof(unsigned), c->stream);
    c->nwx = (nx / 4 + 61) / 62;
    if (e != hipSuccess) { fs_destroy(c); return hip_fail(e, "hipMalloc(ctx)", __FILE__, __LINE__); }
    if (const char *s = getenv("FS_MARCH")) c->use_march = atoi(s) != 0;
    if (const char *s = getenv("FS_F64DIV")) c->use_f64div = atoi(s) != 0;
    if (const char *s = getenv("FS_TILE_LIST")) c->tile_list_mask = atoi(s);
    if (const char *s = getenv("FS_LIMIT_GATE")) c->limit_gate = atoi(s) != 0;
    if (const char *s = getenv("FS_LAZY_BC")) c->use_lazy = atoi(s) != 0;
    if (const char *s = getenv("FS_PAIR_RT")) { const int v = atoi(s); if (v >= 1 && v <= 4) c->pair_rt = v; }
    if (const char *s = getenv("FS_JQUAD_RT")) { const int v = atoi(s); if (v == 2 || v == 4 || v == 6 || v == 8) c->jquad_rt = v; }
    if (const char *s = getenv("FS_RBPAIR_SPLIT")) c->rbpair_split = atoi(s);
    if (const char *s = getenv("FS_SMALL_TILES")) c->small_tiles = atoi(s) != 0;
    if (const char *s = getenv("FS_SMALL_CELLS")) c->small_cells = (size_t)atoll(s);
    if (const char *s = getenv("FS_RBPAIR_RT")) { const int v = atoi(s); if (v == 2 || v == 4 || v == 6) c->rbpair_rt = v; }
    if (const char *s = getenv("FS_RBPAIR_PLAIN_RT")) { const int v = atoi(s); if (v == 4 || v == 8) c->rbpair_plain_rt = v; }
    if (const char *s = getenv("FS_SPLIT_WGW")) { const int v = atoi(s); if (v == 1 || v == 2 || v == 4) c->split_wgw = v; }
    if (const char *s = getenv("FS_RBMARCH")) c->use_rbmarch = atoi(s);
    if (const char *s = getenv("FS_RBM_L")) { const int v = atoi(s); if (v >= 14 && v <= 254 && (v + 10) % 12 == 0) c->rbm_L = v; }
    if (const char *s = getenv("FS_JM_L")) { const int v = atoi(s); if (v >= 2 && v <= 254 && v % 2 == 0) c->jm_L = v; }
    if (const char *s = getenv("FS_JM_PF")) { const int v = atoi(s); if (v == 1 || v == 3) c->jm_pf = v; }
    if (const char *s = getenv("FS_RBM_PF")) { const int v = atoi(s); if (v == 1 || v == 3) c->rbm_pf = v; }
    if (const char *s = getenv("FS_K34_RT")) c->k34_rt = atoi(s) == 2 ? 2 : (atoi(s) == 4 ? 4 : (atoi(s) == 1 ? 1 : 0));
    if (const char *s = getenv("FS_MAC_RT")) { const int v = atoi(s); c->mac_rt = v == 2 || v == 4 ? v : 0; }
    if (const char *s = getenv("FS_K34_N")) c->k34_n = atoi(s) == 4 ? 4 : (atoi(s) == 2 ? 2 : 0);
    if (const char *s = getenv("FS_JACOBI")) c->jacobi_variant = atoi(s);
    if (const char *s = getenv("FS_PACK_HALO")) c->pack_halo = atoi(s) != 0;
    c->xcd_mask = XCD_RBSOR | XCD_VORT | XCD_ADVECT | XCD_NONADV | XCD_GRAD | XCD_JACOBI | XCD_MARCH;
    if (const char *s = getenv("FS_XCD")) c->xcd_mask = atoi(s);
    c->stack_mask = XCD_RBSOR | XCD_ADVECT | XCD_GRAD;      // measured per family: K4 313 -> 301 us, K3 246 -> 243, RB-SOR 129 -> 127.5; the others lose 1 %
    if (const char *s = getenv("FS_STACK")) c->stack_mask = atoi(s);
    if (const char *s = getenv("FS_CG_INNER")) c->cg_inner_mask = atoi(s);
    if (const char *s = getenv("FS_XCD_GROUP")) { int v = atoi(s); if (v >= 1 && v <= 128) c->xcd_group = v; }
    if (const char *s = getenv("FS_XCD_GROUP_FAM")) {      // "bit:rows,bit:rows": tile rows per XCD group of single kernel families (XCD_* bit numbers 0 .. 5)
        int f, v, n = 0;
        while (sscanf(s, "%d:%d%n", &f, &v, &n) == 2) { if (f >= 0 && f < 7 && v >= 1 && v <= 128) c->xcd_group_fam[f] = v; s += n; if (*s == ',') ++s; else break; }
    }
    c->use_pairs = c->use_march && nx % 2 == 0;   // the kernels on lanes of 2 cells (fs_k34n.h, fs_rbpair.h, fs_jquad.h): any even width, i.e. any `res`
    if (nx % 4 != 0) c->use_march = false;        // quads need 16-byte aligned rows
    *out = c;
    return FS_OK;
}

int fs_destroy(fs_ctx *ctx)
{
    if (!ctx) return FS_OK;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    fs_comm_destroy(ctx);
    for (auto g : ctx->graphs) if (g) hipGraphExecDestroy(g);
    for (auto t : ctx->tapes) delete t;
    delete ctx->tape_rec;
    for (auto &r : ctx->prof_recs) { hipEventDestroy(r.start); hipEventDestroy(r.stop); }
    for (auto e : ctx->prof_pool) hipEventDestroy(e);
    free_ops(ctx->ops_vel); free_ops(ctx->ops_prs); free_ops(ctx->ops_dye);
    tile_lists_free(ctx);
    for (fs_field *f : ctx->fields) { if (f->d) hipFree(f->d); if (f->hot) hipFree(f->hot); delete f; }
    for (fs_field *f : ctx->deferred_free) { if (f->d) hipFree(f->d); if (f->hot) hipFree(f->hot); delete f; }
    ctx->fields.clear();
    if (ctx->d_mask) hipFree(ctx->d_mask);
    if (ctx->d_bc_const) hipFree(ctx->d_bc_const);
    if (ctx->d_bc_dye) hipFree(ctx->d_bc_dye);
    if (ctx->d_stage) hipFree(ctx->d_stage);
    if (ctx->d_acc) hipFree(ctx->d_acc);
    if (ctx->d_sync) hipFree(ctx->d_sync);
    if (ctx->d_bcmap) hipFree(ctx->d_bcmap);
    if (ctx->d_lazyflags) hipFree(ctx->d_lazyflags);
    if (ctx->d_rbcode) hipFree(ctx->d_rbcode);
    if (ctx->d_jcode) hipFree(ctx->d_jcode);
    if (ctx->d_pairlist) hipFree(ctx->d_pairlist);
    if (ctx->d_partial) hipFree(ctx->d_partial);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return FS_OK;
}

int fs_sync(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

int fs_ctx_info(const fs_ctx *ctx, int *nx, int *ny, int *dtype, int *y0, int *ny_local, int *halo, int *pitch)
{
    FS_REQUIRE(ctx, "ctx is null");
    if (nx) *nx = ctx->X;
    if (ny) *ny = ctx->Y;
    if (dtype) *dtype = ctx->dtype;
    if (y0) *y0 = ctx->y0;
    if (ny_local) *ny_local = ctx->nyl;
    if (halo) *halo = ctx->halo;
    if (pitch) *pitch = ctx->P;
    return FS_OK;
}

// ---- window upload / download -------------------------------------------------------------------
// rows [row_begin, row_begin + nrows) that fall outside the global domain are skipped on upload.
static int upload_window(fs_ctx *ctx, void *dev, int C, size_t esize, const void *host, int row_begin, int nrows, int pitch)
{
    FS_REQUIRE(host, "host pointer is null");
    FS_REQUIRE(row_begin >= 0 && nrows >= 0 && row_begin + nrows <= ctx->rows, "row window outside the slab");
    if (nrows == 0) return FS_OK;
    FS_REQUIRE(!ctx->capturing, "upload during graph capture");
    const size_t bytes = (size_t)ctx->X * nrows * C * esize;
    int rc = ensure_stage(ctx, bytes);
    if (rc) return rc;
    FS_HIP(hipMemcpyAsync(ctx->d_stage, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    const int Q = nrows * C;
    dim3 grid((ctx->X + 63) / 64, (Q + 63) / 64);
    if (esize == 1) hipLaunchKernelGGL(k_to_device<uint8_t>, grid, dim3(256), 0, ctx->stream, (const uint8_t *)ctx->d_stage, (uint8_t *)dev, ctx->X, Q, pitch, row_begin * C);
    else if (esize == 4) hipLaunchKernelGGL(k_to_device<float>, grid, dim3(256), 0, ctx->stream, (const float *)ctx->d_stage, (float *)dev, ctx->X, Q, pitch, row_begin * C);
    else hipLaunchKernelGGL(k_to_device<double>, grid, dim3(256), 0, ctx->stream, (const double *)ctx->d_stage, (double *)dev, ctx->X, Q, pitch, row_begin * C);
    FS_HIP(hipGetLastError());
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

// Slice rows [g0, g0 + n) of a GLOBAL (X, Y, C) host array into a contiguous (X, n, C) buffer.
static std::vector<uint8_t> slice_rows(const void *src, int X, int Y, int C, size_t esize, int g0, int n)
{
    std::vector<uint8_t> out((size_t)X * n * C * esize);
    const uint8_t *s = (const uint8_t *)src;
    const size_t rowb = (size_t)C * esize;
    for (int i = 0; i < X; ++i)
        memcpy(out.data() + (size_t)i * n * rowb, s + ((size_t)i * Y + g0) * rowb, (size_t)n * rowb);
    return out;
}

static int upload_global(fs_ctx *ctx, void *dev, int C, size_t esize, const void *host_global, int pitch)
{
    // local rows that map inside the global domain
    const int g_lo = std::max(0, ctx->y0 - ctx->halo), g_hi = std::min(ctx->Y, ctx->y0 + ctx->nyl + ctx->halo);
    const int r0 = g_lo - (ctx->y0 - ctx->halo), n = g_hi - g_lo;
    if (g_lo == 0 && n == ctx->Y) return upload_window(ctx, dev, C, esize, host_global, r0, n, pitch);
    std::vector<uint8_t> tmp = slice_rows(host_global, ctx->X, ctx->Y, C, esize, g_lo, n);
    return upload_window(ctx, dev, C, esize, tmp.data(), r0, n, pitch);
}

int fs_upload_mask(fs_ctx *ctx, const uint8_t *mask_xy)
{
    FS_REQUIRE(ctx && mask_xy, "null argument");
    FS_HIP(hipSetDevice(ctx->device));
    FS_HIP(hipMemsetAsync(ctx->d_mask, 1, (size_t)ctx->rows * ctx->Pm, ctx->stream));
    int rc = upload_global(ctx, ctx->d_mask, 1, 1, mask_xy, ctx->Pm);
    if (rc) return rc;
    rc = build_bc_ops(ctx, mask_xy);
    if (rc) return rc;
    if (!ctx->d_bcmap) FS_HIP(hipMalloc(&ctx->d_bcmap, (size_t)ctx->rows * ctx->Pm));
    if (!ctx->d_lazyflags) FS_HIP(hipMalloc(&ctx->d_lazyflags, (size_t)std::max(ctx->nwx, 1) * ctx->rows));
    FS_HIP(hipMemsetAsync(ctx->d_bcmap, 0, (size_t)ctx->rows * ctx->Pm, ctx->stream));
    FS_HIP(hipMemsetAsync(ctx->d_lazyflags, 63, (size_t)std::max(ctx->nwx, 1) * ctx->rows, ctx->stream));
    rc = upload_global(ctx, ctx->d_bcmap, 1, 1, ctx->h_bcmap.data(), ctx->Pm);
    if (rc == FS_OK) {       // the marching red-black pass reads ONE byte per cell: recipe bits 0-6 + "not fluid" (rows outside the domain: wall)
        if (!ctx->d_rbcode) FS_HIP(hipMalloc(&ctx->d_rbcode, (size_t)ctx->rows * ctx->Pm));
        FS_HIP(hipMemsetAsync(ctx->d_rbcode, 0x80, (size_t)ctx->rows * ctx->Pm, ctx->stream));
        std::vector<uint8_t> rb((size_t)ctx->X * ctx->Y);
        for (size_t q = 0; q < rb.size(); ++q) rb[q] = (uint8_t)((ctx->h_bcmap[q] & 0x7f) | (mask_xy[q] != 0 ? 0x80 : 0));
        rc = upload_global(ctx, ctx->d_rbcode, 1, 1, rb.data(), ctx->Pm);
        if (rc == FS_OK) {   // ... and the marching Jacobi passes recipe bits + "wall" (the sweep computes every cell that is not a wall)
            if (!ctx->d_jcode) FS_HIP(hipMalloc(&ctx->d_jcode, (size_t)ctx->rows * ctx->Pm));
            FS_HIP(hipMemsetAsync(ctx->d_jcode, 0x80, (size_t)ctx->rows * ctx->Pm, ctx->stream));
            for (size_t q = 0; q < rb.size(); ++q) rb[q] = (uint8_t)((ctx->h_bcmap[q] & 0x7f) | (mask_xy[q] == 1 ? 0x80 : 0));
            rc = upload_global(ctx, ctx->d_jcode, 1, 1, rb.data(), ctx->Pm);
        }
    }
    // activity of the scene per (wave column, row) for the compact launches: a cell is "deep wall" when it is a wall cell that no
    // boundary kernel writes - workgroups made of such cells only have nothing to do in any kernel
    // captured graphs and recorded tapes hold the device pointers of the lists (and the launch geometry of the old scene): a new mask
    // invalidates them - a later fs_graph_launch / fs_tape_replay of such an id is an error, not a read through a dangling pointer
    for (auto &gexec : ctx->graphs) if (gexec) { hipGraphExecDestroy(gexec); gexec = nullptr; }
    for (auto &tp : ctx->tapes) if (tp) { delete tp; tp = nullptr; }
    tile_lists_free(ctx);
    ctx->h_act4.clear(); ctx->h_act2.clear(); ctx->h_act2w.clear();
    if (ctx->X % 2 == 0 && ctx->tile_list_mask && ctx->rows <= 0xffff) {
        // indexed by LOCAL row (a slab: its ghost rows included; rows outside the domain are deep wall)
        const int X = ctx->X, Y = ctx->Y, R = ctx->rows, g0 = ctx->y0 - ctx->halo;
        struct Geo { std::vector<uint8_t> *act; int w, halo; } geos[3] = {{&ctx->h_act4, 248, 4}, {&ctx->h_act2, 120, 4}, {&ctx->h_act2w, 124, 2}};
        for (const Geo &ge : geos) {
            const int w = ge.w, n = (X + w - 1) / w;
            ge.act->assign((size_t)n * R, 0);
            for (int i = 0; i < X; ++i) {
                const uint8_t *m = mask_xy + (size_t)i * Y, *b = ctx->h_bcmap.data() + (size_t)i * Y;
                uint8_t *a = ge.act->data() + (size_t)(i / w) * R;
                // the neighbouring wave column whose halo lanes cover column i, if any
                const int r = i % w;
                uint8_t *h = r < ge.halo && i / w > 0 ? a - R : (r >= w - ge.halo && i / w + 1 < n ? a + R : nullptr);
                for (int lr = 0; lr < R; ++lr) {
                    const int j = g0 + lr;
                    if (j < 0 || j >= Y) { a[lr] |= 2; if (h) h[lr] |= 2; continue; }
                    const uint8_t nf = m[j] != 0 ? 2 : 0;
                    a[lr] |= (uint8_t)((m[j] != 1) | (b[j] != 0)) | nf;
                    if (h) h[lr] |= nf;
                }
            }
        }
    }
    std::vector<uint8_t>().swap(ctx->h_bcmap);
    if (rc) return rc;
    if (ctx->X % 4 == 0) {      // per-tile flags of the lazy pressure BC
        hipLaunchKernelGGL(k_lazy_flags, dim3((ctx->nwx * ctx->rows + 3) / 4), dim3(256), 0, ctx->stream, ctx->grid(), ctx->nwx, ctx->d_bcmap, ctx->d_lazyflags);
        FS_HIP(hipGetLastError());
        // the rows the two-sweep kernel hands to its general path: list + count (read back once per mask)
        const size_t cap = (size_t)ctx->nwx * ctx->rows;
        if (!ctx->d_pairlist) FS_HIP(hipMalloc(&ctx->d_pairlist, (2 * cap + 2) * sizeof(uint32_t)));
        unsigned *d_count = (unsigned *)(ctx->d_pairlist + 2 * cap);
        FS_HIP(hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(k_pair_list, dim3((ctx->nwx * ctx->rows + 3) / 4), dim3(256), 0, ctx->stream, ctx->grid(), ctx->nwx, (uint8_t *)ctx->d_lazyflags,
                           ctx->d_pairlist, ctx->d_pairlist + cap, d_count);
        FS_HIP(hipGetLastError());
        unsigned n[2] = {0, 0};
        FS_HIP(hipMemcpyAsync(n, d_count, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(hipStreamSynchronize(ctx->stream));
        ctx->n_pairlist[0] = (int)n[0]; ctx->n_pairlist[1] = (int)n[1];
    }
    ctx->mask_set = true;
    return FS_OK;
}

static int upload_const(fs_ctx *ctx, void **slot, int C, const void *host)
{
    FS_REQUIRE(ctx && host, "null argument");
    FS_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->rows * C * ctx->P * ctx->esize;
    if (!*slot) FS_HIP(hipMalloc(slot, bytes));
    FS_HIP(hipMemsetAsync(*slot, 0, bytes, ctx->stream));
    return upload_global(ctx, *slot, C, ctx->esize, host, ctx->P);
}

int fs_upload_bc_const(fs_ctx *ctx, const void *bc_xy2) { return upload_const(ctx, ctx ? &ctx->d_bc_const : nullptr, 2, bc_xy2); }
int fs_upload_bc_dye(fs_ctx *ctx, const void *bc_xy3) { return upload_const(ctx, ctx ? &ctx->d_bc_dye : nullptr, 3, bc_xy3); }

int fs_bc_radius(const fs_ctx *ctx, int *velocity_rows, int *pressure_rows)
{
    FS_REQUIRE(ctx && velocity_rows && pressure_rows, "null argument");
    *velocity_rows = ctx->bc_radius_vel;
    *pressure_rows = ctx->bc_radius_prs;
    // a hazard chain that leaves this slab's ghost rows cannot be evaluated here at all: report "deeper than the halo", so that the
    // callers' collective maximum makes EVERY rank refuse the decomposition (not just this one, with the others waiting in RCCL)
    if (ctx->bc_incomplete) *velocity_rows = std::max(*velocity_rows, ctx->halo + 1);
    return FS_OK;
}

// ---- fields ---------------------------------------------------------------------------------------
int fs_field_alloc(fs_ctx *ctx, int nchan, fs_field **out)
{
    FS_REQUIRE(ctx && out, "null argument");
    FS_REQUIRE(nchan >= 1 && nchan <= 3, "nchan must be 1, 2 or 3");
    FS_HIP(hipSetDevice(ctx->device));
    fs_field *f = new fs_field();
    f->ctx = ctx; f->C = nchan;
    f->bytes = (size_t)ctx->rows * nchan * ctx->P * ctx->esize;
    hipError_t e = hipMalloc(&f->d, f->bytes);
    if (e == hipSuccess) e = hipMemsetAsync(f->d, 0, f->bytes, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&f->hot, 4 * sizeof(unsigned));       // [0]: the flag; [1], [2]: raised by the op list of a k_velocity_bc_limit launch of parity 0 / 1 (fs_march.h)
    if (e == hipSuccess) e = hipMemsetAsync(f->hot, 0, 4 * sizeof(unsigned), ctx->stream);
    if (e != hipSuccess) { if (f->d) hipFree(f->d); if (f->hot) hipFree(f->hot); delete f; return hip_fail(e, "hipMalloc(field)", __FILE__, __LINE__); }
    ctx->fields.insert(f);
    *out = f;
    return FS_OK;
}

static void field_release(fs_field *f)
{
    if (f->d) hipFree(f->d);
    if (f->hot) hipFree(f->hot);
    delete f;
}

int fs_field_free(fs_field *f)
{
    if (!f) return FS_OK;
    fs_ctx *ctx = f->ctx;
    hipSetDevice(ctx->device);
    ctx->fields.erase(f);
    // inside a hipGraph capture neither the synchronisation nor hipFree is legal (either invalidates the capture): a field dropped by
    // the host language's garbage collector at that moment is released when the capture ends
    if (ctx->capturing) { ctx->deferred_free.push_back(f); return FS_OK; }
    hipStreamSynchronize(ctx->stream);
    field_release(f);
    return FS_OK;
}

int fs_field_nchan(const fs_field *f) { return f ? f->C : FS_ERR_ARG; }

int fs_field_fill(fs_field *f, double value)
{
    FS_REQUIRE(f, "field is null");
    fs_ctx *ctx = f->ctx;
    const size_t n = f->bytes / ctx->esize;
    const unsigned hot = 2.0 * value * value > 0.999 * (double)FS_HOT_SQ ? 1u : 0u;      // every channel takes `value` (the margin: x * x + y * y is evaluated in the field type on the device)
    FS_DISPATCH(ctx, {
        return launch(ctx, "fill", [=] {
            hipLaunchKernelGGL(k_fill<T>, dim3(2048), dim3(256), 0, ctx->stream, (T *)f->d, n, (T)value);
            hipLaunchKernelGGL(k_fill<unsigned>, dim3(1), dim3(64), 0, ctx->stream, f->hot, (size_t)4, hot);
        });
    })
}

int fs_field_upload(fs_field *f, const void *host_xrc, int row_begin, int nrows)
{
    FS_REQUIRE(f, "field is null");
    fs_ctx *ctx = f->ctx;
    FS_HIP(hipSetDevice(ctx->device));
    int rc = upload_window(ctx, f->d, f->C, ctx->esize, host_xrc, row_begin, nrows, ctx->P);
    if (rc || f->C != 2 || nrows == 0) return rc;
    FS_DISPATCH(ctx, {      // what came in may exceed the speed the limit_field gate assumes: look at it (fs_device.h "hot" flag)
        hipLaunchKernelGGL(k_scan_hot<T>, cells_grid(ctx, row_begin, row_begin + nrows), dim3(256), 0, ctx->stream, ctx->grid(), row_begin, (const T *)f->d, f->hot);
    })
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_field_download(const fs_field *f, void *host_xrc, int row_begin, int nrows)
{
    FS_REQUIRE(f && host_xrc, "null argument");
    fs_ctx *ctx = f->ctx;
    FS_REQUIRE(row_begin >= 0 && nrows >= 0 && row_begin + nrows <= ctx->rows, "row window outside the slab");
    FS_REQUIRE(!ctx->capturing, "download during graph capture");
    if (nrows == 0) return FS_OK;
    FS_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->X * nrows * f->C * ctx->esize;
    int rc = ensure_stage(ctx, bytes);
    if (rc) return rc;
    const int Q = nrows * f->C;
    dim3 grid((ctx->X + 63) / 64, (Q + 63) / 64);
    FS_DISPATCH(ctx, {
        hipLaunchKernelGGL(k_to_host<T>, grid, dim3(256), 0, ctx->stream, (T *)ctx->d_stage, (const T *)f->d, ctx->X, Q, ctx->P, row_begin * f->C);
    })
    FS_HIP(hipGetLastError());
    FS_HIP(hipMemcpyAsync(host_xrc, ctx->d_stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

// Is the buffer's "may hold a speed above 9.95" flag up (any of its three words, fs_device.h)?  Synchronises the stream: for the host's decision
// between launch sequences (fs/fluid_simulator.py: a run that has gone hot takes limit_field as its own full-grid launch again - inside a
// boundary launch the pass is shared by a few dozen workgroups, 130 us against 47 at res 4096).
int fs_field_hot(const fs_field *f, int *hot)
{
    FS_REQUIRE(f && hot, "null argument");
    fs_ctx *ctx = f->ctx;
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "fs_field_hot during graph capture / tape recording");
    FS_HIP(hipSetDevice(ctx->device));
    unsigned h[4] = {0u, 0u, 0u, 0u};
    FS_HIP(hipMemcpyAsync(h, f->hot, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(hipStreamSynchronize(ctx->stream));
    *hot = (h[0] | h[1] | h[2] | h[3]) != 0u ? 1 : 0;
    return FS_OK;
}

static __global__ void k_hot_fold(unsigned *dst, const unsigned *src)
{
    if (threadIdx.x == 0) { dst[0] = (src[0] | src[1] | src[2] | src[3]) != 0u ? 1u : 0u; dst[1] = 0u; dst[2] = 0u; dst[3] = 0u; }
}

int fs_field_copy(fs_field *dst, const fs_field *src)
{
    FS_REQUIRE(dst && src && dst->ctx == src->ctx && dst->C == src->C, "copy needs two fields of one context and shape");
    FS_HIP(hipMemcpyAsync(dst->d, src->d, src->bytes, hipMemcpyDeviceToDevice, dst->ctx->stream));
    hipLaunchKernelGGL(k_hot_fold, dim3(1), dim3(64), 0, dst->ctx->stream, dst->hot, (const unsigned *)src->hot);      // (one word: the copy starts a new parity sequence)
    FS_HIP(hipGetLastError());
    return FS_OK;
}

int fs_field_devptr(const fs_field *f, void **ptr, size_t *bytes)
{
    FS_REQUIRE(f, "field is null");
    if (ptr) *ptr = f->d;
    if (bytes) *bytes = f->bytes;
    return FS_OK;
}

// ---- boundary-condition kernels ----------------------------------------------------------------------
static int bc_guard(fs_ctx *ctx)
{
    if (ctx->bc_incomplete) {
        set_error("boundary-condition hazard chain extends beyond this slab's ghost rows (thin walls at a slab cut); increase halo");
        return FS_ERR_UNSUPPORTED;
    }
    return FS_OK;
}

int fs_velocity_bc(fs_ctx *ctx, fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v, 2);
    FS_ROWS();
    if (!ctx->d_bc_const) { set_error("bc_const not uploaded"); return FS_ERR_STATE; }
    int rc = bc_guard(ctx); if (rc) return rc;
    if (ctx->ops_vel.lanes() == 0) return FS_OK;
    FS_DISPATCH(ctx, {
        return launch(ctx, "velocity_bc", [=] {
            hipLaunchKernelGGL(k_velocity_bc<T>, dim3((ctx->ops_vel.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_vel.view(), row_begin, row_end, (T *)v->d, (const T *)ctx->d_bc_const, v->hot);
        });
    })
}

// limit_field(v, limit) of the step before + the velocity boundary kernel of this step in one launch (fs_march.h k_velocity_bc_limit):
// the same result as fs_limit_field over [limit_begin, limit_end) followed by fs_velocity_bc over [row_begin, row_end)
int fs_velocity_bc_limit_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    const int wgs = (ctx->ops_vel.lanes() + 255) / 256;
    // every workgroup resident: the grid barrier of the rare path
    *ok = ctx->mask_set && ctx->use_march && ctx->limit_gate && ctx->d_sync && ctx->d_bc_const && wgs >= 1 && wgs <= 1024 ? 1 : 0;
    return FS_OK;
}

int fs_velocity_bc_limit(fs_ctx *ctx, double limit, fs_field *v, int parity, int limit_begin, int limit_end, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(parity == 0 || parity == 1, "parity must be 0 or 1");
    FS_FIELD(v, 2);
    FS_ROWS();
    FS_REQUIRE(limit_begin >= 0 && limit_begin <= limit_end && limit_end <= ctx->rows, "bad row range of the limit pass");
    if (!ctx->d_bc_const) { set_error("bc_const not uploaded"); return FS_ERR_STATE; }
    int rc = bc_guard(ctx); if (rc) return rc;
    int ok = 0;
    fs_velocity_bc_limit_ok(ctx, &ok);
    if (!ok || !((float)limit * (float)limit > FS_HOT_GATE_SQ)) { set_error("fs_velocity_bc_limit is not available for this context / limit (fs_velocity_bc_limit_ok)"); return FS_ERR_UNSUPPORTED; }
    FS_DISPATCH(ctx, {
        return launch(ctx, "velocity_bc", [=] {
            // (at least one workgroup per row of the limit pass, up to 64 (the barrier costs ~40 ns per workgroup): with the flag up - a run that has once exceeded a speed of 8 keeps it
            //  up - the pass is shared by the launch's workgroups; the extra ones find no op and cost nothing while the flag is down)
            hipLaunchKernelGGL(k_velocity_bc_limit<T>, dim3(std::max((ctx->ops_vel.lanes() + 255) / 256, std::min(64, limit_end - limit_begin))), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_vel.view(), row_begin, row_end, limit_begin, limit_end, (T)limit, (T *)v->d, (const T *)ctx->d_bc_const, v->hot, ctx->d_sync, parity);
        });
    })
}

int fs_pressure_bc(fs_ctx *ctx, fs_field *p, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(p, 1);
    FS_ROWS();
    int rc = bc_guard(ctx); if (rc) return rc;
    if (ctx->ops_prs.lanes() == 0) return FS_OK;
    FS_DISPATCH(ctx, {
        return launch(ctx, "pressure_bc", [=] {
            hipLaunchKernelGGL(k_pressure_bc<T>, dim3((ctx->ops_prs.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_prs.view(), row_begin, row_end, (T *)p->d);
        });
    })
}

int fs_dye_bc(fs_ctx *ctx, fs_field *dye, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(dye, 3);
    FS_ROWS();
    if (!ctx->d_bc_dye) { set_error("bc_dye not uploaded"); return FS_ERR_STATE; }
    if (ctx->ops_dye.lanes() == 0) return FS_OK;
    FS_DISPATCH(ctx, {
        return launch(ctx, "dye_bc", [=] {
            hipLaunchKernelGGL(k_dye_bc<T>, dim3((ctx->ops_dye.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_dye.view(), row_begin, row_end, (T *)dye->d, (const T *)ctx->d_bc_dye);
        });
    })
}

int fs_dye_bc_limit_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    const int wgs = (ctx->ops_dye.lanes() + 255) / 256;
    *ok = ctx->mask_set && ctx->use_march && ctx->limit_gate && ctx->d_sync && ctx->d_bc_dye && wgs >= 1 && wgs <= 1024 ? 1 : 0;
    return FS_OK;
}

int fs_dye_bc_limit(fs_ctx *ctx, double limit, fs_field *v, fs_field *dye, int limit_begin, int limit_end, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v, 2); FS_FIELD(dye, 3);
    FS_ROWS();
    FS_REQUIRE(limit_begin >= 0 && limit_begin <= limit_end && limit_end <= ctx->rows, "bad row range of the limit pass");
    int ok = 0;
    fs_dye_bc_limit_ok(ctx, &ok);
    if (!ok || !((float)limit * (float)limit > FS_HOT_GATE_SQ)) { set_error("fs_dye_bc_limit is not available for this context / limit (fs_dye_bc_limit_ok)"); return FS_ERR_UNSUPPORTED; }
    FS_DISPATCH(ctx, {
        return launch(ctx, "dye_bc", [=] {
            hipLaunchKernelGGL(k_dye_bc_limit<T>, dim3(std::max((ctx->ops_dye.lanes() + 255) / 256, std::min(64, limit_end - limit_begin))), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_dye.view(), row_begin, row_end, limit_begin, limit_end, (T)limit, (T *)v->d, v->hot, ctx->d_sync,
                               (T *)dye->d, (const T *)ctx->d_bc_dye);
        });
    })
}

// ---- transport -----------------------------------------------------------------------------------------
#define FS_LAUNCH_CELLS(name, kern, ...)                                                                   \
    return launch(ctx, name, [=] {                                                                         \
        hipLaunchKernelGGL(kern, cells_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, __VA_ARGS__); \
    });

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
            const int rt = ctx->mac_rt ? ctx->mac_rt : (sizeof(T) == 4 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 20) ? 4 : 2);
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_NONADV, 3);
            return launch(ctx, scheme == FS_UPWIND ? "mac_update_upwind" : "mac_update_kk", [=] {
#define FS_K2MN(SS, RR, PP) hipLaunchKernelGGL((k_mac_update_n<SS, 2, RR, PP, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
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
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_NONADV, 3);
            const int clear3 = whole_grid(ctx, row_begin, row_end);      // (fs_device.h "hot" word [3])
            return launch(ctx, "cip_nonadv", [=] {
#define FS_K2N4(DM) hipLaunchKernelGGL((k_cip_nonadv_n<2, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot, clear3)
#define FS_K2N2(DM) hipLaunchKernelGGL((k_cip_nonadv_n<2, 2, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)fn->d, (const T *)fc->d, (const T *)pc->d, fn->hot, clear3)
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
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_NONADV, 3);
            return launch(ctx, "cip_nonadv_dye", [=] {
#define FS_K12N(DM) hipLaunchKernelGGL((k_cip_nonadv_dye_n<2, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)dn->d, (const T *)dc->d)
#define FS_K12N2(DM) hipLaunchKernelGGL((k_cip_nonadv_dye_n<2, 2, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)dn->d, (const T *)dc->d)
                if (small) FS_DMA(dm_all(ctx, k), FS_K12N2); else FS_DMA(dm_all(ctx, k), FS_K12N);
            });
        }
        FS_LAUNCH_CELLS("cip_nonadv_dye", (k_cip_nonadv_dye<T>), ctx->grid(), k, row_begin, (T *)dn->d, (const T *)dc->d)
    })
}

#define FS_K3Q(CC, NC, PP) hipLaunchKernelGGL((k_cip_nonadv_grad_quad<CC, NC, PP, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
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

#define FS_K4Q(CC, NC, SELF, PP) hipLaunchKernelGGL((k_cip_advect_quad<CC, NC, SELF, PP, false, T>), qgrid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
#define FS_K4D(PP) hipLaunchKernelGGL((k_cip_advect_dye<PP, false, T>), qgrid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
        (T *)fn->d, (T *)fxn->d, (T *)fyn->d, (const T *)fc->d, (const T *)fxc->d, (const T *)fyc->d, (const T *)v->d, fn->hot)
#define FS_K4N(CC, PP) hipLaunchKernelGGL((k_cip_advect<CC, PP, T>), cells_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, \
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
#define FS_K4DC(DM) hipLaunchKernelGGL((k_cip_advect_dye<DM, true, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
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
            hipLaunchKernelGGL(k_clamp_inflow<T>, dim3((ctx->ops_dye.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
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
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, small ? 2 : 4, 1, XCD_VORT, 3);
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0, weight);
        const int dm = dm_dx(ctx, k);
        T *w = vort ? (T *)vort->d : nullptr; T *wa = vort_abs ? (T *)vort_abs->d : nullptr;
        const int clear3 = whole_grid(ctx, row_begin, row_end);      // (fs_device.h "hot" word [3])
#define FS_VORTN(DM, ST) hipLaunchKernelGGL((k_vort_n<2, 4, DM, ST, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)vn->d, (const T *)vc->d, w, wa, vn->hot, clear3)
#define FS_VORTN_S(DM) FS_VORTN(DM, true)
#define FS_VORTN_N(DM) FS_VORTN(DM, false)
#define FS_VORTN_2(DM) hipLaunchKernelGGL((k_vort_n<2, 2, DM, false, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)vn->d, (const T *)vc->d, w, wa, vn->hot, clear3)
        return launch(ctx, "vort_confine", [=] { if (vort) FS_DMX(dm, FS_VORTN_S); else if (small) FS_DMX(dm, FS_VORTN_2); else FS_DMX(dm, FS_VORTN_N); });
    })
}

// ---- pressure ------------------------------------------------------------------------------------------------
int fs_jacobi_sweep(fs_ctx *ctx, double dt, double dx, fs_field *pn, const fs_field *pc, const fs_field *vc, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(vc, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_ROWS();
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, dt, dx, 1.0);
        if (ctx->use_march) return launch_jacobi<false, T>(ctx, "jacobi_sweep", k, row_begin, row_end, (T *)pn->d, (const T *)pc->d, (const T *)vc->d);
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
    *ok = ctx->mask_set && ctx->lazy_ok && ctx->use_march && ctx->use_lazy ? 1 : 0;
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
            hipLaunchKernelGGL((k_jacobi_lazy<T>), og.grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end,
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
    const int rt = (mode & 2) ? std::min(ctx->pair_rt, 2) : ctx->pair_rt;      // (the third tile path at 3 rows: 97 VGPRs, one wave per SIMD less)
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
            hipLaunchKernelGGL((k_rbsor<false, T>), rb_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k,
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
#define FS_RBN4(DM) hipLaunchKernelGGL((k_rbsor_iter_n<2, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (T *)pn->d, (const T *)pc->d, (const T *)vc->d)
        return launch(ctx, "rbsor_iteration", [=] { FS_DMC(dm_const(ctx, k), FS_RBN4); });
    })
}

// diagnostic: how many of ~2^28 dividends (see k_verify_f64div) does the f64-multiply division of f32 values get wrong for this divisor?
int fs_selftest_f64div(fs_ctx *ctx, double divisor, int *mismatches)
{
    FS_REQUIRE(ctx && mismatches, "null argument");
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "self test during graph capture / tape recording");
    const float d = (float)divisor;
    FS_REQUIRE(d != 0.0f && d == d, "divisor must be a non-zero number");
    FS_HIP(hipSetDevice(ctx->device));
    unsigned *flag = nullptr, h = ~0u;
    FS_HIP(hipMalloc(&flag, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_verify_f64div, dim3(1u << 15, 10 + 2), dim3(256), 0, ctx->stream, d, 1.0 / (double)d, tie_free((double)d) ? 0 : 1, flag);   // the form the library uses for this divisor
        e = hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(flag);
    if (e != hipSuccess) return hip_fail(e, "fs_selftest_f64div", __FILE__, __LINE__);
    *mismatches = (int)std::min<unsigned>(h, 0x7fffffffu);
    return FS_OK;
}

// ---- what THIS box streams at (measurement hygiene: the pool's boxes differ by several per cent) ---------------------
// float4 read of one buffer and float4 copy between two buffers of `bytes` each (step-sized: beyond the 256 MiB Infinity Cache), timed with
// HIP events on the context's stream for about budget_ms each.  bench.py prints both next to every roofline fraction.
__global__ __launch_bounds__(256) static void k_box_read(const float4 *__restrict__ a, float *sink, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t k = i + u * stride; v[u] = k < n ? a[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(256) static void k_box_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    // a block copies 4 consecutive segments of 256 float4: every load / store instruction of a wave is one coalesced 1 KiB segment
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        const float4 v0 = a[i], v1 = a[i + 256], v2 = a[i + 512], v3 = a[i + 768];
        b[i] = v0; b[i + 256] = v1; b[i + 512] = v2; b[i + 768] = v3;
    }
}

// ... and what it ISSUES at: 8 independent chains of dependent f32 multiplies and adds per lane (no memory), 4 waves per SIMD - the shape of the
// issue-bound part of K3+K4: the rate that kernel is priced against (bench.py roofline.valu_issue).  (The pool's boxes measured alike here, 0.528-0.537 G
// per second and SIMD, also where the real kernels differed by 5-9 %: DESIGN.md section 8.)
__global__ __launch_bounds__(256) static void k_box_valu(float *sink, float a, float b, int iters)
{
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = a + (float)(threadIdx.x + u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = x[u] * a;
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = x[u] + b;
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u];
    if (s == 1.2345f) sink[0] = s;
}

int fs_box_valu_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd)
{
    FS_REQUIRE(ctx && ginstr_per_simd, "null argument");
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "fs_box_valu_rate during graph capture / tape recording");
    FS_REQUIRE(budget_ms > 0.0, "need a positive time budget");
    FS_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    FS_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount, iters = 2000;
    float *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&sink, sizeof(float));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    int reps = 1;
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
        (void)hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_box_valu, dim3(cus * 4), dim3(256), 0, ctx->stream, sink, 1.0000001f, 1e-9f, iters);
        (void)hipEventRecord(e1, ctx->stream);
        e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) break;
        // per SIMD: 4 waves x iters x 16 wave-instructions
        *ginstr_per_simd = 4.0 * iters * 16.0 * reps / (ms * 1e-3) / 1e9;
        reps = std::max(1, std::min(1000, (int)(budget_ms / std::max((double)ms / reps, 1e-3))));
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (sink) hipFree(sink);
    if (e != hipSuccess) return hip_fail(e, "fs_box_valu_rate", __FILE__, __LINE__);
    return FS_OK;
}

// ... and both at once: a float4 copy with 176 f32 multiplies / adds per 16 bytes on the way - 5.5 lane-operations per byte moved, the instruction
// density of K3+K4 (131 M wave-instructions for 1.5 GB).  The pure stream and the pure ALU loop above measured alike on boxes whose real kernels
// differed by 5-9 %; this is the probe that loads the memory system and the SIMDs together.
__global__ __launch_bounds__(256) static void k_box_mixed(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n, float m, float c)
{
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 >= n) return;
    float4 v[4] = {a[i], a[i + 256], a[i + 512], a[i + 768]};
#pragma unroll
    for (int r = 0; r < 22; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u].x = v[u].x * m; v[u].y = v[u].y * m; v[u].z = v[u].z * m; v[u].w = v[u].w * m; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u].x = v[u].x + c; v[u].y = v[u].y + c; v[u].z = v[u].z + c; v[u].w = v[u].w + c; }
    }
    b[i] = v[0]; b[i + 256] = v[1]; b[i + 512] = v[2]; b[i + 768] = v[3];
}

int fs_box_mixed_rate(fs_ctx *ctx, size_t bytes, double budget_ms, double *GBps)
{
    FS_REQUIRE(ctx && GBps, "null argument");
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "fs_box_mixed_rate during graph capture / tape recording");
    FS_REQUIRE(bytes >= (1u << 20) && budget_ms > 0.0, "need at least 1 MiB and a positive time budget");
    FS_HIP(hipSetDevice(ctx->device));
    bytes = bytes / 4096 * 4096;
    const size_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    int reps = 1;
    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
        (void)hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_box_mixed, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n, 1.0000001f, 1e-9f);
        (void)hipEventRecord(e1, ctx->stream);
        e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) break;
        *GBps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
        reps = std::max(1, std::min(4000, (int)(budget_ms / std::max((double)ms / reps, 1e-3))));
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (b) hipFree(b);
    if (e != hipSuccess) return hip_fail(e, "fs_box_mixed_rate", __FILE__, __LINE__);
    return FS_OK;
}

int fs_box_rates(fs_ctx *ctx, size_t bytes, double budget_ms, double *read_GBps, double *copy_GBps)
{
    FS_REQUIRE(ctx && read_GBps && copy_GBps, "null argument");
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "fs_box_rates during graph capture / tape recording");
    FS_REQUIRE(bytes >= (1u << 20) && budget_ms > 0.0, "need at least 1 MiB and a positive time budget");
    FS_HIP(hipSetDevice(ctx->device));
    bytes = bytes / 4096 * 4096;
    const size_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    float *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e == hipSuccess) e = hipMalloc(&sink, sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto timed = [&](bool copy, double *out) {
        // one launch to learn the rate, then as many as fit the budget
        int reps = 1;
        for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {
            (void)hipEventRecord(e0, ctx->stream);
            for (int r = 0; r < reps; ++r) {
                if (copy) hipLaunchKernelGGL(k_box_copy, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n);
                else hipLaunchKernelGGL(k_box_read, dim3(2048), dim3(256), 0, ctx->stream, a, sink, n);
            }
            (void)hipEventRecord(e1, ctx->stream);
            e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) return;
            *out = (copy ? 2.0 : 1.0) * (double)bytes * reps / (ms * 1e-3) / 1e9;
            reps = std::max(1, std::min(1000, (int)(budget_ms / std::max((double)ms / reps, 1e-3))));
        }
    };
    if (e == hipSuccess) timed(false, read_GBps);
    if (e == hipSuccess) timed(true, copy_GBps);
    if (e == hipSuccess) e = hipGetLastError();
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (b) hipFree(b);
    if (sink) hipFree(sink);
    if (e != hipSuccess) return hip_fail(e, "fs_box_rates", __FILE__, __LINE__);
    return FS_OK;
}

// four lazily-bounded Jacobi sweeps in one pass (fs_jquad.h): pn[not wall] <- sweep^4(pc); both buffers hold raw sweep output
int fs_jacobi_quad_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = ctx->mask_set && ctx->jq_ok && ctx->use_march && ctx->use_lazy && ctx->dtype == 0 ? 1 : 0;
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
    const Grid gg = ctx->grid();
    // lanes of 2 cells (116 VGPRs = 4 waves per SIMD at 4 rows; quads: 182 = 2 waves, 44.9 against 34.3 us per pass at bc2 res 1600)
    const int rt = ctx->jquad_rt;
#define FS_JQ(RT, PATH) hipLaunchKernelGGL((k_jacobi_quad<2, RT, PATH, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pn->d, (const T *)pc->d, (const T *)src->d)
    // plain and boundary workgroups as two compact launches (as fs_rbsor_pair) - on large grids: a second launch costs ~5 us, which a
    // cache-resident grid does not earn back (bc2 res 1600: 18.1 + 21.3 against 34.6 us; bc5 res 4096: 81.4 + 49.8 against 137.5)
    if ((ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 23))) && rt == 4) {
        const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, true, 1, 4, ctx->split_wgw);
        const OvGrid ogb = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, true, 2, 4, ctx->split_wgw);
        if (og.g.tiles && ogb.g.tiles) {
            int rc = launch(ctx, "jacobi_quad_lazy", [=] { FS_JQ(4, 3); });
            if (rc) return rc;
            { const OvGrid og = ogb; return launch(ctx, "jacobi_quad_lazy_bnd", [=] { FS_JQ(4, 2); }); }
        }
    }
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, true, 0, 4);      // (per-wave plain hints in the list, as fs_rbsor_pair)
    return launch(ctx, "jacobi_quad_lazy", [=] {
        if (rt == 2) FS_JQ(2, 2); else if (rt == 6) FS_JQ(6, 2); else if (rt == 8) FS_JQ(8, 2); else FS_JQ(4, 2);
    });
}

// `sweeps` (4, 6 or 8) lazily-bounded Jacobi sweeps in one row-marching pass (fs_jmarch.h): pn[not wall] <- sweep^S(pc); the conditions of
// the four-sweep pass (fs_jacobi_quad_ok).  Strip height: 12 m - 2 S rows (the 12-step loop body then runs whole), the tallest that still
// gives every SIMD of the chip a few waves (FS_JM_L overrides).
int fs_jacobi_march(fs_ctx *ctx, fs_field *pn, const fs_field *pc, const fs_field *src, int sweeps, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(pn, 1); FS_FIELD(pc, 1); FS_FIELD(src, 2);
    FS_REQUIRE(pn != pc, "Jacobi needs two distinct pressure fields");
    FS_REQUIRE(sweeps == 4 || sweeps == 6 || sweeps == 8, "sweeps per pass: 4, 6 or 8");
    FS_ROWS();
    if (!(ctx->jq_ok && ctx->use_pairs && ctx->dtype == 0)) { set_error("this mask / precision does not admit the multi-sweep Jacobi passes (fs_jacobi_quad_ok)"); return FS_ERR_UNSUPPORTED; }
    using T = float;
    int L = ctx->jm_L;
    if (L <= 0) {
        const int ow = 64 - sweeps, cols = (ctx->X / 2 + ow - 1) / ow, rows = row_end - row_begin;
        L = 12 - 2 * sweeps > 0 ? 12 - 2 * sweeps : 24 - 2 * sweeps;
        while (L + 12 <= 254 && (long long)cols * (rows / (L + 12)) >= 2048) L += 12;       // at least ~2 waves per SIMD
    }
    const int pf = ctx->jm_pf;
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, L, 1, XCD_MARCH, sweeps == 4 ? 2 : (sweeps == 6 ? 5 : 6), true, 0);
    const JmArgs a{(const uint8_t *)ctx->d_jcode, pn->d, pc->d, src->d};
#define FS_JM_K(S, PF) hipLaunchKernelGGL((k_jacobi_march<2, S, PF, T>), og.grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end, L, a)
    return launch(ctx, "jacobi_march", [=] {
        if (sweeps == 4) { if (pf == 3) FS_JM_K(4, 3); else FS_JM_K(4, 1); }
        else if (sweeps == 6) { if (pf == 3) FS_JM_K(6, 3); else FS_JM_K(6, 1); }
        else FS_JM_K(8, 1);            // (8 sweeps: the code-word ring of 12 rows holds S + 2 + PF = 11)
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
        hipLaunchKernelGGL((k_jacobi_finish<2, 4, T>), og.grid, dim3(256), 0, ctx->stream, og.g, og.nbx, og.nby, row_begin, row_end,
                           (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn->d, (const T *)pc->d, (const T *)src->d);
    });
}

int fs_rbsor_pair_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    *ok = ctx->mask_set && ctx->rb_pair_ok && ctx->use_pairs && ctx->use_lazy ? 1 : 0;      // (f32 and, since round 4, f64)
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
#define FS_RBPD_K(RT, PAR, PATH, FULL) hipLaunchKernelGGL((k_rbsor_pair<2, RT, PAR, 0, PATH, FULL, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn_out->d, (const T *)pc->d, (const T *)pn->d, (const T *)vc->d)
#define FS_RBPD(RT, PATH, FULL) do { if (par0) FS_RBPD_K(RT, 1, PATH, FULL); else FS_RBPD_K(RT, 0, PATH, FULL); } while (0)
        if (!full && (ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 23))) && ctx->split_wgw == 1) {
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
    // lanes of 2 cells (8-byte loads: 126 - 156 VGPRs where quads need 223 - 248), RT = 4 (FS_RBPAIR_RT=6: 6) rows per tile.  The carrying
    // pass after an upload (full) is rare: one configuration.
    // (grids below 1 M cells: 2-row tiles - fewer waves than SIMDs there, the pass takes as long as ONE wave's chain of loads and stages:
    //  res 200 12.1 -> 9.2 us per launch, BASELINE configs[0] 53.3 -> 62.8 k steps/s; res 1600: 4 rows, 5602 against 5435 steps/s)
    const int rt = full ? 4 : (ctx->rbpair_rt ? ctx->rbpair_rt : (small_tiles(ctx) ? 2 : 4));
    if (!full && ctx->use_rbmarch) {
        // the row-marching form (fs_rbmarch.h): strips of L rows, one wave column each, plain and boundary rows in one kernel; the compact
        // list leaves out the strips of nothing but deep wall
        const int L = ctx->rbm_L, pf = ctx->rbm_pf;
        const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, L, 1, XCD_MARCH, 2, true, 0);
        const RbmArgs a{(const uint8_t *)ctx->d_rbcode, pc_out->d, pn_out->d, pc->d, pn->d, vc->d};
#define FS_RBM_K(PF, PAR, DM) hipLaunchKernelGGL((k_rbsor_march<2, PF, PAR, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, L, a)
#define FS_RBM_PAR(PF, DM) do { if (par0) FS_RBM_K(PF, 1, DM); else FS_RBM_K(PF, 0, DM); } while (0)
#define FS_RBM_DM(PF) do { if (dm & DM_F64) FS_RBM_PAR(PF, 4); else FS_RBM_PAR(PF, 0); } while (0)
        return launch(ctx, "rbsor_pair", [=] { if (pf == 1) FS_RBM_DM(1); else FS_RBM_DM(3); });
    }
#define FS_RBP_K(RT, PAR, DM, PATH, FULL) hipLaunchKernelGGL((k_rbsor_pair<2, RT, PAR, DM, PATH, FULL, T>), og.grid, dim3(og.threads), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, \
                               (const uint8_t *)ctx->d_bcmap, (T *)pc_out->d, (T *)pn_out->d, (const T *)pc->d, (const T *)pn->d, (const T *)vc->d)
#define FS_RBP_PAR(RT, DM, PATH, FULL) do { if (par0) FS_RBP_K(RT, 1, DM, PATH, FULL); else FS_RBP_K(RT, 0, DM, PATH, FULL); } while (0)
#define FS_RBP_DM(RT, PATH) do { if (dm & DM_F64) FS_RBP_PAR(RT, 4, PATH, false); else FS_RBP_PAR(RT, 0, PATH, false); } while (0)
    // Compact launch in two parts where the lists exist (single GPU, whole grid): the workgroups that see nothing but fluid within reach run
    // the plain path as its own kernel (PATH 3: no mask loads, 126 VGPRs = 4 waves per SIMD), the others the kernel with both paths.
    if (!full && (ctx->rbpair_split == 2 || (ctx->rbpair_split == 1 && (size_t)ctx->X * ctx->Y >= ((size_t)1 << 23))) && rt == 4) {
        // the plain part on tiles of 8 rows (round 4: 125 VGPRs, still 4 waves per SIMD, since the DPP shifts lost their init moves - 2 rows
        // of window per output row instead of 3), the boundary part on tiles of 4 rows that lie in no plain 8-row tile; one wave per workgroup
        const int prt = ctx->split_wgw == 1 && ctx->rbpair_plain_rt == 8 ? 8 : rt;
        const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, prt, 1, XCD_RBSOR, 2, true, 1, 4, ctx->split_wgw);
        const OvGrid ogb = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, true, 2, 4, ctx->split_wgw, prt);
        if (og.g.tiles && ogb.g.tiles) {
            int rc = launch(ctx, "rbsor_pair", [=] { if (prt == 8) FS_RBP_DM(8, 3); else FS_RBP_DM(4, 3); });      // (12-row tiles: 151 VGPRs = 3 waves, 188 against 177 us)
            if (rc) return rc;
            { const OvGrid og = ogb; return launch(ctx, "rbsor_pair_bnd", [=] { FS_RBP_DM(4, 2); }); }
        }
    }
    // (one launch: the list's entries carry a per-wave "plain" hint - a wave that sees nothing but fluid within 4 rows skips its mask loads)
    const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, rt, 1, XCD_RBSOR, 2, !full, 0, 4);
    return launch(ctx, "rbsor_pair", [=] {
        if (full) FS_RBP_PAR(4, 0, 2, true);
        else if (rt == 6) FS_RBP_DM(6, 2);
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
            hipLaunchKernelGGL((k_rbsor<true, T>), rb_grid(ctx, row_begin, row_end), dim3(256), 0, ctx->stream, ctx->grid(), k,
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
        if (ctx->use_pairs && !getenv("FS_SRC_CELLS")) {
            const OvGrid og = ov_grid_lanes(ctx, row_begin, row_end, 4, 1, XCD_JACOBI, 3);      // (deep-wall workgroups skipped: nobody reads the source there)
#define FS_PSN(DM) hipLaunchKernelGGL((k_poisson_source_n<2, 4, DM, T>), og.grid, dim3(256), 0, ctx->stream, og.g, k, og.nbx, og.nby, row_begin, row_end, (T *)src->d, (const T *)vc->d)
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
            hipLaunchKernelGGL((k_residual<T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, row_end,
                               (const T *)p->d, (const T *)vc->d, ctx->d_partial);
            hipLaunchKernelGGL((k_residual_final<double>), dim3(1), dim3(1024), 0, ctx->stream, (const double *)ctx->d_partial, (int)nblocks, ctx->d_acc);
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

// ---- pointwise -----------------------------------------------------------------------------------------------
int fs_limit_field(fs_ctx *ctx, double limit, fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v, 2);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        if (ctx->use_march) {
            // gated by the buffer's "hot" flag (fs_device.h): while no writer has stored a speed above 9.95 the pass has nothing to do
            const int gated = (T)limit * (T)limit > (T)FS_HOT_GATE_SQ && ctx->limit_gate ? 1 : 0;
            const int lanes = std::min(row_end - row_begin, 256);
            return launch(ctx, "limit_field", [=] {
                hipLaunchKernelGGL((k_limit_quad<T>), dim3((ctx->X / 4 + 255) / 256, lanes), dim3(256), 0, ctx->stream,
                                   ctx->grid(), row_begin, row_end, (T)limit, (T *)v->d, v->hot, gated);
            });
        }
        FS_LAUNCH_CELLS("limit_field", (k_limit<T>), ctx->grid(), row_begin, (T)limit, (T *)v->d)
    })
}

int fs_clamp_field(fs_ctx *ctx, double low, double high, fs_field *f, int row_begin, int row_end)
{
    FS_REQUIRE(ctx && f, "null argument");
    FS_REQUIRE(f->ctx == ctx, "field from another context");
    FS_ROWS();
    const int C = f->C;
    FS_DISPATCH(ctx, {
        if (C == 1) { FS_LAUNCH_CELLS("clamp_field_c1", (k_clamp<1, T>), ctx->grid(), row_begin, (T)low, (T)high, (T *)f->d) }
        else if (C == 2) { FS_LAUNCH_CELLS("clamp_field_c2", (k_clamp<2, T>), ctx->grid(), row_begin, (T)low, (T)high, (T *)f->d) }
        else { FS_LAUNCH_CELLS("clamp_field", (k_clamp<3, T>), ctx->grid(), row_begin, (T)low, (T)high, (T *)f->d) }
    })
}

// ---- visualisation (GUI side of the reference; device kernels so that a frame costs one pass + one download) ------------
static int visualize(fs_ctx *ctx, int mode, double dx, fs_field *rgb, const fs_field *a, const fs_field *b, int row_begin, int row_end)
{
    static const char *names[4] = {"vis_norm", "vis_pressure", "vis_vorticity", "vis_dye"};
    FS_DISPATCH(ctx, {
        auto k = make_konst<T>(ctx, 1.0, dx, 1.0);
        const T *pa = (const T *)a->d, *pb = b ? (const T *)b->d : nullptr;
        return launch(ctx, names[mode], [=] {
            const dim3 grid = cells_grid(ctx, row_begin, row_end);
            if (mode == 0) hipLaunchKernelGGL((k_visualize<0, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else if (mode == 1) hipLaunchKernelGGL((k_visualize<1, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else if (mode == 2) hipLaunchKernelGGL((k_visualize<2, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else hipLaunchKernelGGL((k_visualize<3, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
        });
    })
}

int fs_vis_norm(fs_ctx *ctx, fs_field *rgb, const fs_field *v, const fs_field *p, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(rgb, 3); FS_FIELD(v, 2); FS_FIELD(p, 1);
    FS_ROWS();
    return visualize(ctx, 0, 1.0, rgb, v, p, row_begin, row_end);
}

int fs_vis_pressure(fs_ctx *ctx, fs_field *rgb, const fs_field *p, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(rgb, 3); FS_FIELD(p, 1);
    FS_ROWS();
    return visualize(ctx, 1, 1.0, rgb, p, nullptr, row_begin, row_end);
}

int fs_vis_vorticity(fs_ctx *ctx, double dx, fs_field *rgb, const fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(rgb, 3); FS_FIELD(v, 2);
    FS_ROWS();
    return visualize(ctx, 2, dx, rgb, v, nullptr, row_begin, row_end);
}

int fs_vis_dye(fs_ctx *ctx, fs_field *rgb, const fs_field *dye, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(rgb, 3); FS_FIELD(dye, 3);
    FS_REQUIRE(rgb != dye, "rgb must not alias dye");
    FS_ROWS();
    return visualize(ctx, 3, 1.0, rgb, dye, nullptr, row_begin, row_end);
}

// ---- hipGraph capture ---------------------------------------------------------------------------------------
int fs_graph_begin(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(!ctx->capturing, "already capturing");
    FS_REQUIRE(!ctx->comm, "graph capture is single-GPU only");
    int rc = prof_drain(ctx); if (rc) return rc;
    FS_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return FS_OK;
}

int fs_graph_end(fs_ctx *ctx, int *graph_id)
{
    FS_REQUIRE(ctx && graph_id, "null argument");
    FS_REQUIRE(ctx->capturing, "not capturing");
    hipGraph_t g = nullptr;
    ctx->capturing = false;
    const hipError_t ec = hipStreamEndCapture(ctx->stream, &g);
    for (fs_field *f : ctx->deferred_free) field_release(f);          // fields dropped while the capture was open (fs_field_free)
    ctx->deferred_free.clear();
    if (ec != hipSuccess) return hip_fail(ec, "hipStreamEndCapture", __FILE__, __LINE__);
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
    ctx->graphs.push_back(ex);
    *graph_id = (int)ctx->graphs.size() - 1;
    return FS_OK;
}

int fs_graph_launch(fs_ctx *ctx, int graph_id, int times)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(graph_id >= 0 && graph_id < (int)ctx->graphs.size() && ctx->graphs[graph_id], "bad graph id");
    for (int t = 0; t < times; ++t) FS_HIP(hipGraphLaunch(ctx->graphs[graph_id], ctx->stream));
    return FS_OK;
}

int fs_graph_free(fs_ctx *ctx, int graph_id)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(graph_id >= 0 && graph_id < (int)ctx->graphs.size(), "bad graph id");
    if (ctx->graphs[graph_id]) {
        FS_HIP(hipStreamSynchronize(ctx->stream));
        FS_HIP(hipGraphExecDestroy(ctx->graphs[graph_id]));
        ctx->graphs[graph_id] = nullptr;
    }
    return FS_OK;
}

// ---- command tapes: the N > 1 counterpart of the hipGraph replay -------------------------------------------------------------
// A hipGraph cannot hold the RCCL ghost-row exchange of a slab run portably, so the launch sequence of a slab step (kernels on
// the compute stream + mark / begin / wait of the exchanges) is recorded as a list of host closures instead and re-issued by
// fs_tape_replay in a C++ loop: no Python, no ctypes marshalling and no validity bookkeeping between two launches (a 130 us
// slab step is otherwise driven by ~25 Python calls of 10-20 us each).
int fs_tape_begin(fs_ctx *ctx, int execute)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(!ctx->tape_rec && !ctx->capturing, "already recording / capturing");
    int rc = prof_drain(ctx); if (rc) return rc;
    ctx->tape_rec = new Tape();
    ctx->tape_execute = execute != 0;
    return FS_OK;
}

int fs_tape_end(fs_ctx *ctx, int *tape_id)
{
    FS_REQUIRE(ctx && tape_id, "null argument");
    FS_REQUIRE(ctx->tape_rec, "not recording");
    ctx->tapes.push_back(ctx->tape_rec);
    ctx->tape_rec = nullptr;
    ctx->tape_execute = true;
    *tape_id = (int)ctx->tapes.size() - 1;
    return FS_OK;
}

int fs_tape_length(fs_ctx *ctx, int tape_id, int *nops)
{
    FS_REQUIRE(ctx && nops, "null argument");
    FS_REQUIRE(tape_id >= 0 && tape_id < (int)ctx->tapes.size() && ctx->tapes[tape_id], "bad tape id");
    *nops = (int)ctx->tapes[tape_id]->ops.size();
    return FS_OK;
}

int fs_tape_replay(fs_ctx *ctx, int tape_id, int times)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(tape_id >= 0 && tape_id < (int)ctx->tapes.size() && ctx->tapes[tape_id], "bad tape id");
    FS_REQUIRE(!ctx->tape_rec && !ctx->capturing, "replay while recording / capturing");
    FS_HIP(hipSetDevice(ctx->device));
    const Tape *t = ctx->tapes[tape_id];
    for (int n = 0; n < times; ++n)
        for (const auto &op : t->ops) { int rc = op(); if (rc) return rc; }
    return FS_OK;
}

int fs_tape_free(fs_ctx *ctx, int tape_id)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(tape_id >= 0 && tape_id < (int)ctx->tapes.size(), "bad tape id");
    delete ctx->tapes[tape_id];
    ctx->tapes[tape_id] = nullptr;
    return FS_OK;
}

// ---- profiling --------------------------------------------------------------------------------------------------
int fs_prof_enable(fs_ctx *ctx, int on)
{
    FS_REQUIRE(ctx, "ctx is null");
    int rc = prof_drain(ctx); if (rc) return rc;
    ctx->prof_on = on != 0;
    return FS_OK;
}

int fs_prof_reset(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    int rc = prof_drain(ctx); if (rc) return rc;
    std::fill(ctx->prof_launches.begin(), ctx->prof_launches.end(), 0);
    std::fill(ctx->prof_ms.begin(), ctx->prof_ms.end(), 0.0);
    return FS_OK;
}

int fs_prof_count(fs_ctx *ctx, int *n)
{
    FS_REQUIRE(ctx && n, "null argument");
    int rc = prof_drain(ctx); if (rc) return rc;
    *n = (int)ctx->prof_names.size();
    return FS_OK;
}

int fs_prof_get(fs_ctx *ctx, int idx, char *name, int name_cap, int *launches, double *total_ms)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(idx >= 0 && idx < (int)ctx->prof_names.size(), "bad profile index");
    if (name && name_cap > 0) { strncpy(name, ctx->prof_names[idx].c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = ctx->prof_launches[idx];
    if (total_ms) *total_ms = ctx->prof_ms[idx];
    return FS_OK;
}

}  // extern "C"
