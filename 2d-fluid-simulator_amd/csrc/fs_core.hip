// fs_core.hip - C-ABI entry points (include/fs_hip.h): contexts, fields, scene upload, boundary kernels, pointwise passes, visualisation,
// hipGraph capture, command tapes, profiling, box-rate probes.  Transport kernels: fs_transport.hip; pressure kernels: fs_pressure.hip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <mutex>
#include <numeric>
#include <type_traits>
#include <unordered_map>

#include "fs_launch.h"

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
thread_local KernelNotes kernel_notes = {{nullptr, nullptr, nullptr, nullptr}, 0};
hipEvent_t prof_event(fs_ctx *c)
{
    hipEvent_t e;
    if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
    hipEventCreate(&e);
    return e;
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
        c->prof_kernels.emplace_back();
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

// Compact list of the workgroups of a dense XCD-band launch that have anything to do (Grid::tiles), built once per geometry from the
// host-side activity maps of the scene.  lanes = cells per lane (4: wave columns of 248 cells, 2: of 120), rt = rows per tile.
// cls: 0 = every workgroup with work; 1 / 2 = those whose tiles see nothing but fluid within `reach` rows and the halo lanes ("plain":
// no mask loads, no boundary views - their own kernel and register budget) / the others
// `lanes` names the wave geometry: 4 = quads, 62 owner lanes (248 cells, 4 halo cells per side); 2 = pairs, 60 owner lanes (120 cells, 4 halo
// cells); 3 = pairs, 62 owner lanes (124 cells, 2 halo cells)
const fs_ctx::TileList *tile_list(fs_ctx *c, int lanes, int rt, bool stacked, int group, int nbx, int nby, int cls, int reach, int wgw, int jb, int je, int parent_rt)
{
    if (c->h_act4.empty() || nbx > 0xfff || nby > 0xffff || lanes > 4 || c->rows > 0xffff) return nullptr;      // (entry: class hints << 28 | by << 12 | bx)
    if (je < 0) je = c->rows;
    if (parent_rt == rt) parent_rt = 0;
    if (parent_rt && (wgw != 1 || parent_rt % rt != 0 || parent_rt > 64)) return nullptr;
    if (cls == 3 && !(rt == 8 && parent_rt == 16 && wgw == 1 && lanes == 2)) return nullptr;      // (a coarser plain tiling is defined for one-wave workgroups)
    const fs_ctx::TileKey key{{lanes, rt, stacked ? 1 : 0, group, cls, reach, wgw, parent_rt, jb, je}};      // (slab launches cover varying row ranges: one list per range)
    auto it = c->tile_lists.find(key);
    if (it != c->tile_lists.end()) return it->second.d ? &it->second : nullptr;
    // (building one allocates and synchronises the stream: not inside a capture or a tape recording - the dense grid then; and slab launches over ever new
    //  row ranges stop at 512 lists.  Both cases are counted: fs_tile_list_stats - a run whose warm-up covered its period reports none)
    if (c->capturing || c->tape_rec || c->tile_lists.size() >= 512) { ++c->tile_list_misses; return nullptr; }
    const std::vector<uint8_t> &act = lanes == 4 ? c->h_act4 : (lanes == 2 ? c->h_act2 : c->h_act2w);
    const int ow = geo_owners(lanes), waves = (c->X / geo_cells(lanes) + ow - 1) / ow, Y = c->rows;       // (activity maps are indexed by LOCAL row)
    std::vector<uint32_t> per[8];
    bool any_hint = false;
    // no non-fluid cell (bit 1 of the activity byte: halo lanes included) in wave columns [wx0, wx1) within `reach` rows of rows [p0, p1) - and
    // the whole box inside the domain: a wave column at the domain's first / last column clamps its halo lanes onto the edge cells, a row
    // range that leaves the slab has rows nobody classified (the reference's scenes keep walls there; an uploaded mask need not)
    auto plain_box = [&](int wx0, int wx1, int p0, int p1) -> bool {
        if (wx0 <= 0 || wx1 >= waves || p0 - reach < 0 || p1 + reach > Y) return false;
        for (int wx = wx0; wx < wx1; ++wx)
            for (int j = p0 - reach; j < p1 + reach; ++j)
                if (act[(size_t)wx * Y + j] & 2) return false;
        return true;
    };
    const int groups = (nby + group - 1) / group;
    // inside a group the workgroups are listed column by column: vertically adjacent workgroups, which re-read each other's halo rows, are
    // neighbours in dispatch order (bc5 res 4096: K3+K4 333 -> 319 us, the red-black pair 195 -> 191 against row by row)
    constexpr bool col_major = true;
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
                    if (cls == 3) {
                        // mixed list of the one-launch red-black pair (fs_rbpair.h k_rbsor_pair_all): units of rt = 8 rows; an all-fluid parent tile of
                        // parent_rt = 16 rows is ONE entry at its lower unit (hint bit 0), any other unit with work an entry with the per-4-row-tile
                        // "fluid in its own rows" bits (1, 2)
                        const int p0 = jb + (j0 - jb) / parent_rt * parent_rt, p1 = std::min(je, p0 + parent_rt);
                        if (p1 - p0 == parent_rt && plain_box(wx0, wx1, p0, p1)) {
                            if (j0 == p0) per[xcd].push_back((1u << 28) | ((uint32_t)by << 12) | (uint32_t)bx);
                        } else if (any) {
                            uint32_t h = 0u;
                            for (int s = 0; s < 2; ++s)
                                for (int j = j0 + 4 * s; j < std::min(j1, j0 + 4 * s + 4); ++j)
                                    if (act[(size_t)bx * Y + j] & 4) { h |= 2u << s; break; }
                            per[xcd].push_back((h << 28) | ((uint32_t)by << 12) | (uint32_t)bx);
                        }
                        any_hint = true;
                        continue;
                    }
                    if (any && cls) {
                        // plain: no non-fluid cell (bit 1 of the activity byte; halo lanes included) within `reach` rows of the tile - or, for the
                        // boundary list of a launch whose plain part runs on tiles of parent_rt rows, of the parent tile this tile lies in
                        // A tile (or parent tile) the row range cuts short is never plain: the plain kernels may store every row of their tile
                        // (the stacked red-black pair does - ADVICE r5: rows past row_end of a slab range), the kernels with masks guard `je`.
                        int p0 = j0, p1 = j1, full_rows = (stacked ? wgw : 1) * rt;
                        if (parent_rt) { p0 = jb + (j0 - jb) / parent_rt * parent_rt; p1 = std::min(je, p0 + parent_rt); full_rows = parent_rt; }
                        any = (p1 - p0 == full_rows && plain_box(wx0, wx1, p0, p1)) == (cls == 1);
                    }
                    uint32_t hints = 0u;
                    if (any && !cls && reach > 0 && wgw <= 4) {
                        // per-wave hint for a kernel that holds both paths (unsplit launches): wave w is plain - no non-fluid cell within `reach` rows
                        // of ITS tile, halo lanes included - and may skip its mask loads and the classification (band_coords cls)
                        for (int w = 0; w < wgw; ++w) {
                            const int wx = stacked ? bx : bx * wgw + w;
                            const int t0 = jb + (stacked ? by * wgw + w : by) * rt, t1 = std::min(je, t0 + rt);
                            if (wx >= waves || t0 >= je) continue;
                            const bool plain = plain_box(wx, wx + 1, t0, t1);
                            if (plain) hints |= 1u << w;
                        }
                        any_hint = any_hint || hints != 0u;
                    }
                    if (any && cls == 2 && wgw == 1) {
                        // boundary list of a multi-part launch (one-wave workgroups): bit 1 of the hint = "a fluid cell in the tile's own rows, halo lanes
                        // included" - the general kernels then request their window without waiting for the masks that would tell them so
                        for (int j = j0; j < j1; ++j)
                            if (act[(size_t)bx * Y + j] & 4) { hints |= 2u; break; }
                    }
                    if (any) per[xcd].push_back((hints << 28) | ((uint32_t)by << 12) | (uint32_t)bx);
                }
            }
    if ((!cls || cls == 3) && any_hint && wgw == 1) {
        // one launch over both kinds of tile (fs_cip_step): the tiles that take the longer, masked body go FIRST in each XCD's list - the all-fluid tiles fill in
        // behind them and the launch does not end on the slow ones (round 6: 281.5-282.7 -> 279.1-280.7 us; the other way round 283.6-284.6)
        for (auto &v : per) std::stable_partition(v.begin(), v.end(), [](uint32_t e) { return ((e >> 28) & 1u) == 0u; });
    }
    size_t K = 0, total = 0;
    for (auto &v : per) total += v.size();
    if (total >= 64) {
        // The geometry deals a class of tiles unevenly (bc5 res 4096: the boundary tiles of the red-black pair 1003 .. 1365 per XCD) and a compact
        // launch lasts as long as its fullest XCD.  An entry names its tile, so any XCD may run it: the surplus of an XCD - the END of its list, whole
        // runs of vertically adjacent tiles - goes to the end of the emptiest lists.  Those tiles read their halo rows through another L2; they are
        // a few per cent of the list.
        const size_t target = (total + 7) / 8;
        for (int d = 0; d < 8; ++d)
            while (per[d].size() > target) {
                int r = 0;
                for (int x = 1; x < 8; ++x) if (per[x].size() < per[r].size()) r = x;
                if (per[r].size() >= target) break;
                const size_t n = std::min(per[d].size() - target, target - per[r].size());
                per[r].insert(per[r].end(), per[d].end() - n, per[d].end());
                per[d].resize(per[d].size() - n);
            }
    }
    for (auto &v : per) K = std::max(K, v.size());
    fs_ctx::TileList tl;
    if (K > 0 && (cls || any_hint || total < (size_t)nbx * nby)) {        // (nothing to skip, no hint to give: the dense grid needs no list)
        std::vector<uint32_t> h(K * 8, 0xffffffffu);
        for (int xcd = 0; xcd < 8; ++xcd)
            for (size_t k = 0; k < per[xcd].size(); ++k) h[k * 8 + xcd] = per[xcd][k];
        if (hipMalloc(&tl.d, h.size() * sizeof(uint32_t)) == hipSuccess &&
            hipMemcpyAsync(tl.d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess)
            { tl.per_xcd = (int)K; tl.count = (int)total; }
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

int check_rows(const fs_ctx *c, int jb, int je)
{
    if (!(0 <= jb && jb <= je && je <= c->rows)) {
        set_error("row range outside the local slab");
        return FS_ERR_ARG;
    }
    return FS_OK;
}

int check_field(const fs_ctx *c, const fs_field *f, int C, const char *what)
{
    if (!f || f->ctx != c || f->C != C) {
        set_error(std::string("field argument '") + what + "' is null, from another context, or has the wrong channel count");
        return FS_ERR_ARG;
    }
    return FS_OK;
}

int ensure_stage(fs_ctx *c, size_t bytes)
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
        } else if (kept.size() == 2) {      // a chain of two: two flat records side by side (BcOps::pair)
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
    // The components are independent of each other (no cell in common), so the flat records may run in any order: sorted by (target row, position
    // in the row) the lanes of a wave walk ALONG the rows of the field layout - the targets and sources of a horizontal wall share cache lines - where
    // the reference's serial (i-major) order hands consecutive lanes cells 2 P elements apart (round 6).
#ifndef FS_BC_UNSORTED      // (A/B build flavour: the serial order of round 5)
    {
        auto key = [](const int4 &r) { return ((long long)(r.w >> 2) << 32) | (unsigned)r.x; };
        std::sort(h_simple.begin(), h_simple.end(), [&](const int4 &a, const int4 &b) { return key(a) < key(b); });
        std::vector<std::pair<int4, int4>> pr;
        for (size_t q = 0; q + 1 < h_pair.size(); q += 2) pr.emplace_back(h_pair[q], h_pair[q + 1]);
        std::sort(pr.begin(), pr.end(), [&](const std::pair<int4, int4> &a, const std::pair<int4, int4> &b) { return key(a.first) < key(b.first); });
        for (size_t q = 0; q < pr.size(); ++q) { h_pair[2 * q] = pr[q].first; h_pair[2 * q + 1] = pr[q].second; }
    }
#endif
    h_begin.push_back((int)h_kind.size());
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

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }
const char *fs_last_error(void) { return g_err.c_str(); }

// Launch lists of this context (fs_core.hip tile_list): how many were built so far (each costs one hipMalloc + a stream synchronisation at the first
// launch of its geometry / row range), and how many launches wanted one they could not build - inside a hipGraph capture or a tape recording, or beyond the
// cap of 512 - and ran (or were recorded) as dense grids instead.  bench.py samples this around its timed region: 0 built, 0 misses.
int fs_tile_list_stats(const fs_ctx *ctx, int *built, int *misses)
{
    FS_REQUIRE(ctx && built && misses, "null argument");
    *built = (int)ctx->tile_lists.size();
    *misses = ctx->tile_list_misses;
    return FS_OK;
}

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
    if (e == hipSuccess) {
        // how many workgroups of the merged limit + boundary kernels THIS device holds at once (their rare path meets at a grid barrier: every
        // workgroup must be resident - a partitioned or smaller part holds fewer than a whole MI355X): occupancy x compute units, half of it
        // left to whatever else is resident (the tail of the kernel before)
        hipDeviceProp_t prop;
        int bv = 0, bd = 0;
        e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) e = dtype == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&bv, k_velocity_bc_limit<float>, 256, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&bv, k_velocity_bc_limit<double>, 256, 0);
        if (e == hipSuccess) e = dtype == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&bd, k_dye_bc_limit<float>, 256, 0)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&bd, k_dye_bc_limit<double>, 256, 0);
        if (e == hipSuccess) c->barrier_wgs = std::min(bv, bd) * prop.multiProcessorCount / 2;
    }
    c->nwx = (nx / 4 + 61) / 62;
    if (e != hipSuccess) { fs_destroy(c); return hip_fail(e, "hipMalloc(ctx)", __FILE__, __LINE__); }
    if (const char *s = getenv("FS_MARCH")) c->use_march = atoi(s) != 0;
    if (const char *s = getenv("FS_TILE_LIST")) c->tile_list_mask = atoi(s);
    if (const char *s = getenv("FS_LIMIT_GATE")) c->limit_gate = atoi(s) != 0;
    if (const char *s = getenv("FS_FUSE_K2")) c->fuse_k2 = std::max(0, std::min(2, atoi(s)));
    if (const char *s = getenv("FS_RBPAIR_SPLIT")) c->rbpair_split = atoi(s);
    if (const char *s = getenv("FS_SMALL_CELLS")) c->small_cells = (size_t)atoll(s);
    c->stack_mask = XCD_RBSOR | XCD_ADVECT | XCD_GRAD;      // measured per family: K4 313 -> 301 us, K3 246 -> 243, RB-SOR 129 -> 127.5; the others lose 1 %
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
    for (hipEvent_t ev : ctx->span_ev) if (ev) hipEventDestroy(ev);
    if (ctx->d_bcmap) hipFree(ctx->d_bcmap);
    if (ctx->d_lazyflags) hipFree(ctx->d_lazyflags);
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
    if (esize == 1) FS_KLAUNCH(k_to_device<uint8_t>, grid, dim3(256), 0, ctx->stream, (const uint8_t *)ctx->d_stage, (uint8_t *)dev, ctx->X, Q, pitch, row_begin * C);
    else if (esize == 4) FS_KLAUNCH(k_to_device<float>, grid, dim3(256), 0, ctx->stream, (const float *)ctx->d_stage, (float *)dev, ctx->X, Q, pitch, row_begin * C);
    else FS_KLAUNCH(k_to_device<double>, grid, dim3(256), 0, ctx->stream, (const double *)ctx->d_stage, (double *)dev, ctx->X, Q, pitch, row_begin * C);
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
                    const uint8_t nf = m[j] != 0 ? 2 : 4;            // bit 1: a cell that is not fluid, bit 2: a fluid cell - both with the halo lanes of the neighbouring column
                    a[lr] |= (uint8_t)((m[j] != 1) | (b[j] != 0)) | nf;
                    if (h) h[lr] |= nf;
                }
            }
        }
    }
    std::vector<uint8_t>().swap(ctx->h_bcmap);
    if (rc) return rc;
    if (ctx->X % 4 == 0) {      // per-tile flags of the lazy pressure BC
        FS_KLAUNCH(k_lazy_flags, dim3((ctx->nwx * ctx->rows + 3) / 4), dim3(256), 0, ctx->stream, ctx->grid(), ctx->nwx, ctx->d_bcmap, ctx->d_lazyflags);
        FS_HIP(hipGetLastError());
        // the rows the two-sweep kernel hands to its general path: list + count (read back once per mask)
        const size_t cap = (size_t)ctx->nwx * ctx->rows;
        if (!ctx->d_pairlist) FS_HIP(hipMalloc(&ctx->d_pairlist, (2 * cap + 2) * sizeof(uint32_t)));
        unsigned *d_count = (unsigned *)(ctx->d_pairlist + 2 * cap);
        FS_HIP(hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned), ctx->stream));
        FS_KLAUNCH(k_pair_list, dim3((ctx->nwx * ctx->rows + 3) / 4), dim3(256), 0, ctx->stream, ctx->grid(), ctx->nwx, (uint8_t *)ctx->d_lazyflags,
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
            FS_KLAUNCH(k_fill<T>, dim3(2048), dim3(256), 0, ctx->stream, (T *)f->d, n, (T)value);
            FS_KLAUNCH(k_fill<unsigned>, dim3(1), dim3(64), 0, ctx->stream, f->hot, (size_t)4, hot);
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
        FS_KLAUNCH(k_scan_hot<T>, cells_grid(ctx, row_begin, row_begin + nrows), dim3(256), 0, ctx->stream, ctx->grid(), row_begin, (const T *)f->d, f->hot);
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
        FS_KLAUNCH(k_to_host<T>, grid, dim3(256), 0, ctx->stream, (T *)ctx->d_stage, (const T *)f->d, ctx->X, Q, ctx->P, row_begin * f->C);
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
    FS_KLAUNCH(k_hot_fold, dim3(1), dim3(64), 0, dst->ctx->stream, dst->hot, (const unsigned *)src->hot);      // (one word: the copy starts a new parity sequence)
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
            FS_KLAUNCH(k_velocity_bc<T>, dim3((ctx->ops_vel.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
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
    // every workgroup resident (the grid barrier of the rare path): the launch has max(wgs, <= 64) workgroups, the device holds barrier_wgs at once
    // (use_pairs: every `res` - the limit pass walks the rows in quads and stops at the row's width, fs_march.h limit_pass_and_barrier)
    *ok = ctx->mask_set && ctx->use_pairs && ctx->limit_gate && ctx->d_sync && ctx->d_bc_const && wgs >= 1 && std::max(wgs, 64) <= ctx->barrier_wgs ? 1 : 0;
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
            FS_KLAUNCH(k_velocity_bc_limit<T>, dim3(std::max((ctx->ops_vel.lanes() + 255) / 256, std::min(64, limit_end - limit_begin))), dim3(256), 0, ctx->stream,
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
            FS_KLAUNCH(k_pressure_bc<T>, dim3((ctx->ops_prs.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
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
            FS_KLAUNCH(k_dye_bc<T>, dim3((ctx->ops_dye.lanes() + 255) / 256), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_dye.view(), row_begin, row_end, (T *)dye->d, (const T *)ctx->d_bc_dye);
        });
    })
}

int fs_dye_bc_limit_ok(const fs_ctx *ctx, int *ok)
{
    FS_REQUIRE(ctx && ok, "null argument");
    const int wgs = (ctx->ops_dye.lanes() + 255) / 256;
    *ok = ctx->mask_set && ctx->use_pairs && ctx->limit_gate && ctx->d_sync && ctx->d_bc_dye && wgs >= 1 && std::max(wgs, 64) <= ctx->barrier_wgs ? 1 : 0;
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
            FS_KLAUNCH(k_dye_bc_limit<T>, dim3(std::max((ctx->ops_dye.lanes() + 255) / 256, std::min(64, limit_end - limit_begin))), dim3(256), 0, ctx->stream,
                               ctx->grid(), ctx->ops_dye.view(), row_begin, row_end, limit_begin, limit_end, (T)limit, (T *)v->d, v->hot, ctx->d_sync,
                               (T *)dye->d, (const T *)ctx->d_bc_dye);
        });
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
        FS_KLAUNCH(k_verify_f64div, dim3(1u << 15, 10 + 2), dim3(256), 0, ctx->stream, d, 1.0 / (double)d, tie_free((double)d) ? 0 : 1, flag);   // the form the library uses for this divisor
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

// the same chains on PACKED operands (v2f: v_pk_mul_f32 / v_pk_add_f32, two IEEE f32 operations per lane and instruction): what the packed bodies
// of the transport kernels issue.  (Round 3 read "half rate" from tools/valu_rate.hip - whose scalar baseline the SLP vectoriser had packed as
// well; built with the library's flags, -fno-slp-vectorize, the scalar loop above stays scalar and the comparison is what it says.)
__global__ __launch_bounds__(256) static void k_box_valu_pk(float *sink, float a, float b, int iters)
{
    v2f x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u].x = a + (float)(threadIdx.x + u); x[u].y = b + (float)(threadIdx.x + 2 * u); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = x[u] * a;
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = x[u] + b;
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u].x + x[u].y;
    if (s == 1.2345f) sink[0] = s;
}

// packed != 0: the packed chains (wave-instructions per second and SIMD, each doing two operations per lane)
static int box_valu_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd, int packed)
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
        for (int r = 0; r < reps; ++r) {
            if (packed) FS_KLAUNCH(k_box_valu_pk, dim3(cus * 4), dim3(256), 0, ctx->stream, sink, 1.0000001f, 1e-9f, iters);
            else FS_KLAUNCH(k_box_valu, dim3(cus * 4), dim3(256), 0, ctx->stream, sink, 1.0000001f, 1e-9f, iters);
        }
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
int fs_box_valu_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd) { return box_valu_rate(ctx, budget_ms, ginstr_per_simd, 0); }
int fs_box_valu_pk_rate(fs_ctx *ctx, double budget_ms, double *ginstr_per_simd) { return box_valu_rate(ctx, budget_ms, ginstr_per_simd, 1); }

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
        for (int r = 0; r < reps; ++r) FS_KLAUNCH(k_box_mixed, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n, 1.0000001f, 1e-9f);
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
                if (copy) FS_KLAUNCH(k_box_copy, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, a, b, n);
                else FS_KLAUNCH(k_box_read, dim3(2048), dim3(256), 0, ctx->stream, a, sink, n);
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

// ---- pointwise -----------------------------------------------------------------------------------------------
int fs_limit_field(fs_ctx *ctx, double limit, fs_field *v, int row_begin, int row_end)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_FIELD(v, 2);
    FS_ROWS();
    FS_DISPATCH(ctx, {
        if (ctx->use_pairs) {      // (any even width: the quads stop at the row's width)
            // gated by the buffer's "hot" flag (fs_device.h): while no writer has stored a speed above 9.95 the pass has nothing to do
            const int gated = (T)limit * (T)limit > (T)FS_HOT_GATE_SQ && ctx->limit_gate ? 1 : 0;
            const int lanes = std::min(row_end - row_begin, 256);
            return launch(ctx, "limit_field", [=] {
                FS_KLAUNCH((k_limit_quad<T>), dim3(((ctx->X + 3) / 4 + 255) / 256, lanes), dim3(256), 0, ctx->stream,
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
            if (mode == 0) FS_KLAUNCH((k_visualize<0, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else if (mode == 1) FS_KLAUNCH((k_visualize<1, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else if (mode == 2) FS_KLAUNCH((k_visualize<2, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
            else FS_KLAUNCH((k_visualize<3, T>), grid, dim3(256), 0, ctx->stream, ctx->grid(), k, row_begin, (T *)rgb->d, pa, pb);
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

// one HIP-event pair around whatever the caller queues in between (bench.py: a whole ping-pong of sweeps - launch boundaries included,
// SURVEY.md 8d's "bytes / event time averaged over the sweeps"); independent of the per-launch profile
int fs_span_begin(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    FS_REQUIRE(!ctx->capturing && !ctx->tape_rec, "fs_span_begin during graph capture / tape recording");
    FS_HIP(hipSetDevice(ctx->device));
    if (!ctx->span_ev[0]) { FS_HIP(hipEventCreate(&ctx->span_ev[0])); FS_HIP(hipEventCreate(&ctx->span_ev[1])); }
    FS_HIP(hipEventRecord(ctx->span_ev[0], ctx->stream));
    return FS_OK;
}
int fs_span_end(fs_ctx *ctx, double *ms)
{
    FS_REQUIRE(ctx && ms, "null argument");
    FS_REQUIRE(ctx->span_ev[0], "fs_span_end without fs_span_begin");
    FS_HIP(hipEventRecord(ctx->span_ev[1], ctx->stream));
    FS_HIP(hipEventSynchronize(ctx->span_ev[1]));
    float f = 0.f;
    FS_HIP(hipEventElapsedTime(&f, ctx->span_ev[0], ctx->span_ev[1]));
    *ms = f;
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

// The __global__ functions launched under profile name `name` since profiling was enabled, demangled, one per line (the names a
// rocprofv3 --kernel-trace of the same run shows).  Returns the number of kernels in *n; `out` may be null.
int fs_prof_kernels(fs_ctx *ctx, const char *name, char *out, int capacity, int *n)
{
    FS_REQUIRE(ctx && name && n, "null argument");
    *n = 0;
    if (out && capacity > 0) out[0] = 0;
    auto it = ctx->prof_ids.find(name);
    if (it == ctx->prof_ids.end()) return FS_OK;
    std::string all;
    for (const void *fn : ctx->prof_kernels[it->second]) {
        const char *mangled = hipKernelNameRefByPtr(fn, ctx->stream);
        if (!mangled) continue;
        int status = 0;
        char *dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
        std::string s = status == 0 && dem ? dem : mangled;
        free(dem);
        // (the signature is noise: "void fs::k_x<4, 0>(fs::Grid, ...)" -> "fs::k_x<4, 0>")
        if (s.rfind("void ", 0) == 0) s = s.substr(5);
        int depth = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == '<') ++depth;
            else if (s[i] == '>') --depth;
            else if (s[i] == '(' && depth == 0) { s.resize(i); break; }
        }
        if (!all.empty()) all += "\n";
        all += s;
        ++*n;
    }
    if (out && capacity > 0) { strncpy(out, all.c_str(), capacity - 1); out[capacity - 1] = 0; }
    return FS_OK;
}

}  // extern "C"
