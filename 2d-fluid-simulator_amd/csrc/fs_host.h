// fs_host.h - host-side objects behind the opaque C-ABI handles (include/fs_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <array>
#include <cmath>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/fs_hip.h"
#include "fs_device.h"
#include "fs_kernels.h"
#include "fs_march.h"
#include "fs_rbpair.h"
#include "fs_jquad.h"
#include "fs_k34n.h"

namespace fs {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define FS_HIP(call)                                                         \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return fs::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define FS_REQUIRE(cond, msg)                  \
    do {                                       \
        if (!(cond)) {                         \
            fs::set_error(msg);                \
            return FS_ERR_ARG;                 \
        }                                      \
    } while (0)

struct BcOpsDev {
    int nsimple = 0, npair = 0, ncomp = 0, nops = 0;
    int4 *simple = nullptr, *pair = nullptr;
    int *comp_begin = nullptr, *comp_rlo = nullptr, *comp_rhi = nullptr;
    int *kind = nullptr, *tgt = nullptr, *s1 = nullptr, *s2 = nullptr, *row = nullptr, *srow = nullptr;
    int lanes() const { return nsimple + npair + ncomp; }
    BcOps view() const { return BcOps{nsimple, simple, npair, pair, ncomp, comp_begin, comp_rlo, comp_rhi, kind, tgt, s1, s2, row, srow}; }
};

// Which __global__ functions a profiled launch name stands for (fs_prof_kernels: bench.py quotes the demangled symbols instead of string
// literals).  Every kernel launch of the library is written FS_KLAUNCH(kernel, grid, block, lds, stream, args...): the host stub's address
// is noted per thread, and fs::launch() (fs_launch.h) files the notes of the callable it just ran under the launch's profile name.
struct KernelNotes { const void *fn[4]; int n; };
extern thread_local KernelNotes kernel_notes;      // fs_core.hip
#define FS_KLAUNCH(kern, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                       \
        if (fs::kernel_notes.n < 4) fs::kernel_notes.fn[fs::kernel_notes.n++] = (const void *)(kern);           \
        hipLaunchKernelGGLInternal((kern), (grid), (block), (lds), (stream), __VA_ARGS__);                     \
    } while (0)

struct ProfRec {
    int name_id;
    hipEvent_t start, stop;
};

struct Comm;  // fs_comm.hip

// recorded launch sequence (fs_tape_*): closures that re-issue a kernel launch / an exchange step with the arguments of the recording
struct Tape {
    std::vector<std::function<int()>> ops;
};

}  // namespace fs

struct fs_ctx {
    int device = 0;
    int X = 0, Y = 0, dtype = 0, y0 = 0, nyl = 0, halo = 0;
    int rows = 0, P = 0, Pm = 0;
    size_t esize = 4;
    hipStream_t stream = nullptr;
    uint8_t *d_mask = nullptr;
    uint8_t *d_bcmap = nullptr;    // [rows][Pm] recipe byte of the pressure boundary condition per cell (fs_march.h k_jacobi_lazy)
    uint8_t *d_lazyflags = nullptr;   // [nwx][rows] tile needs the lazy evaluation
    std::vector<uint8_t> h_bcmap;  // host copy between build_bc_ops and the upload
    bool lazy_ok = false;                    // mask admits the lazy pressure BC
    bool rb_pair_ok = false;                 // mask admits the two-iteration red-black pass (fs_rbpair.h; decided in build_bc_ops)
    bool jq_ok = false;                      // mask admits the four-sweep Jacobi pass (fs_jquad.h)
    int rbpair_split = 1;                    // plain and boundary workgroups of that pass (and of the four-sweep Jacobi pass) as two compact
                                             // launches: env FS_RBPAIR_SPLIT = 0 never, 1 on grids of 8 M cells or more, 2 always
    size_t small_cells = (size_t)1 << 21;    // 2-row tiles on grids below this many cells (env FS_SMALL_CELLS; 0: never): res 800 +4.7 %, res 1024 (2 M cells) +0.3 %
    uint32_t *d_pairlist = nullptr; int n_pairlist[2] = {0, 0};   // wave-tile rows of the two-sweep kernel's general path, without / with its
                                                                   // vertical-recipe tile path (fs_march.h k_pair_list): [2][nwx * rows] + 2 counters
    int nwx = 0;                   // wave columns of 62 quads across a row
    void *d_bc_const = nullptr, *d_bc_dye = nullptr;
    bool mask_set = false, bc_incomplete = false;
    int bc_radius_vel = 0, bc_radius_prs = 0;   // rows of pre-kernel data a rewritten boundary cell depends on
    fs::BcOpsDev ops_vel, ops_prs, ops_dye;
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    double *d_acc = nullptr;  // 2 doubles (residual)
    int barrier_wgs = 0;          // workgroups of 256 threads of those kernels this device keeps resident at once, with headroom (fs_create): the bound of their grids
    unsigned *d_sync = nullptr;   // k_velocity_bc_limit / k_dye_bc_limit: arrive / depart counters of the grid barrier of their rare path [0, 1] (zero between launches)
    double *d_partial = nullptr;   // per-block partial (sum, count) pairs of the residual reduction
    size_t partial_cap = 0;        // pairs
    // graphs
    bool capturing = false;
    std::vector<hipGraphExec_t> graphs;
    // tapes
    fs::Tape *tape_rec = nullptr;      // open recording
    bool tape_execute = true;          // ... that also executes what it records (false: record only)
    std::vector<fs::Tape *> tapes;
    // profiling
    bool prof_on = false;
    std::vector<std::string> prof_names;
    std::map<std::string, int> prof_ids;
    std::vector<fs::ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::vector<int> prof_launches;
    std::vector<double> prof_ms;
    int tile_list_misses = 0;     // launches that wanted a launch list and could not build one (capture / tape recording / cap): fs_tile_list_stats
    std::vector<std::vector<const void *>> prof_kernels;      // per name: the host stubs of the __global__ functions launched under it (fs_prof_kernels)
    hipEvent_t span_ev[2] = {nullptr, nullptr};      // fs_span_begin / fs_span_end
    // comm
    fs::Comm *comm = nullptr;
    std::set<fs_field *> fields;  // live fields, released with the context
    std::vector<fs_field *> deferred_free;   // fs_field_free during a hipGraph capture: released when the capture ends
    // tuning knobs (env FS_MARCH=0: one-cell-per-lane kernels only)
    bool use_march = true;
    bool use_pairs = true;     // lanes of 2 cells: even widths (every `res`); use_march: the quad kernels, X % 4 == 0
    int fuse_k2 = 2;           // env FS_FUSE_K2: 0 - fs_cip_step as its two calls, K2 then the fused K3 + K4 pass; 1 - K2 in registers on every tile, two launches
                               // (all-fluid tiles, the others); 2 - the same in ONE launch over both kinds of tile.  Same observable results.
    bool limit_gate = true;    // env FS_LIMIT_GATE=0: limit_field always reads the whole field (A/B; the results are the same)
    int stack_mask = 0;       // kernel families (XCD_* bits) launched with stacked workgroups (fs_create)

    // compact launches (fs_device.h Grid::tiles): per-cell activity of the scene on the host (bit 0: some cell of wave column wx - 248
    // cells - in row j is not deep wall, bit 1: the same for the 120-cell wave columns of the 2-cell-lane kernels), and the lists built from
    // it per launch geometry (key: lane width, rows per tile, stacked, group size)
    int tile_list_mask = 1 | 2 | 4 | 8 | 32;              // env FS_TILE_LIST: kernel families (XCD_* bits) launched compactly.  Measured at bc5 res 4096:
                                             // K3+K4 363 -> 346 us, red-black pair 215 -> 192, vorticity confinement (2-cell lanes) 97 -> 95, K2 (2-cell lanes) 105 -> 102, the plain Jacobi sweeps 87.2 -> 85.8 (reading v) / 75.5 -> 74.1 (source pair)
    std::vector<uint8_t> h_act4, h_act2, h_act2w;     // [wave column][local row]
    struct TileList { uint32_t *d = nullptr; int per_xcd = 0; int count = 0; };      // count: listed workgroups (without the padding)
    using TileKey = std::array<int, 10>;     // lanes, rows per tile, stacked, group, class, reach, waves per workgroup, parent tile rows, row range
    std::map<TileKey, TileList> tile_lists;

    fs::Grid grid() const
    {
        fs::Grid g;
        g.tiles = nullptr;
        g.X = X; g.P = P; g.Pm = Pm; g.rows = rows;
        int jlo = halo - y0, jhi = halo - y0 + (Y - 1);
        g.jlo = jlo < 0 ? 0 : jlo;
        g.jhi = jhi > rows - 1 ? rows - 1 : jhi;
        g.ybase = y0 - halo;
        g.mask = d_mask;
        return g;
    }
};

struct fs_field {
    fs_ctx *ctx = nullptr;
    int C = 1;
    void *d = nullptr;
    size_t bytes = 0;
    unsigned *hot = nullptr;   // device words: [0] "may hold a speed above 9.95" (fs_device.h; meaningful for 2-channel fields), [1], [2] the same, raised by the op list of a k_velocity_bc_limit launch of parity 0 / 1
};

namespace fs {

// HIP-event pair around a span of stream work that is not one kernel launch (fs_api.hip; the ghost-row exchange chain of fs_comm.hip)
ProfRec prof_span_begin(fs_ctx *c, const char *name, hipStream_t stream);
void prof_span_end(fs_ctx *c, const ProfRec &rec, hipStream_t stream);

// dx (as rounded to T) is an exact power of two: division by it may be replaced by multiplication
template <typename T>
inline bool is_pow2(T x)
{
    int e;
    return x > 0 && std::frexp(x, &e) == (T)0.5;
}

template <typename T>
inline Konst<T> make_konst(fs_ctx *ctx, double dt, double dx, double re, double weight = 0.0, double omega = 0.0)
{
    Konst<T> k;
    k.dt = (T)dt; k.dx = (T)dx; k.re = (T)re;
    k.two_dx = (T)(2.0 * dx);
    k.dx2_fold = (T)std::pow(dx, 2.0);   // Python's float ** int is libm pow()
    k.dx3_fold = (T)std::pow(dx, 3.0);
    k.dx_sq = k.dx * k.dx;
    k.six_dx = (T)6 * k.dx;
    k.eight_dt = (T)8 * k.dt;
    k.dtw = (T)(dt * weight);
    k.om = (T)omega;
    k.om1 = (T)(1.0 - omega);
    // exact-reciprocal shortcut: only when every dx-derived divisor is a power of two and its inverse is finite
    k.p2 = is_pow2(k.dx) && is_pow2(k.dx2_fold) && is_pow2(k.dx3_fold) && std::isfinite((double)((T)1 / k.dx3_fold)) ? 1 : 0;
    k.inv_dx = (T)1 / k.dx;
    k.inv_two_dx = (T)1 / k.two_dx;
    k.inv_dx_sq = (T)1 / k.dx_sq;
    k.inv_dx2_fold = (T)1 / k.dx2_fold;
    k.inv_dx3_fold = (T)1 / k.dx3_fold;
    k.r_dx = 1.0 / (double)k.dx; k.r_two_dx = 1.0 / (double)k.two_dx; k.r_dx_sq = 1.0 / (double)k.dx_sq; k.r_dx2_fold = 1.0 / (double)k.dx2_fold;
    k.r_dx3_fold = 1.0 / (double)k.dx3_fold; k.r_six_dx = 1.0 / (double)k.six_dx; k.r_eight_dt = 1.0 / (double)k.eight_dt; k.r_re = 1.0 / (double)k.re;
    return k;
}

// fs_device.h f64div: x / d can be EXACTLY a tie between two f32 denormals iff d is an even integer (d = D 2^e, D odd, e >= 1); the plain
// f64-multiply division is used only for divisors that admit no such tie
inline bool tie_free(double d)
{
    if (!(d == d) || d == 0.0 || std::isinf(d)) return false;
    int e;
    double m = std::frexp(std::fabs(d), &e);          // |d| = m 2^e, 0.5 <= m < 1
    while (m != std::floor(m)) { m *= 2.0; --e; }     // -> odd integer m times 2^e
    return e < 1;
}
// division mode of a launch (fs_device.h DM_*): which kinds of divisors a kernel has decides how many modes it instantiates
template <typename T> inline int f64_mode(const fs_ctx *c, const Konst<T> &k)      // f32 fields: the f64-multiply division, if every dx- / dt-derived divisor is tie-free
{
    const bool ok = tie_free(k.dx) && tie_free(k.two_dx) && tie_free(k.dx_sq) && tie_free(k.dx2_fold) && tie_free(k.dx3_fold) && tie_free(k.six_dx) && tie_free(k.eight_dt);
    return sizeof(T) == 4 && ok ? DM_F64 : DM_IEEE;
}
template <typename T> inline int dm_all(const fs_ctx *c, const Konst<T> &k) { return (k.p2 ? DM_P2 : 0) | f64_mode<T>(c, k); }   // dx-derived AND other divisors
template <typename T> inline int dm_dx(const fs_ctx *c, const Konst<T> &k) { return k.p2 ? DM_P2 : f64_mode<T>(c, k); }          // dx-derived divisors only
template <typename T> inline int dm_const(const fs_ctx *c, const Konst<T> &k) { return f64_mode<T>(c, k); }                      // no dx-derived divisor

}  // namespace fs
