// fs_host.h - host-side objects behind the opaque C-ABI handles (include/fs_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    int ncomp = 0, nops = 0;
    int *comp_begin = nullptr, *comp_rlo = nullptr, *comp_rhi = nullptr;
    int *kind = nullptr, *tgt = nullptr, *s1 = nullptr, *s2 = nullptr;
    BcOps view() const { return BcOps{ncomp, comp_begin, comp_rlo, comp_rhi, kind, tgt, s1, s2}; }
};

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
    void *d_bc_const = nullptr, *d_bc_dye = nullptr;
    bool mask_set = false, bc_incomplete = false;
    int bc_radius_vel = 0, bc_radius_prs = 0;   // rows of pre-kernel data a rewritten boundary cell depends on
    fs::BcOpsDev ops_vel, ops_prs, ops_dye;
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    double *d_acc = nullptr;  // 2 doubles (residual)
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
    // comm
    fs::Comm *comm = nullptr;
    std::set<fs_field *> fields;  // live fields, released with the context
    // tuning knobs (env FS_MARCH=0: one-cell-per-lane kernels only)
    bool use_march = true;
    int xcd_group = 8;  // tile rows per XCD group (env FS_XCD_GROUP)
    int xcd_mask = 0;   // env FS_XCD: bit per kernel family that uses the XCD-group block mapping (see ov_grid)
    int stack_mask = 0;       // env FS_STACK: kernel families (XCD_* bits) launched with stacked workgroups
    bool pack_halo = true;    // env FS_PACK_HALO=0: one ncclSend/ncclRecv per field instead of one packed message per neighbour
    int jacobi_variant = 0;   // env FS_JACOBI: 0 = per-form default, 22 / 24 / 21 = overlapped-wave tiles of 2 / 4 / 1 rows, 30 = LDS tile

    fs::Grid grid() const
    {
        fs::Grid g;
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
};

namespace fs {

// dx (as rounded to T) is an exact power of two: division by it may be replaced by multiplication
template <typename T>
inline bool is_pow2(T x)
{
    int e;
    return x > 0 && std::frexp(x, &e) == (T)0.5;
}

template <typename T>
inline Konst<T> make_konst(double dt, double dx, double re, double weight = 0.0, double omega = 0.0)
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
    return k;
}


}  // namespace fs
