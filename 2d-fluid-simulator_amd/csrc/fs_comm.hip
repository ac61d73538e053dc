// fs_comm.hip - y-slab halo exchange over RCCL (xGMI on an MI355X node).
//
// New relative to the reference (which is single-device): the grid is cut along y into one slab per GPU
// / process; a field's ghost rows are refreshed from the slab neighbours with one grouped
// ncclSend/ncclRecv pair per neighbour.  A y-halo of `depth` rows is ONE contiguous block of
// depth*C*P elements in the [row][channel][x] device layout, so no packing kernels are needed.
// Each rank talks to at most two peers (its slab neighbours), i.e. point-to-point traffic on dedicated
// xGMI links; messages are 64 KiB - a few MiB.
//
// Streams: every RCCL call of a context is issued on ONE dedicated communication stream (the communicator never sees
// two streams).  fs_halo_exchange_begin() makes that stream wait for everything already queued on the compute stream
// (event), queues pack -> grouped send/recv -> unpack there and records a completion event; fs_halo_exchange_wait() makes
// the compute stream wait for it.  Between the two calls the caller may launch kernels that neither read ghost rows nor
// write the `depth` outermost owned rows of the exchanged fields (the interior rows of the kernel that needed the
// exchange): they overlap with the transfer.  A blocking exchange is begin + wait.
//
// librccl is dlopen()ed on first use so that single-GPU users never pay for loading it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include "fs_host.h"

namespace fs {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int load_rccl()
{
    if (g_rccl.handle) return FS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { set_error(std::string("cannot load librccl: ") + dlerror()); return FS_ERR_COMM; }
#define SYM(field, name)                                                                   \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                            \
    if (!g_rccl.field) { set_error("librccl lacks symbol " name); dlclose(h); return FS_ERR_COMM; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.handle = h;
    return FS_OK;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    double *d_red = nullptr;
    char *stage = nullptr;       // 4 equal parts: send-to-lower, send-to-upper, recv-from-lower, recv-from-upper
    size_t stage_part = 0;       // bytes per part
    hipStream_t stream = nullptr;            // the communication stream
    hipEvent_t ev_compute = nullptr, ev_comm = nullptr;
    bool in_flight = false;      // begin() issued, wait() pending
    bool armed = false;          // ... and ev_comm was recorded for it
    bool marked = false;         // fs_halo_exchange_mark() already recorded ev_compute for the next begin()
    bool loopback = false;       // 1-rank communicator: the rank is its own lower and upper neighbour (self-test)
    bool own_stream = false;     // exchanges run on the communication stream (needed to overlap them with kernels: FS_OVERLAP=1); default:
                                 // on the compute stream itself - no event hand-offs, ~25 us less per exchange (measured in loop-back)
};

// Ghost-row blocks of several fields <-> one contiguous staging buffer per direction.  A grouped RCCL call costs ~2 us per
// ncclSend/ncclRecv in it (measured in loop-back: 8 us for one field, 49 us for six, independent of the bytes), so n fields
// are packed into ONE message per neighbour and direction: 2 sends + 2 receives however many fields are stale.
constexpr int MAX_PACK = 16;
struct PackTable {
    char *lo[MAX_PACK];          // field rows adjacent to the lower neighbour (owned rows when packing, ghost rows when unpacking)
    char *hi[MAX_PACK];
    size_t bytes[MAX_PACK];      // block size of field k (depth * C * P * esize), a multiple of 256
    size_t off[MAX_PACK];        // offset of field k inside a staging part
    unsigned *hot[MAX_PACK];     // unpacking: flag word of a 2-channel (velocity) field, else null (fs_device.h "hot" flag)
    int n, f64;
};
template <bool PACK>
__global__ __launch_bounds__(256) void k_halo_pack(PackTable t, char *stage_lo, char *stage_hi)
{
    const int k = blockIdx.y >> 1, side = blockIdx.y & 1;
    char *fieldp = side ? t.hi[k] : t.lo[k];
    char *stagep = (side ? stage_hi : stage_lo) + t.off[k];
    if (!fieldp) return;
    const size_t nvec = t.bytes[k] >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(PACK ? fieldp : stagep);
    uint4 *dst = reinterpret_cast<uint4 *>(PACK ? stagep : fieldp);
    unsigned *hot = PACK ? nullptr : t.hot[k];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        dst[i] = v;
        if (!PACK && hot) {      // ghost rows of a velocity buffer arrive from the neighbour: a speed above 9.95 there raises OUR flag too
            bool h;              // (component-wise and therefore conservative: the two channels of a cell sit in different rows of the block)
            if (t.f64) {
                const double a = __hiloint2double((int)v.y, (int)v.x), b = __hiloint2double((int)v.w, (int)v.z);
                h = hot1(a) || hot1(b);
            } else
                h = hot1(__uint_as_float(v.x)) || hot1(__uint_as_float(v.y)) || hot1(__uint_as_float(v.z)) || hot1(__uint_as_float(v.w));
            raise_hot(hot, h);
        }
    }
}

static int nccl_fail(ncclResult_t r, const char *what)
{
    set_error(std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    return FS_ERR_COMM;
}
#define FS_NCCL(call)                                               \
    do {                                                            \
        ncclResult_t r__ = (call);                                  \
        if (r__ != ncclSuccess) return fs::nccl_fail(r__, #call);   \
    } while (0)

}  // namespace fs

using namespace fs;

extern "C" {

// librccl can be loaded and has every entry point the exchange uses (no GPU call, no communicator): what an N > 1 job checks on every rank
// BEFORE any rank blocks in ncclCommInitRank (bench.py preflight).  *ok = 0 leaves the reason in fs_last_error().
int fs_comm_available(int *ok)
{
    FS_REQUIRE(ok, "null argument");
    *ok = load_rccl() == FS_OK ? 1 : 0;
    return FS_OK;
}

int fs_comm_unique_id(void *out_128_bytes)
{
    FS_REQUIRE(out_128_bytes, "null argument");
    static_assert(sizeof(ncclUniqueId) == FS_UNIQUE_ID_BYTES, "ncclUniqueId size");
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id;
    FS_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out_128_bytes, &id, sizeof id);
    return FS_OK;
}

int fs_comm_init(fs_ctx *ctx, int rank, int nranks, const void *unique_id_128_bytes)
{
    FS_REQUIRE(ctx && unique_id_128_bytes, "null argument");
    FS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    FS_REQUIRE(!ctx->comm, "communicator already initialised");
    int rc = load_rccl(); if (rc) return rc;
    FS_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof id);
    Comm *cm = new Comm();
    cm->rank = rank; cm->nranks = nranks;
    // ncclCommInitRank blocks until every rank has arrived.  If one never does (a rank that died while importing, a stale unique id) the
    // others would hang for ever: it runs in a helper thread and the caller gives up after FS_COMM_TIMEOUT seconds (default 180) with a
    // clear error.  (The helper stays blocked and is abandoned; the process is expected to exit on this error.)
    struct InitState { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto st = std::make_shared<InitState>();
    const int device = ctx->device;
    std::thread([st, nranks, id, rank, device] {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = g_rccl.CommInitRank(&c, nranks, id, rank);
        std::lock_guard<std::mutex> lock(st->mu);
        st->r = r; st->comm = c; st->done = true;
        st->cv.notify_all();
    }).detach();
    double limit = 180.0;
    if (const char *s = getenv("FS_COMM_TIMEOUT")) limit = atof(s);
    {
        std::unique_lock<std::mutex> lock(st->mu);
        if (!st->cv.wait_for(lock, std::chrono::duration<double>(limit), [&] { return st->done; })) {
            delete cm;
            char buf[256];
            snprintf(buf, sizeof buf, "ncclCommInitRank: rank %d of %d still waiting for its peers after %.0f s (FS_COMM_TIMEOUT) - "
                     "a rank is missing or the unique id is stale", rank, nranks, limit);
            set_error(buf);
            return FS_ERR_COMM;
        }
    }
    cm->comm = st->comm;
    const ncclResult_t r = st->r;
    if (r != ncclSuccess) { delete cm; return nccl_fail(r, "ncclCommInitRank"); }
    hipError_t e = hipMalloc(&cm->d_red, 16 * sizeof(double));
    if (e == hipSuccess) {      // highest priority: the short pack / RCCL / unpack kernels must not queue behind the interior rows they overlap with
        int least = 0, greatest = 0;
        e = hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&cm->stream, hipStreamNonBlocking, greatest);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_compute, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_comm, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (cm->d_red) hipFree(cm->d_red);
        if (cm->stream) hipStreamDestroy(cm->stream);
        if (cm->ev_compute) hipEventDestroy(cm->ev_compute);
        g_rccl.CommDestroy(cm->comm); delete cm;
        return hip_fail(e, "hipMalloc / stream / event (comm)", __FILE__, __LINE__);
    }
    if (const char *s = getenv("FS_OVERLAP")) cm->own_stream = atoi(s) != 0;
    ctx->comm = cm;
    return FS_OK;
}

int fs_comm_destroy(fs_ctx *ctx)
{
    if (!ctx || !ctx->comm) return FS_OK;
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm->stream) hipStreamSynchronize(ctx->comm->stream);
    if (ctx->comm->d_red) hipFree(ctx->comm->d_red);
    if (ctx->comm->stage) hipFree(ctx->comm->stage);
    if (ctx->comm->ev_compute) hipEventDestroy(ctx->comm->ev_compute);
    if (ctx->comm->ev_comm) hipEventDestroy(ctx->comm->ev_comm);
    if (ctx->comm->stream) hipStreamDestroy(ctx->comm->stream);
    if (ctx->comm->comm) g_rccl.CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
    return FS_OK;
}

// valid[k] (or 0): ghost rows of field k that are still correct, counted from the slab edge outwards - only the rows beyond them,
// i.e. depth offsets [valid[k], depth), travel.
static int exchange_direct(fs_ctx *ctx, hipStream_t xs, fs_field *const *fields, const int *valid, int nfields, int depth, int lower, int upper)
{
    Comm *cm = ctx->comm;
    const ncclDataType_t dt = ctx->dtype == 0 ? ncclFloat32 : ncclFloat64;
    const int H = ctx->halo, n = ctx->nyl;
    FS_NCCL(g_rccl.GroupStart());
    for (int k = 0; k < nfields; ++k) {
        fs_field *f = fields[k];
        const int v = valid ? valid[k] : 0;
        if (v >= depth) continue;
        const size_t row_elems = (size_t)f->C * ctx->P;
        const size_t count = (size_t)(depth - v) * row_elems;
        char *base = (char *)f->d;
        auto rowp = [&](int r) { return base + (size_t)r * row_elems * ctx->esize; };
        if (lower >= 0) {
            FS_NCCL(g_rccl.Send(rowp(H + v), count, dt, lower, cm->comm, xs));
            FS_NCCL(g_rccl.Recv(rowp(H - depth), count, dt, lower, cm->comm, xs));
        }
        if (upper >= 0) {
            FS_NCCL(g_rccl.Send(rowp(H + n - depth), count, dt, upper, cm->comm, xs));
            FS_NCCL(g_rccl.Recv(rowp(H + n + v), count, dt, upper, cm->comm, xs));
        }
    }
    FS_NCCL(g_rccl.GroupEnd());
    for (int k = 0; k < nfields; ++k) {      // ghost rows of a velocity buffer came in: keep its "hot" flag honest (fs_device.h)
        fs_field *f = fields[k];
        const int v = valid ? valid[k] : 0;
        if (f->C != 2 || v >= depth) continue;
        const dim3 gridv((ctx->X + 255) / 256, depth - v);
        if (ctx->dtype == 0) {
            if (lower >= 0) FS_KLAUNCH(k_scan_hot<float>, gridv, dim3(256), 0, xs, ctx->grid(), H - depth, (const float *)f->d, f->hot);
            if (upper >= 0) FS_KLAUNCH(k_scan_hot<float>, gridv, dim3(256), 0, xs, ctx->grid(), H + n + v, (const float *)f->d, f->hot);
        } else {
            if (lower >= 0) FS_KLAUNCH(k_scan_hot<double>, gridv, dim3(256), 0, xs, ctx->grid(), H - depth, (const double *)f->d, f->hot);
            if (upper >= 0) FS_KLAUNCH(k_scan_hot<double>, gridv, dim3(256), 0, xs, ctx->grid(), H + n + v, (const double *)f->d, f->hot);
        }
    }
    FS_HIP(hipGetLastError());
    return FS_OK;
}

static int exchange_packed(fs_ctx *ctx, hipStream_t xs, fs_field *const *fields, const int *valid, int nfields, int depth, int lower, int upper)
{
    Comm *cm = ctx->comm;
    const int H = ctx->halo, n = ctx->nyl;
    PackTable own, ghost;
    own.n = ghost.n = nfields;
    own.f64 = ghost.f64 = ctx->dtype == 1;
    size_t total = 0;
    for (int k = 0; k < nfields; ++k) {
        fs_field *f = fields[k];
        const size_t row_bytes = (size_t)f->C * ctx->P * ctx->esize;
        char *base = (char *)f->d;
        const int v = valid ? std::min(valid[k], depth) : 0;
        own.bytes[k] = ghost.bytes[k] = (size_t)(depth - v) * row_bytes;
        own.hot[k] = nullptr;
        ghost.hot[k] = f->C == 2 ? f->hot : nullptr;
        own.off[k] = ghost.off[k] = total;
        total += own.bytes[k];
        const bool any = v < depth;
        own.lo[k] = lower >= 0 && any ? base + (size_t)(H + v) * row_bytes : nullptr;             // my first owned rows, offsets [v, depth)
        own.hi[k] = upper >= 0 && any ? base + (size_t)(H + n - depth) * row_bytes : nullptr;     // my last owned rows, offsets [v, depth) from the top
        ghost.lo[k] = lower >= 0 && any ? base + (size_t)(H - depth) * row_bytes : nullptr;       // lower ghost rows at depth (v, depth]
        ghost.hi[k] = upper >= 0 && any ? base + (size_t)(H + n + v) * row_bytes : nullptr;       // upper ghost rows at depth (v, depth]
    }
    if (total == 0) return FS_OK;
    if (total > cm->stage_part) {
        FS_HIP(hipStreamSynchronize(xs));
        if (cm->stage) { FS_HIP(hipFree(cm->stage)); cm->stage = nullptr; cm->stage_part = 0; }
        const size_t part = (total + 4095) / 4096 * 4096 * 2;     // headroom: more / deeper fields may follow
        FS_HIP(hipMalloc(&cm->stage, 4 * part));
        cm->stage_part = part;
    }
    char *send_lo = cm->stage, *send_hi = cm->stage + cm->stage_part, *recv_lo = cm->stage + 2 * cm->stage_part, *recv_hi = cm->stage + 3 * cm->stage_part;
    const dim3 grid(64, 2 * nfields);
    FS_KLAUNCH(k_halo_pack<true>, grid, dim3(256), 0, xs, own, send_lo, send_hi);
    FS_NCCL(g_rccl.GroupStart());
    if (lower >= 0) {
        FS_NCCL(g_rccl.Send(send_lo, total, ncclUint8, lower, cm->comm, xs));
        FS_NCCL(g_rccl.Recv(recv_lo, total, ncclUint8, lower, cm->comm, xs));
    }
    if (upper >= 0) {
        FS_NCCL(g_rccl.Send(send_hi, total, ncclUint8, upper, cm->comm, xs));
        FS_NCCL(g_rccl.Recv(recv_hi, total, ncclUint8, upper, cm->comm, xs));
    }
    FS_NCCL(g_rccl.GroupEnd());
    FS_KLAUNCH(k_halo_pack<false>, grid, dim3(256), 0, xs, ghost, recv_lo, recv_hi);
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// Refresh `depth` ghost rows on each side of the owned rows [halo, halo + nyl).
//   to the lower neighbour: my first `depth` owned rows  -> their upper ghost rows
//   to the upper neighbour: my last  `depth` owned rows  -> their lower ghost rows
// `lower` / `upper` are the peer ranks (-1 = domain edge, no neighbour).  One field goes straight from / to its rows
// (a ghost-row block is contiguous in the [row][channel][x] layout); several fields travel as one packed message.
static int exchange(fs_ctx *ctx, hipStream_t xs, fs_field *const *fields, const int *valid, int nfields, int depth, int lower, int upper)
{
    if (nfields >= 2 && nfields <= MAX_PACK) return exchange_packed(ctx, xs, fields, valid, nfields, depth, lower, upper);
    return exchange_direct(ctx, xs, fields, valid, nfields, depth, lower, upper);
}

static int check_exchange_args(fs_ctx *ctx, fs_field *const *fields, const int *valid, int nfields, int depth)
{
    FS_REQUIRE(ctx && fields && nfields >= 0, "null argument");
    if (valid) for (int n = 0; n < nfields; ++n) FS_REQUIRE(valid[n] >= 0, "negative valid-row count");
    FS_REQUIRE(depth >= 0 && depth <= ctx->halo && depth <= ctx->nyl, "halo depth exceeds the slab's ghost rows or owned rows");
    for (int n = 0; n < nfields; ++n) FS_REQUIRE(fields[n] && fields[n]->ctx == ctx, "null / foreign field");
    return FS_OK;
}

static int begin(fs_ctx *ctx, fs_field *const *fields, const int *valid, int nfields, int depth, bool self)
{
    int rc = check_exchange_args(ctx, fields, valid, nfields, depth); if (rc) return rc;
    Comm *cm = ctx->comm;
    if (!cm) { set_error("fs_halo_exchange without fs_comm_init"); return FS_ERR_COMM; }
    if (ctx->tape_rec) {      // recorded like a kernel launch (fs_tape_*): the closure owns copies of the argument arrays
        std::vector<fs_field *> fv(fields, fields + nfields);
        std::vector<int> vv;
        if (valid) vv.assign(valid, valid + nfields);
        Tape *t = ctx->tape_rec;
        ctx->tape_rec = nullptr;                       // the closure re-enters begin() at replay time with no tape open
        t->ops.emplace_back([=]() -> int { return begin(ctx, fv.data(), vv.empty() ? nullptr : vv.data(), (int)fv.size(), depth, self); });
        ctx->tape_rec = t;
        if (!ctx->tape_execute) return FS_OK;
        ctx->tape_rec = nullptr;
        rc = begin(ctx, fields, valid, nfields, depth, self);
        ctx->tape_rec = t;
        return rc;
    }
    FS_REQUIRE(!cm->in_flight, "fs_halo_exchange_begin while another exchange is in flight (call fs_halo_exchange_wait first)");
    FS_REQUIRE(!ctx->capturing, "halo exchange during graph capture");
    if (self && cm->nranks != 1) { set_error("loop-back exchange needs a 1-rank communicator"); return FS_ERR_COMM; }
    int lower = cm->rank > 0 ? cm->rank - 1 : -1, upper = cm->rank < cm->nranks - 1 ? cm->rank + 1 : -1;
    if (self || (cm->loopback && cm->nranks == 1)) lower = upper = 0;
    if (depth == 0 || nfields == 0 || (lower < 0 && upper < 0)) { cm->in_flight = true; cm->armed = false; cm->marked = false; return FS_OK; }
    if (!cm->own_stream) {      // in line on the compute stream: ordered by the stream itself, nothing to wait for later
        cm->marked = false;
        const ProfRec span = prof_span_begin(ctx, "halo_exchange", ctx->stream);       // pack -> grouped send / recv -> unpack, as one span
        rc = exchange(ctx, ctx->stream, fields, valid, nfields, depth, lower, upper);
        prof_span_end(ctx, span, ctx->stream);
        if (rc) return rc;
        cm->in_flight = true;
        cm->armed = false;
        return FS_OK;
    }
    if (!cm->marked) FS_HIP(hipEventRecord(cm->ev_compute, ctx->stream));   // everything queued so far produces the rows we send
    cm->marked = false;
    FS_HIP(hipStreamWaitEvent(cm->stream, cm->ev_compute, 0));
    const ProfRec span = prof_span_begin(ctx, "halo_exchange", cm->stream);
    rc = exchange(ctx, cm->stream, fields, valid, nfields, depth, lower, upper);
    prof_span_end(ctx, span, cm->stream);
    if (rc) return rc;
    FS_HIP(hipEventRecord(cm->ev_comm, cm->stream));
    cm->in_flight = true;
    cm->armed = true;
    return FS_OK;
}

int fs_halo_exchange_begin(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth) { return begin(ctx, fields, nullptr, nfields, depth, false); }

// As begin(), but field k's ghost rows are already correct to depth valid_rows[k]: only the rows beyond travel
// (the validity tracker of fs/runtime.py knows these numbers; they are the same on both sides of a slab boundary).
int fs_halo_exchange_begin_partial(fs_ctx *ctx, fs_field *const *fields, const int *valid_rows, int nfields, int depth)
{ return begin(ctx, fields, valid_rows, nfields, depth, false); }

// Where the exchanges of this context run from now on: 0 - in line on the compute stream (the default), 1 - on the communication stream (what
// lets them overlap with kernels; FS_OVERLAP=1 sets it at fs_comm_init).  Both give the same bits.  bench.py times both on the first
// steps of an N > 1 run and keeps the faster.  Not while an exchange is in flight or a tape is being recorded.
int fs_comm_set_overlap(fs_ctx *ctx, int on)
{
    FS_REQUIRE(ctx, "ctx is null");
    Comm *cm = ctx->comm;
    if (!cm) { set_error("fs_comm_set_overlap without fs_comm_init"); return FS_ERR_COMM; }
    FS_REQUIRE(!cm->in_flight && !ctx->tape_rec && !ctx->capturing, "fs_comm_set_overlap while an exchange is in flight / a tape is being recorded");
    FS_HIP(hipStreamSynchronize(ctx->stream));
    FS_HIP(hipStreamSynchronize(cm->stream));
    cm->own_stream = on != 0;
    cm->marked = false;
    return FS_OK;
}

// Optional, before begin(): fix the point of the compute stream the exchange depends on NOW, so that kernels launched between
// mark() and begin() (the interior rows) are already running while the host is still issuing the exchange.
int fs_halo_exchange_mark(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    Comm *cm = ctx->comm;
    if (!cm) { set_error("fs_halo_exchange without fs_comm_init"); return FS_ERR_COMM; }
    if (ctx->tape_rec) {
        ctx->tape_rec->ops.emplace_back([=]() -> int { return fs_halo_exchange_mark(ctx); });     // replay runs with no tape open
        if (!ctx->tape_execute) return FS_OK;
    }
    FS_REQUIRE(!cm->in_flight, "fs_halo_exchange_mark while an exchange is in flight");
    if (!cm->own_stream) return FS_OK;
    FS_HIP(hipEventRecord(cm->ev_compute, ctx->stream));
    cm->marked = true;
    return FS_OK;
}

int fs_halo_exchange_wait(fs_ctx *ctx)
{
    FS_REQUIRE(ctx, "ctx is null");
    Comm *cm = ctx->comm;
    if (ctx->tape_rec && cm) {
        ctx->tape_rec->ops.emplace_back([=]() -> int { return fs_halo_exchange_wait(ctx); });
        if (!ctx->tape_execute) return FS_OK;
    }
    if (!cm || !cm->in_flight) return FS_OK;
    cm->in_flight = false;
    if (cm->armed) FS_HIP(hipStreamWaitEvent(ctx->stream, cm->ev_comm, 0));   // later compute-stream work sees the filled ghost rows
    cm->armed = false;
    return FS_OK;
}

int fs_halo_exchange_multi(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth)
{
    int rc = begin(ctx, fields, nullptr, nfields, depth, false);
    return rc ? rc : fs_halo_exchange_wait(ctx);
}

// Loop-back self-test of the same code on a 1-rank communicator: the rank is its own lower AND upper neighbour.  RCCL matches
// the sends and receives of one peer in issue order, so afterwards   lower ghost rows == first owned rows   and
// upper ghost rows == last owned rows.  Lets a single-GPU box check row offsets, counts, dtype and stream order of the RCCL leg.
int fs_halo_exchange_self(fs_ctx *ctx, fs_field *const *fields, int nfields, int depth)
{
    int rc = begin(ctx, fields, nullptr, nfields, depth, true);
    return rc ? rc : fs_halo_exchange_wait(ctx);
}

// 1-rank communicators only: from now on every exchange (begin / multi) treats the rank as its own neighbour on both sides, so a
// whole slab simulation can run through the real RCCL path on one GPU (results are those of a domain whose ghost rows mirror its
// own edge rows - meaningless physically, but deterministic: blocking and overlapped exchanges must give the same bits).
int fs_comm_loopback(fs_ctx *ctx, int on)
{
    FS_REQUIRE(ctx && ctx->comm, "no communicator");
    FS_REQUIRE(ctx->comm->nranks == 1, "loop-back needs a 1-rank communicator");
    ctx->comm->loopback = on != 0;
    return FS_OK;
}

int fs_halo_exchange(fs_ctx *ctx, fs_field *f, int depth) { return fs_halo_exchange_multi(ctx, &f, 1, depth); }

int fs_allreduce_sum(fs_ctx *ctx, double *values, int n)
{
    FS_REQUIRE(ctx && values && n >= 0 && n <= 16, "bad argument (n <= 16)");
    Comm *cm = ctx->comm;
    if (!cm || cm->nranks == 1 || n == 0) return FS_OK;
    FS_REQUIRE(!cm->in_flight, "fs_allreduce_sum while a halo exchange is in flight");
    FS_REQUIRE(!ctx->tape_rec, "fs_allreduce_sum while recording a tape");
    FS_HIP(hipStreamSynchronize(ctx->stream));                      // host-side collective: the compute stream drains first
    FS_HIP(hipMemcpyAsync(cm->d_red, values, n * sizeof(double), hipMemcpyHostToDevice, cm->stream));
    FS_NCCL(g_rccl.AllReduce(cm->d_red, cm->d_red, n, ncclFloat64, ncclSum, cm->comm, cm->stream));
    FS_HIP(hipMemcpyAsync(values, cm->d_red, n * sizeof(double), hipMemcpyDeviceToHost, cm->stream));
    FS_HIP(hipStreamSynchronize(cm->stream));
    return FS_OK;
}

}  // extern "C"
