// gf_dist.hip -- data parallelism behind the C ABI: one RCCL communicator per context (= per GPU, per process or per
// worker thread), used for the ONE exchange the SMP path has -- the sum of the flat parameter-gradient buffer over ranks --
// and for the parameter broadcast that precedes it in the reference.
//
// Reference analogue: SMP_omega::Threaded_BatchLearn (GraphFlow/SMP_omega.h:750-792): the master model's parameter values
// are copied into every worker clone (copy_value, :710-728, :771-773), the workers run forward/backward on their slice of
// the batch, and the master adds every clone's gradients serially (add_gradient, :730-740, :784-786).  Here a worker is a
// GPU, the copy is gf_dist_broadcast_f32 and the serial add loop is gf_dist_allreduce_sum_f32 (ring/tree over xGMI).
//
// RCCL is bound at run time, once per process and never unloaded (dlopen of librccl.so.1, the SONAME both the ROCm install and PyTorch's bundled copy carry, so
// a process that already has torch's RCCL mapped shares that instance): libgf_hip.so keeps loading on boxes and in programs
// that never go multi-GPU, and fails loudly -- GF_ERR_UNSUPPORTED with the dlerror text -- when RCCL is asked for and absent.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include <rccl/rccl.h>

#include "gf_internal.h"

// RCCL's entry points, resolved ONCE per process and never unloaded: ncclGetUniqueId starts the bootstrap root's listener thread
// inside librccl and a communicator owns proxy threads, so a dlclose that drops the last reference would unmap code under live
// threads (or let a later dlopen start from a fresh image without the root's state) in any program that does not also hold
// torch's copy of the library.  RTLD_NODELETE pins the image; the table is shared by every context of the process.
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[320] = {0};  // why the binding failed (empty when it succeeded)
};

struct gf_dist_state {
    const RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;  // the collectives' own stream (overlaps the rest of the reverse sweep)
    // watchdog bookkeeping: what the last collective handed to RCCL was (a rank stuck behind a peer that never arrives says which
    // exchange it is waiting in instead of hanging the job: gf_dist_quiesce, the join of gf_smp_backward)
    unsigned long long issued = 0;
    char stage[96] = {0};
    // Set when a watchdog limit fired: a collective that will never complete is (or may be) queued on `stream` and, through the
    // sweep's join, in front of everything later on the context's stream.  From then on nothing may wait for those streams without a
    // bound: teardown ABORTS the communicator (ncclCommAbort makes RCCL's kernels leave their spin loops) instead of draining it.
    bool poisoned = false;
};

namespace gf {
namespace {

const RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {std::getenv("GF_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        char err[256] = {0};
        for (const char *n : names) {
            if (!n || !n[0]) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
            if (api.lib) break;
            std::snprintf(err, sizeof err, "%s", dlerror());
        }
        if (!api.lib) {
            std::snprintf(api.why, sizeof api.why, "RCCL is not available (dlopen librccl.so.1: %s); multi-GPU needs it", err);
            return;
        }
#define GF_BIND(field, sym)                                                                \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym));               \
    if (!api.field && !api.why[0]) std::snprintf(api.why, sizeof api.why, "RCCL library lacks %s", sym);
        GF_BIND(GetUniqueId, "ncclGetUniqueId")
        GF_BIND(CommInitRank, "ncclCommInitRank")
        GF_BIND(CommDestroy, "ncclCommDestroy")
        GF_BIND(CommAbort, "ncclCommAbort")
        GF_BIND(AllReduce, "ncclAllReduce")
        GF_BIND(Broadcast, "ncclBroadcast")
        GF_BIND(GetErrorString, "ncclGetErrorString")
#undef GF_BIND
    });
    return &api;
}

// the process-wide table, or GF_ERR_UNSUPPORTED with the reason
gf_status bind_rccl(gf_ctx *ctx, const RcclApi **out) {
    const RcclApi *a = rccl_api();
    if (a->why[0]) return fail(ctx, GF_ERR_UNSUPPORTED, "%s", a->why);
    *out = a;
    return GF_OK;
}

#define GF_NCCL_TRY(ctx, d, expr)                                                                                        \
    do {                                                                                                                 \
        ncclResult_t r__ = (expr);                                                                                       \
        if (r__ != ncclSuccess) return gf::fail((ctx), GF_ERR_HIP, "%s failed: %s", #expr, (d)->api->GetErrorString(r__));    \
    } while (0)

}  // namespace

bool dist_active(const gf_ctx *ctx) { return ctx && ctx->dist && ctx->dist->comm; }
hipStream_t dist_stream(gf_ctx *ctx) { return dist_active(ctx) ? ctx->dist->stream : nullptr; }

// seconds a rank waits for its peers before it gives up with GF_ERR_TIMEOUT (GF_DIST_TIMEOUT_S; 0 = wait for ever, as RCCL does)
double dist_timeout_s() {
    const char *e = std::getenv("GF_DIST_TIMEOUT_S");
    if (e && e[0]) {
        const double v = std::atof(e);
        return v < 0.0 ? 0.0 : v;
    }
    return 1800.0;   // (skew between ranks counts: a rank-0 validation pass or checkpoint between two steps must fit inside the limit)
}

bool dist_poisoned(const gf_ctx *ctx) { return ctx && ctx->dist && ctx->dist->poisoned; }

gf_status dist_allreduce_on(gf_ctx *ctx, float *buf, size_t n, hipStream_t stream, const char *what) {
    if (!dist_active(ctx)) return fail(ctx, GF_ERR_INVALID, "gf_dist: no communicator on this context (gf_dist_init)");
    if (n == 0) return GF_OK;
    gf_dist_state *d = ctx->dist;
    d->issued += 1;
    std::snprintf(d->stage, sizeof d->stage, "all-reduce #%llu of %zu floats (%s)", d->issued, n, what ? what : "gf_dist_allreduce_sum_f32");
    LaunchTimer lt(ctx, "rccl_allreduce", stream);   // (HIP events on the collective's own stream when the context's timing is on)
    GF_NCCL_TRY(ctx, d, d->api->AllReduce(buf, buf, n, ncclFloat32, ncclSum, d->comm, stream));
    lt.done();
    return GF_OK;
}

// Bounded wait for `ev` (recorded behind collectives): polls instead of blocking, so that a peer that never joins the exchange turns
// into an error that names this rank, the world and the exchange, not into a job that hangs until its scheduler kills it.
gf_status dist_wait_event(gf_ctx *ctx, hipEvent_t ev, const char *where) {
    if (!dist_active(ctx) || !ev) return GF_OK;
    if (ctx->dist->poisoned)
        return fail(ctx, GF_ERR_TIMEOUT, "gf_dist: %s: an earlier wait on this communicator timed out; gf_dist_finalize (which aborts it) is the only way on", where);
    const double limit = dist_timeout_s();
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return GF_OK;
        if (e != hipErrorNotReady) return fail(ctx, GF_ERR_HIP, "gf_dist: %s: %s", where, hipGetErrorString(e));
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (limit > 0.0 && waited > limit) {
            gf_dist_state *d = ctx->dist;
            d->poisoned = true;   // (gf_dist_finalize / gf_ctx_destroy abort the communicator instead of waiting for it)
            return fail(ctx, GF_ERR_TIMEOUT, "gf_dist: rank %d of %d (device %d) waited %.0f s in %s; last collective handed to RCCL: %s "
                                              "-- a peer rank never joined it (GF_DIST_TIMEOUT_S sets the limit, 0 = none)",
                        d->rank, d->world, ctx->device, waited, where, d->stage[0] ? d->stage : "none");
        }
        if (++spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

void dist_teardown(gf_ctx *ctx) {
    gf_dist_state *d = ctx->dist;
    if (!d) return;
    if (d->poisoned) {
        // a collective whose peer never came is still queued: ncclCommAbort sets the communicator's abort flag (its kernels return,
        // the streams behind them drain) and frees it without the handshake with the peers that ncclCommDestroy performs
        if (d->comm) (void)d->api->CommAbort(d->comm);
        if (d->stream) {   // bounded: the aborted kernels leave within milliseconds; if the stream still does not drain, leak it
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e;
            while ((e = hipStreamQuery(d->stream)) == hipErrorNotReady &&
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0)
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (e != hipErrorNotReady) (void)hipStreamDestroy(d->stream);
        }
    } else {
        if (d->stream) (void)hipStreamSynchronize(d->stream);
        if (d->comm) (void)d->api->CommDestroy(d->comm);
        if (d->stream) (void)hipStreamDestroy(d->stream);
    }
    delete d;
    ctx->dist = nullptr;
}

}  // namespace gf

using gf::fail;

extern "C" {

gf_status gf_dist_unique_id(gf_ctx *ctx, void *id_out) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!id_out) return fail(ctx, GF_ERR_INVALID, "gf_dist_unique_id: null argument");
    static_assert(GF_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "GF_DIST_ID_BYTES must equal RCCL's unique-id size");
    const RcclApi *api = nullptr;
    gf_status st = gf::bind_rccl(ctx, &api);
    if (st != GF_OK) return st;
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(ctx, GF_ERR_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
    std::memcpy(id_out, id.internal, GF_DIST_ID_BYTES);
    return GF_OK;
}

gf_status gf_dist_init(gf_ctx *ctx, const void *id, int rank, int world) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(ctx, GF_ERR_INVALID, "gf_dist_init: bad argument (rank %d of %d)", rank, world);
    if (ctx->dist) return fail(ctx, GF_ERR_INVALID, "gf_dist_init: the context already has a communicator");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    gf_dist_state *d = new gf_dist_state();
    gf_status st = gf::bind_rccl(ctx, &d->api);
    if (st != GF_OK) {
        delete d;
        return st;
    }
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, GF_DIST_ID_BYTES);
    // ncclCommInitRank returns when EVERY rank of the world has called it.  It runs on a helper thread so that a rank whose peers
    // never arrive (a crashed process, a wrong world size, a rank left on another id) fails after GF_DIST_TIMEOUT_S with its rank and
    // world in gf_last_error instead of blocking for ever; on a timeout the helper is left behind (it owns its state; should
    // the peers arrive after all it aborts the communicator it then gets, nothing leaks) and the context stays without a communicator.
    // (ncclCommInitRankConfig with blocking = 0 would hand out an abortable handle at once, but it turns every later collective of the
    // communicator into an ncclInProgress call whose kernel is enqueued by a proxy thread some time after the call returns -- the
    // sweep's stream-ordered join would have to poll ncclCommGetAsyncError per segment; not worth it for the start-up path.)
    struct InitJob {
        std::mutex m;
        std::condition_variable cv;
        bool done = false, abandoned = false;
        ncclResult_t r = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    std::shared_ptr<InitJob> job = std::make_shared<InitJob>();
    const RcclApi *api = d->api;
    const int device = ctx->device;
    std::thread([job, api, uid, rank, world, device]() {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = api->CommInitRank(&c, world, uid, rank);
        std::lock_guard<std::mutex> g(job->m);
        if (job->abandoned) {   // the caller gave up (GF_ERR_TIMEOUT) and the peers came after all: nobody will ever own this communicator
            if (r == ncclSuccess && c) (void)api->CommAbort(c);
            return;
        }
        job->r = r;
        job->comm = c;
        job->done = true;
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(job->m);
        const double limit = gf::dist_timeout_s();
        const bool ok = limit > 0.0 ? job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; })
                                    : (job->cv.wait(lk, [&] { return job->done; }), true);
        if (!ok) {
            job->abandoned = true;   // (under job->m: the helper aborts its communicator itself should ncclCommInitRank ever return)
            st = fail(ctx, GF_ERR_TIMEOUT, "gf_dist_init: rank %d of %d (device %d) waited %.0f s in ncclCommInitRank -- not every rank of the world "
                                           "called gf_dist_init with this id (GF_DIST_TIMEOUT_S sets the limit, 0 = none)",
                      rank, world, ctx->device, limit);
            delete d;
            return st;
        }
    }
    d->comm = job->comm;
    const ncclResult_t r = job->r;
    if (r != ncclSuccess) {
        st = fail(ctx, GF_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device, d->api->GetErrorString(r));
        delete d;
        return st;
    }
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)d->api->CommDestroy(d->comm);
        delete d;
        return fail(ctx, GF_ERR_HIP, "gf_dist_init: hipStreamCreate failed");
    }
    d->rank = rank;
    d->world = world;
    ctx->dist = d;
    return GF_OK;
}

gf_status gf_dist_finalize(gf_ctx *ctx) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    // (after a watchdog timeout the context's stream waits on a join that will not come before the abort: abort first, drain after)
    if (!gf::dist_poisoned(ctx)) (void)hipStreamSynchronize(ctx->stream);
    gf::dist_teardown(ctx);
    return GF_OK;
}

int gf_dist_rank(const gf_ctx *ctx) { return gf::dist_active(ctx) ? ctx->dist->rank : 0; }
int gf_dist_world(const gf_ctx *ctx) { return gf::dist_active(ctx) ? ctx->dist->world : 1; }

gf_status gf_dist_allreduce_sum_f32(gf_ctx *ctx, float *buf, size_t n) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!buf && n) return fail(ctx, GF_ERR_INVALID, "gf_dist_allreduce_sum_f32: null buffer");
    return gf::dist_allreduce_on(ctx, buf, n, ctx->stream, nullptr);
}

gf_status gf_dist_quiesce(gf_ctx *ctx) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!gf::dist_active(ctx)) return GF_OK;
    // an event behind everything handed to the communicator's stream and behind the context's stream (collectives issued through
    // gf_dist_allreduce_sum_f32 run there), waited for with the watchdog's limit
    hipEvent_t ev = nullptr;
    GF_HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    gf_status st = GF_OK;
    hipStream_t streams[2] = {ctx->dist->stream, ctx->stream};
    for (int i = 0; i < 2 && st == GF_OK; ++i) {
        if (hipEventRecord(ev, streams[i]) != hipSuccess) {
            st = fail(ctx, GF_ERR_HIP, "gf_dist_quiesce: hipEventRecord failed");
            break;
        }
        st = gf::dist_wait_event(ctx, ev, i == 0 ? "gf_dist_quiesce (the communicator's stream)" : "gf_dist_quiesce (the context's stream)");
    }
    (void)hipEventDestroy(ev);
    return st;
}

gf_status gf_dist_broadcast_f32(gf_ctx *ctx, float *buf, size_t n, int root) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!gf::dist_active(ctx)) return fail(ctx, GF_ERR_INVALID, "gf_dist: no communicator on this context (gf_dist_init)");
    if (!buf && n) return fail(ctx, GF_ERR_INVALID, "gf_dist_broadcast_f32: null buffer");
    gf_dist_state *d = ctx->dist;
    if (root < 0 || root >= d->world) return fail(ctx, GF_ERR_INVALID, "gf_dist_broadcast_f32: root %d of %d", root, d->world);
    if (n == 0) return GF_OK;
    d->issued += 1;
    std::snprintf(d->stage, sizeof d->stage, "broadcast #%llu of %zu floats from rank %d", d->issued, n, root);
    GF_NCCL_TRY(ctx, d, d->api->Broadcast(buf, buf, n, ncclFloat32, root, d->comm, ctx->stream));
    return GF_OK;
}

}  // extern "C"
