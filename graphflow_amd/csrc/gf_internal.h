// gf_internal.h -- shared by the translation units of libgf_hip.so (not part of the public ABI).
#ifndef GF_INTERNAL_H_INCLUDED
#define GF_INTERNAL_H_INCLUDED

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "gf_hip.h"

struct gf_dist_state;  // gf_dist.hip: RCCL communicator of the context (gf_dist_init), null until then

struct gf_ctx {
    int device = 0;
    gf_dist_state *dist = nullptr;
    int fp32_products = 0;  // GF_OPT_SMP_FP32_PRODUCTS: the C = 64 level's block products on the fp32 matrix pipe
    int r18_generic = 0;  // GF_OPT_R18_GENERIC_KERNELS: route RisiContraction_18 through the generic kernels (parity tests)
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    void *ws = nullptr;  // device scratch, grown on demand (never inside a timed region if gf_ctx_reserve was called)
    size_t ws_bytes = 0;
    void *stage = nullptr;  // device staging for the host-pointer (mode A) entry points
    size_t stage_bytes = 0;
    void *repack = nullptr;  // device scratch of the contraction entry points: operands repacked to a multiple of four channels
    size_t repack_bytes = 0;
    void *pinned = nullptr;  // pinned host staging for mode A
    size_t pinned_bytes = 0;
    char err[512] = {0};
    // optional per-kernel timing (gf_ctx_set_timing): HIP events around every launch on ctx->stream
    bool timing = false;
    char timing_filter[64] = {0};  // non-empty: only launches whose name equals it are timed (gf_ctx_set_timing_filter)
    struct Timer {
        const char *name;
        double ms;
        long long launches;
    };
    struct Pending {
        int slot;
        hipEvent_t start, stop;
    };
    std::vector<Timer> timers;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
};

namespace gf {

gf_status fail(gf_ctx *ctx, gf_status st, const char *fmt, ...);
bool poison_buffers();  // GF_POISON=1 (gf_capi.hip)
gf_status ensure_ws(gf_ctx *ctx, size_t bytes);
gf_status ensure_stage(gf_ctx *ctx, size_t bytes);
gf_status ensure_repack(gf_ctx *ctx, size_t bytes);
gf_status ensure_pinned(gf_ctx *ctx, size_t bytes);
// kernels that want more than the default 32 KiB dynamic-LDS window opt in once per (device, kernel), process-wide, raise-only
gf_status opt_in_lds_fn(gf_ctx *ctx, const void *kernel, size_t bytes);
template <typename Kern>
gf_status opt_in_lds(gf_ctx *ctx, Kern kern, size_t bytes) {
    return opt_in_lds_fn(ctx, reinterpret_cast<const void *>(kern), bytes);
}
// data-parallel hooks (gf_dist.hip): is a communicator attached, and the collective on an explicit stream
bool dist_active(const gf_ctx *ctx);
bool dist_poisoned(const gf_ctx *ctx);   // a watchdog limit fired on this context's communicator: never wait for its streams unbounded
gf_status dist_allreduce_on(gf_ctx *ctx, float *buf, size_t n, hipStream_t stream, const char *what = nullptr);
// bounded wait (GF_DIST_TIMEOUT_S) for an event recorded behind collectives: GF_ERR_TIMEOUT names the rank, the world and the exchange
gf_status dist_wait_event(gf_ctx *ctx, hipEvent_t ev, const char *where);
hipStream_t dist_stream(gf_ctx *ctx);

// Streaming (non-temporal) accesses of the SMP level's kernels, selectable per site for A/B builds (-DGF_NT_SITES=<mask>):
//   1 products: operand loads   2 products: stores   4 weight gradients: A-operand (T) loads   256: their B-operand loads   8 combine: O / f / df loads   16 combine: stores
//   32 tables-forward: T stores   64 consumer gather: df stores   128 consumer gather: the S_ab / T6 gradient blocks (read once)
// tools/micro/copy_probe.hip: `nt` on both sides of a streaming kernel is worth 8 - 10 % of a copy's rate.  Measured per site on the
// cfg3 step (round 6, two passes each, one box: no `nt` 6.83 - 6.85 ms): 8 + 16 -> 6.73 - 6.75, 32 -> 6.74 - 6.75 (its readers gain
// as much as tables-forward itself: the dirty lines of a plain store are written back while the NEXT kernel runs), 64 ~ 0.  NOT
// everywhere: 1 doubles the product kernels' time (a lane reads its 128-byte row segment as eight 16-byte requests -- with `nt`
// the line is gone again before the second one), 2 and 256 cost 1 - 3 % of their kernels; 4 alone (a whole 1 KB row of T per request)
// -0.01 ms.  Default: 4 | 8 | 16 | 32 | 64.
#ifndef GF_NT_SITES
#define GF_NT_SITES 124
#endif
#if defined(__HIPCC__)
template <int SITE, typename V>
__device__ __forceinline__ V gf_ld_s(const V *p) {
    if constexpr ((GF_NT_SITES & SITE) != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int SITE, typename V>
__device__ __forceinline__ void gf_st_s(V *p, V v) {
    if constexpr ((GF_NT_SITES & SITE) != 0) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#endif

#define GF_HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return gf::fail((ctx), GF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                   \
    } while (0)

#define GF_LAUNCH_CHECK(ctx, what)                                                                  \
    do {                                                                                            \
        hipError_t e__ = hipGetLastError();                                                         \
        if (e__ != hipSuccess)                                                                      \
            return gf::fail((ctx), GF_ERR_HIP, "launch of %s failed: %s", (what), hipGetErrorString(e__)); \
    } while (0)

// Brackets one kernel launch with HIP events when ctx->timing is on (otherwise free).
struct LaunchTimer {
    gf_ctx *ctx;
    int slot = -1;
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream = nullptr;   // where the bracketed work runs (default: the context's stream)
    LaunchTimer(gf_ctx *c, const char *name, hipStream_t on = nullptr);
    void done();
};
gf_status resolve_timers(gf_ctx *ctx);

#define GF_LAUNCH(ctx, name, kern, grid, block, lds, ...)                                  \
    do {                                                                                   \
        gf::LaunchTimer lt__((ctx), (name));                                               \
        hipLaunchKernelGGL(kern, grid, block, lds, (ctx)->stream, __VA_ARGS__);           \
        lt__.done();                                                                       \
        GF_LAUNCH_CHECK((ctx), (name));                                                    \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- contraction back ends (device pointers, stream = ctx->stream) --------------------------------------------
// r18: hand-specialised slab kernels (contract18.hip); fall back to the generic table kernels when the shape is
// outside the fast path.
gf_status r18_forward(gf_ctx *ctx, const float *P, const float *A, float *Out, int N, int C, int batch);
gf_status r18_backward(gf_ctx *ctx, const float *G, const float *A, float *dP, int N, int C, int batch, int accumulate);
size_t r18_workspace_bytes(int N, int C, int batch);
// ragged batches: device-resident node tables (see contract18.hip "Ragged")
struct gf_ragged_nodes {
    const int *pair_node;
    const int *node_s;
    const long long *node_p, *node_row, *node_pair;
    long long total_rows, total_pairs;
};
bool r18_ragged_supported(int smax, int C, const void *p0, const void *p1);
size_t r18_ragged_workspace_bytes(long long total_rows, long long total_pairs, int C);
gf_status r18_forward_ragged(gf_ctx *ctx, const float *P, const float *A, float *Out, const gf_ragged_nodes &t,
                             long long pair_lo, long long pair_hi, int smax, int C);
gf_status r18_backward_ragged(gf_ctx *ctx, const float *G, const float *A, float *dP, const gf_ragged_nodes &t,
                              long long pair_lo, long long pair_hi, int smax, int C, int accumulate);
// r4 / r10 / r50: table kernels (contract_families.hip)
gf_status family_forward(gf_ctx *ctx, int K, const float *P, const float *A, float *Out, int N, int C, int batch);
gf_status family_backward(gf_ctx *ctx, int K, const float *G, const float *A, float *dP, int N, int C, int batch,
                          int accumulate);
size_t family_workspace_bytes(int K, int N, int C, int batch);

}  // namespace gf
#endif
