// gf_capi.hip -- the extern "C" surface declared in include/gf_hip.h: context, error reporting, family dispatch
// and the host-pointer ("mode A") staging wrappers used by the Entity-style op classes.
#include <cstring>

#include "gf_internal.h"

namespace gf {

static thread_local char g_create_err[512] = {0};

gf_status fail(gf_ctx *ctx, gf_status st, const char *fmt, ...) {
    char *dst = ctx ? ctx->err : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return st;
}

static gf_status grow(gf_ctx *ctx, void **buf, size_t *have, size_t want, bool pinned_host) {
    if (want <= *have) return GF_OK;
    // Growing means the old scratch may still be read by kernels in flight on the stream.
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*buf) {
        if (pinned_host)
            GF_HIP_TRY(ctx, hipHostFree(*buf));
        else
            GF_HIP_TRY(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    want = align_up(want, 1 << 20);
    hipError_t e = pinned_host ? hipHostMalloc(buf, want, hipHostMallocDefault) : hipMalloc(buf, want);
    if (e != hipSuccess) {
        *buf = nullptr;
        return fail(ctx, GF_ERR_NOMEM, "allocating %zu bytes of %s failed: %s", want,
                    pinned_host ? "pinned host memory" : "device memory", hipGetErrorString(e));
    }
    *have = want;
    return GF_OK;
}

gf_status ensure_ws(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->ws, &ctx->ws_bytes, bytes, false); }
gf_status ensure_stage(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->stage, &ctx->stage_bytes, bytes, false); }
gf_status ensure_pinned(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->pinned, &ctx->pinned_bytes, bytes, true); }

void r18_force_generic(int on);

LaunchTimer::LaunchTimer(gf_ctx *c, const char *name) : ctx(c) {
    if (!c->timing) return;
    for (size_t i = 0; i < c->timers.size(); ++i)
        if (c->timers[i].name == name || std::strcmp(c->timers[i].name, name) == 0) slot = (int)i;
    if (slot < 0) {
        c->timers.push_back({name, 0.0, 0});
        slot = (int)c->timers.size() - 1;
    }
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
        if (!c->event_pool.empty()) {
            ev[i] = c->event_pool.back();
            c->event_pool.pop_back();
        } else if (hipEventCreate(&ev[i]) != hipSuccess) {
            slot = -1;
            return;
        }
    }
    start = ev[0];
    stop = ev[1];
    (void)hipEventRecord(start, c->stream);
}

void LaunchTimer::done() {
    if (slot < 0) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->pending.push_back({slot, start, stop});
}

gf_status resolve_timers(gf_ctx *ctx) {
    if (ctx->pending.empty()) return GF_OK;
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (const auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            ctx->timers[p.slot].ms += ms;
            ctx->timers[p.slot].launches += 1;
        }
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->pending.clear();
    return GF_OK;
}

namespace {

bool known_family(int K) { return K == 4 || K == 10 || K == 18 || K == 50; }

gf_status check_contract_args(gf_ctx *ctx, int K, const void *in, const void *A, const void *out, int N, int C,
                              int batch) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!known_family(K)) return fail(ctx, GF_ERR_INVALID, "unknown contraction family K=%d (expected 4, 10, 18 or 50)", K);
    if (N <= 0 || C <= 0 || batch < 0)
        return fail(ctx, GF_ERR_INVALID, "bad shape N=%d C=%d batch=%d", N, C, batch);
    if (batch > 0 && (!in || !out || (K != 4 && !A))) return fail(ctx, GF_ERR_INVALID, "null tensor pointer");
    if ((size_t)batch * N > 0x7fffffffu / 4) return fail(ctx, GF_ERR_UNSUPPORTED, "batch*N too large for one launch");
    return GF_OK;
}

template <typename T>
__global__ void cast_to_f32(const T *__restrict__ in, float *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
template <typename T>
__global__ void cast_from_f32(const float *__restrict__ in, T *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (T)in[i];
}
unsigned cast_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

// Host-pointer forward: gather the N neighbour tensors into pinned memory (this IS StackTensor3D::forward,
// StackTensor3D.h:54-73, done during staging), one H2D, cast to f32 on device, kernels, cast back, one D2H.
template <typename T>
gf_status contract_forward_host(gf_ctx *ctx, int K, const T *const *tensors, const T *A, T *out_value, int N, int C) {
    gf_status st = check_contract_args(ctx, K, tensors, A, out_value, N, C, 1);
    if (st != GF_OK) return st;
    const size_t nP = (size_t)N * N * N * C, nA = (K == 4) ? 0 : (size_t)N * N, nO = (size_t)N * N * K * C;
    const size_t per = (size_t)N * N * C;
    // pinned: [P | A | Out] in T;  device stage: the same in T, then [P | A | Out] in f32
    const size_t tBytes = sizeof(T) * (nP + nA + nO);
    st = ensure_pinned(ctx, tBytes);
    if (st != GF_OK) return st;
    st = ensure_stage(ctx, align_up(tBytes, 256) + sizeof(float) * (nP + nA + nO) + 256);
    if (st != GF_OK) return st;
    const size_t oOut = (nP + nA + 3) & ~(size_t)3;  // keep Out 16-byte aligned in the f32 image whatever N*N is
    T *hP = static_cast<T *>(ctx->pinned), *hA = hP + nP, *hO = hA + nA;
    for (int a = 0; a < N; ++a) {
        if (!tensors[a]) return fail(ctx, GF_ERR_INVALID, "tensors[%d] is null", a);
        std::memcpy(hP + a * per, tensors[a], sizeof(T) * per);
    }
    if (nA) std::memcpy(hA, A, sizeof(T) * nA);
    T *dT = static_cast<T *>(ctx->stage);
    float *dF = reinterpret_cast<float *>(static_cast<char *>(ctx->stage) + align_up(tBytes, 256));
    GF_HIP_TRY(ctx, hipMemcpyAsync(dT, hP, sizeof(T) * (nP + nA), hipMemcpyHostToDevice, ctx->stream));
    GF_LAUNCH(ctx, "cast_to_f32", cast_to_f32<T>, dim3(cast_grid(nP + nA)), dim3(256), 0, dT, dF, nP + nA);
    st = gf_contract_forward_f32(ctx, K, dF, nA ? dF + nP : nullptr, dF + oOut, N, C, 1);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "cast_from_f32", cast_from_f32<T>, dim3(cast_grid(nO)), dim3(256), 0, dF + oOut, dT + nP + nA, nO);
    GF_HIP_TRY(ctx, hipMemcpyAsync(hO, dT + nP + nA, sizeof(T) * nO, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out_value, hO, sizeof(T) * nO);
    return GF_OK;
}

// Host-pointer backward: the vjp is computed write-only on the device and added into the N gradient buffers on the
// host (`tensors[a]->gradient[...] += ...`, RisiContraction_18.h:345-560) -- so the inputs' old gradients never
// cross PCIe.
template <typename T>
gf_status contract_backward_host(gf_ctx *ctx, int K, const T *out_gradient, const T *A, T *const *grads, int N, int C) {
    gf_status st = check_contract_args(ctx, K, out_gradient, A, grads, N, C, 1);
    if (st != GF_OK) return st;
    const size_t nP = (size_t)N * N * N * C, nA = (K == 4) ? 0 : (size_t)N * N, nO = (size_t)N * N * K * C;
    const size_t per = (size_t)N * N * C;
    const size_t tBytes = sizeof(T) * (nO + nA + nP);
    st = ensure_pinned(ctx, tBytes);
    if (st != GF_OK) return st;
    st = ensure_stage(ctx, align_up(tBytes, 256) + sizeof(float) * (nO + nA + nP) + 256);
    if (st != GF_OK) return st;
    T *hG = static_cast<T *>(ctx->pinned), *hA = hG + nO, *hD = hA + nA;
    std::memcpy(hG, out_gradient, sizeof(T) * nO);
    if (nA) std::memcpy(hA, A, sizeof(T) * nA);
    T *dT = static_cast<T *>(ctx->stage);
    float *dF = reinterpret_cast<float *>(static_cast<char *>(ctx->stage) + align_up(tBytes, 256));
    // f32 layout on device: [dP | G | A] so that dP and G stay 16-byte aligned whatever N*N is
    float *fD = dF, *fG = dF + nP, *fA = fG + nO;
    GF_HIP_TRY(ctx, hipMemcpyAsync(dT, hG, sizeof(T) * (nO + nA), hipMemcpyHostToDevice, ctx->stream));
    GF_LAUNCH(ctx, "cast_to_f32", cast_to_f32<T>, dim3(cast_grid(nO + nA)), dim3(256), 0, dT, fG, nO + nA);
    st = gf_contract_backward_f32(ctx, K, fG, nA ? fA : nullptr, fD, N, C, 1, /*accumulate=*/0);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "cast_from_f32", cast_from_f32<T>, dim3(cast_grid(nP)), dim3(256), 0, fD, dT + nO + nA, nP);
    GF_HIP_TRY(ctx, hipMemcpyAsync(hD, dT + nO + nA, sizeof(T) * nP, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < N; ++a) {
        if (!grads[a]) return fail(ctx, GF_ERR_INVALID, "grads[%d] is null", a);
        T *dst = grads[a];
        const T *src = hD + a * per;
        for (size_t i = 0; i < per; ++i) dst[i] += src[i];
    }
    return GF_OK;
}

}  // namespace
}  // namespace gf

extern "C" {

const char *gf_version(void) { return "graphflow_amd 0.1 (gfx950)"; }

gf_status gf_ctx_create(gf_ctx **out, int device, void *stream) {
    if (!out) return gf::fail(nullptr, GF_ERR_INVALID, "gf_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return gf::fail(nullptr, GF_ERR_HIP, "no HIP device available (%s); the HIP path has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count) return gf::fail(nullptr, GF_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return gf::fail(nullptr, GF_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    gf_ctx *ctx = new gf_ctx();
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream);  // NULL = the device's default stream, like the reference's ops
    *out = ctx;
    return GF_OK;
}

gf_status gf_ctx_use_private_stream(gf_ctx *ctx) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->owns_stream) {
        GF_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->owns_stream = true;
    }
    return GF_OK;
}

gf_status gf_ctx_destroy(gf_ctx *ctx) {
    if (!ctx) return GF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)gf::resolve_timers(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GF_OK;
}

gf_status gf_ctx_set_stream(gf_ctx *ctx, void *stream) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->owns_stream) {
        GF_HIP_TRY(ctx, hipStreamDestroy(ctx->stream));
        ctx->owns_stream = false;
    }
    ctx->stream = static_cast<hipStream_t>(stream);  // NULL = the default stream
    return GF_OK;
}

void *gf_ctx_get_stream(gf_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

gf_status gf_ctx_synchronize(gf_ctx *ctx) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GF_OK;
}

gf_status gf_ctx_reserve(gf_ctx *ctx, size_t workspace_bytes) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    return gf::ensure_ws(ctx, workspace_bytes);
}

gf_status gf_ctx_set_timing(gf_ctx *ctx, int enable) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    gf_status st = gf::resolve_timers(ctx);
    if (st != GF_OK) return st;
    if (enable) ctx->timers.clear();
    ctx->timing = enable != 0;
    return GF_OK;
}

int gf_ctx_timing_count(gf_ctx *ctx) {
    if (!ctx) return 0;
    (void)gf::resolve_timers(ctx);
    return (int)ctx->timers.size();
}

gf_status gf_ctx_timing_get(gf_ctx *ctx, int index, const char **name, double *total_ms, long long *launches) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    gf_status st = gf::resolve_timers(ctx);
    if (st != GF_OK) return st;
    if (index < 0 || index >= (int)ctx->timers.size()) return gf::fail(ctx, GF_ERR_INVALID, "timer index %d out of range", index);
    if (name) *name = ctx->timers[index].name;
    if (total_ms) *total_ms = ctx->timers[index].ms;
    if (launches) *launches = ctx->timers[index].launches;
    return GF_OK;
}

const char *gf_last_error(gf_ctx *ctx) { return ctx ? ctx->err : gf::g_create_err; }

size_t gf_contract_workspace_bytes(int K, int N, int C, int batch) {
    if (N <= 0 || C <= 0 || batch <= 0) return 0;
    switch (K) {
        case 18: return gf::r18_workspace_bytes(N, C, batch);
        case 4: case 10: case 50: return gf::family_workspace_bytes(K, N, C, batch);
        default: return 0;
    }
}

gf_status gf_contract_forward_f32(gf_ctx *ctx, int K, const float *P, const float *A, float *Out, int N, int C,
                                  int batch) {
    gf_status st = gf::check_contract_args(ctx, K, P, A, Out, N, C, batch);
    if (st != GF_OK || batch == 0) return st;
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    switch (K) {
        case 18: return gf::r18_forward(ctx, P, A, Out, N, C, batch);
        default: return gf::family_forward(ctx, K, P, A, Out, N, C, batch);
    }
}

gf_status gf_contract_backward_f32(gf_ctx *ctx, int K, const float *G, const float *A, float *dP, int N, int C,
                                   int batch, int accumulate) {
    gf_status st = gf::check_contract_args(ctx, K, G, A, dP, N, C, batch);
    if (st != GF_OK || batch == 0) return st;
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    switch (K) {
        case 18: return gf::r18_backward(ctx, G, A, dP, N, C, batch, accumulate);
        default: return gf::family_backward(ctx, K, G, A, dP, N, C, batch, accumulate);
    }
}

gf_status gf_contract_forward_host_f64(gf_ctx *ctx, int K, const double *const *tensors, const double *A,
                                       double *out_value, int N, int C) {
    return gf::contract_forward_host<double>(ctx, K, tensors, A, out_value, N, C);
}
gf_status gf_contract_backward_host_f64(gf_ctx *ctx, int K, const double *out_gradient, const double *A,
                                        double *const *grads, int N, int C) {
    return gf::contract_backward_host<double>(ctx, K, out_gradient, A, grads, N, C);
}
gf_status gf_contract_forward_host_f32(gf_ctx *ctx, int K, const float *const *tensors, const float *A,
                                       float *out_value, int N, int C) {
    return gf::contract_forward_host<float>(ctx, K, tensors, A, out_value, N, C);
}
gf_status gf_contract_backward_host_f32(gf_ctx *ctx, int K, const float *out_gradient, const float *A,
                                        float *const *grads, int N, int C) {
    return gf::contract_backward_host<float>(ctx, K, out_gradient, A, grads, N, C);
}

/* test hook, not declared in the public header: force the generic (layout-agnostic) kernels */
void gf_debug_force_generic(int on) { gf::r18_force_generic(on); }

}  // extern "C"
