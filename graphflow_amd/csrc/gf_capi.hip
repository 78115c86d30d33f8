// gf_capi.hip -- the extern "C" surface declared in include/gf_hip.h: context, error reporting, family dispatch
// and the host-pointer ("mode A") staging wrappers used by the Entity-style op classes.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "gf_internal.h"

namespace gf {

static thread_local char g_create_err[512] = {0};

// (a context's error text can be written by a loader thread inside gf_smp_prepare while the compute thread fails a launch on the
//  same context: one process-wide lock keeps the two messages from interleaving)
static std::mutex g_err_mutex;
gf_status fail(gf_ctx *ctx, gf_status st, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (ctx) {
        std::lock_guard<std::mutex> lock(g_err_mutex);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    } else {
        vsnprintf(g_create_err, sizeof g_create_err, fmt, ap);   // (thread-local: nobody else writes it)
    }
    va_end(ap);
    return st;
}
// what gf_last_error hands out: a per-thread copy taken under the writers' lock, so a message a loader thread is writing into the
// context is never read half-way (round-3 advice)
static thread_local char g_err_snapshot[512] = {0};
static const char *error_snapshot(gf_ctx *ctx) {
    if (!ctx) return g_create_err;
    std::lock_guard<std::mutex> lock(g_err_mutex);
    std::memcpy(g_err_snapshot, ctx->err, sizeof g_err_snapshot);
    g_err_snapshot[sizeof g_err_snapshot - 1] = 0;
    return g_err_snapshot;
}

// GF_POISON=1: every device buffer the library hands out WITHOUT contents (workspace, staging, the SMP handle's pool blocks) is
// filled with 0xff bytes (NaN as floats, -1 as indices) first, so a kernel that reads what nobody wrote shows up in the parity
// tests instead of depending on what the allocation happened to hold.  Debug aid; off by default.
bool poison_buffers() {
    static const bool on = [] {
        const char *e = std::getenv("GF_POISON");
        return e && e[0] == '1';
    }();
    return on;
}

static gf_status grow(gf_ctx *ctx, void **buf, size_t *have, size_t want, bool pinned_host) {
    if (want <= *have) return GF_OK;
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    // GF_POISON=1 (debug): see poison_buffers().  On the context's stream: a null-stream hipMemset is not ordered against a
    // non-blocking stream and landed on top of results now and then
    if (!pinned_host && poison_buffers()) {
        GF_HIP_TRY(ctx, hipMemsetAsync(*buf, 0xff, want, ctx->stream));
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GF_OK;
}

void dist_teardown(gf_ctx *ctx);  // gf_dist.hip

gf_status ensure_ws(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->ws, &ctx->ws_bytes, bytes, false); }
gf_status ensure_stage(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->stage, &ctx->stage_bytes, bytes, false); }
gf_status ensure_pinned(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->pinned, &ctx->pinned_bytes, bytes, true); }
gf_status ensure_repack(gf_ctx *ctx, size_t bytes) { return grow(ctx, &ctx->repack, &ctx->repack_bytes, bytes, false); }

gf_status opt_in_lds_fn(gf_ctx *ctx, const void *kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(ctx, GF_ERR_UNSUPPORTED, "kernel needs %zu B of LDS (> 160 KiB)", bytes);
    if (bytes <= 32 * 1024) return GF_OK;
    // hipFuncSetAttribute SETS the kernel's limit on the current device for the whole process (it is not a maximum), and several
    // contexts can share a device (one default context per thread, RisiContraction_hip::set_gpu_stream): the record is per
    // (device, kernel), process-wide, and only ever raised -- a context that needs less never lowers another one's grant.
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted_by_device;
    std::lock_guard<std::mutex> lock(mu);
    size_t &granted = granted_by_device[std::make_pair(ctx->device, kernel)];
    if (bytes > granted) {
        GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
        GF_HIP_TRY(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        granted = bytes;
    }
    return GF_OK;
}

LaunchTimer::LaunchTimer(gf_ctx *c, const char *name, hipStream_t on) : ctx(c), stream(on ? on : c->stream) {
    if (!c->timing) return;
    // (the collectives of a data-parallel step are timed whatever the filter says: a few events per step on their own stream)
    if (c->timing_filter[0] && std::strcmp(c->timing_filter, name) != 0 && std::strncmp(name, "rccl_", 5) != 0) return;
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
    (void)hipEventRecord(start, stream);
}

// gf_hbm_copy_probe_f32: the copy the box's practical HBM ceiling is read from (bench.py: roofline.hbm_copy_*).  A workgroup walks
// contiguous 16 KiB tiles (four 16-byte requests per lane, 4 KiB apart) -- the shape tools/micro/copy_probe.hip found fastest: 5.5 - 5.9
// TB/s plain, 6.0 - 6.4 TB/s with `nt` on loads AND stores; a grid-stride copy whose four requests lie a whole grid apart gets 4.6.
template <bool NT>
__global__ __launch_bounds__(256) void copy_probe_f4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    using v4 = __attribute__((ext_vector_type(4))) float;
    const v4 *s = reinterpret_cast<const v4 *>(src);
    v4 *d = reinterpret_cast<v4 *>(dst);
    const size_t tile = 4 * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
        v4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = NT ? __builtin_nontemporal_load(s + i + k * 256) : s[i + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], d + i + k * 256);
            else d[i + k * 256] = v[k];
        }
    }
    for (size_t i = ntiles * tile + (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

void LaunchTimer::done() {
    if (slot < 0) return;
    (void)hipEventRecord(stop, stream);
    ctx->pending.push_back({slot, start, stop});
}

gf_status resolve_timers(gf_ctx *ctx) {
    if (ctx->pending.empty()) return GF_OK;
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (dist_poisoned(ctx)) return fail(ctx, GF_ERR_TIMEOUT, "gf_dist: the communicator timed out earlier; timers are not resolved before gf_dist_finalize");
    if (dist_active(ctx)) GF_HIP_TRY(ctx, hipStreamSynchronize(dist_stream(ctx)));   // (the collectives' events live on that stream)
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
// RisiContraction_18_dropout: per-slice factors (0 for a dropped slice) applied to an [.., 18, C] buffer
struct SliceFactors {
    float f[18];
};
__global__ void slice_scale(const float *__restrict__ in, float *__restrict__ out, size_t n, int C, SliceFactors fac) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float w = fac.f[(i / C) % 18];
        out[i] = w == 0.f ? 0.f : in[i] * w;  // a dropped slice is exactly 0 even if the input is not finite
    }
}
SliceFactors slice_factors(unsigned keep_mask, float scale) {
    SliceFactors fac;
    for (int k = 0; k < 18; ++k) fac.f[k] = ((keep_mask >> k) & 1u) ? scale : 0.f;
    return fac;
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


template <typename T>
gf_status dropout_forward_host(gf_ctx *ctx, unsigned keep_mask, double scale, const T *const *tensors, const T *A, T *out_value,
                               int N, int C) {
    gf_status st = contract_forward_host<T>(ctx, 18, tensors, A, out_value, N, C);
    if (st != GF_OK) return st;
    const size_t n = (size_t)N * N * 18 * C;
    for (size_t i = 0; i < n; ++i) {
        if (!((keep_mask >> ((i / C) % 18)) & 1u)) out_value[i] = 0;
        else if (scale != 1.0) out_value[i] = (T)(out_value[i] * scale);
    }
    return GF_OK;
}

template <typename T>
gf_status dropout_backward_host(gf_ctx *ctx, unsigned keep_mask, const T *out_gradient, const T *A, T *const *grads, int N,
                                int C) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!out_gradient || N <= 0 || C <= 0) return fail(ctx, GF_ERR_INVALID, "dropout_backward_host: bad argument");
    const size_t n = (size_t)N * N * 18 * C;
    std::vector<T> g(n);
    for (size_t i = 0; i < n; ++i) g[i] = ((keep_mask >> ((i / C) % 18)) & 1u) ? out_gradient[i] : (T)0;
    return contract_backward_host<T>(ctx, 18, g.data(), A, grads, N, C);
}

// ---- generic host-pointer staging for the mixers (mode A) -----------------------------------------------------
// Declare the operands of one op call; begin() packs the inputs into pinned memory, uploads them in one copy and
// converts to f32 on the device; end() converts the outputs back, downloads them in one copy, synchronises and
// either stores or adds them into the caller's buffers (`+=` is how every reference backward() writes gradients).
template <typename T>
class HostStaging {
public:
    explicit HostStaging(gf_ctx *c) : ctx(c) {}
    int in(const T *p, size_t n) { return push(const_cast<T *>(p), n, true, false, false); }
    int out(T *p, size_t n) { return push(p, n, false, true, false); }
    int out_add(T *p, size_t n) { return push(p, n, false, true, true); }
    float *dev(int i) { return fbase + bufs[i].off; }

    gf_status begin() {
        size_t total = 0;
        for (auto &b : bufs) {
            b.off = total;
            total += (b.n + 3) & ~(size_t)3;  // keep every operand 16-byte aligned in the f32 image
        }
        ntotal = total;
        gf_status st = ensure_pinned(ctx, sizeof(T) * total);
        if (st != GF_OK) return st;
        st = ensure_stage(ctx, align_up(sizeof(T) * total, 256) + sizeof(float) * total + 256);
        if (st != GF_OK) return st;
        hbase = static_cast<T *>(ctx->pinned);
        tbase = static_cast<T *>(ctx->stage);
        fbase = reinterpret_cast<float *>(static_cast<char *>(ctx->stage) + align_up(sizeof(T) * total, 256));
        for (auto &b : bufs)
            if (b.up && b.n) std::memcpy(hbase + b.off, b.host, sizeof(T) * b.n);
        // inputs are declared first, so [0, in_end) is one contiguous upload
        size_t in_end = 0;
        for (auto &b : bufs)
            if (b.up) in_end = b.off + ((b.n + 3) & ~(size_t)3);
        if (in_end) {
            GF_HIP_TRY(ctx, hipMemcpyAsync(tbase, hbase, sizeof(T) * in_end, hipMemcpyHostToDevice, ctx->stream));
            GF_LAUNCH(ctx, "cast_to_f32", cast_to_f32<T>, dim3(cast_grid(in_end)), dim3(256), 0, tbase, fbase, in_end);
        }
        out_begin = in_end;
        return GF_OK;
    }

    gf_status end() {
        const size_t n = ntotal - out_begin;
        if (n) {
            GF_LAUNCH(ctx, "cast_from_f32", cast_from_f32<T>, dim3(cast_grid(n)), dim3(256), 0, fbase + out_begin,
                      tbase + out_begin, n);
            GF_HIP_TRY(ctx, hipMemcpyAsync(hbase + out_begin, tbase + out_begin, sizeof(T) * n, hipMemcpyDeviceToHost,
                                           ctx->stream));
        }
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &b : bufs) {
            if (!b.down) continue;
            const T *src = hbase + b.off;
            if (b.add)
                for (size_t i = 0; i < b.n; ++i) b.host[i] += src[i];
            else
                std::memcpy(b.host, src, sizeof(T) * b.n);
        }
        return GF_OK;
    }

private:
    struct Buf {
        T *host;
        size_t n, off;
        bool up, down, add;
    };
    int push(T *p, size_t n, bool up, bool down, bool add) {
        bufs.push_back({p, n, 0, up, down, add});
        return (int)bufs.size() - 1;
    }
    gf_ctx *ctx;
    std::vector<Buf> bufs;
    T *hbase = nullptr, *tbase = nullptr;
    float *fbase = nullptr;
    size_t ntotal = 0, out_begin = 0;
};

template <typename T>
gf_status matmul_forward_host(gf_ctx *ctx, const T *A, const T *B, T *C, int M, int K, int N) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!A || !B || !C || M <= 0 || K <= 0 || N <= 0) return fail(ctx, GF_ERR_INVALID, "matmul_forward_host: bad argument");
    HostStaging<T> s(ctx);
    const int a = s.in(A, (size_t)M * K), b = s.in(B, (size_t)K * N), c = s.out(C, (size_t)M * N);
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    st = gf_matmul_forward_f32(ctx, s.dev(a), s.dev(b), s.dev(c), M, K, N);
    if (st != GF_OK) return st;
    return s.end();
}

template <typename T>
gf_status matmul_backward_host(gf_ctx *ctx, const T *dC, const T *A, const T *B, T *dA, T *dB, int M, int K, int N) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!dC || !A || !B || M <= 0 || K <= 0 || N <= 0) return fail(ctx, GF_ERR_INVALID, "matmul_backward_host: bad argument");
    HostStaging<T> s(ctx);
    const int g = s.in(dC, (size_t)M * N), a = s.in(A, (size_t)M * K), b = s.in(B, (size_t)K * N);
    const int da = dA ? s.out_add(dA, (size_t)M * K) : -1, db = dB ? s.out_add(dB, (size_t)K * N) : -1;
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    st = gf_matmul_backward_f32(ctx, s.dev(g), s.dev(a), s.dev(b), dA ? s.dev(da) : nullptr, dB ? s.dev(db) : nullptr, M, K,
                                N, 0);
    if (st != GF_OK) return st;
    return s.end();
}

// which = 0: MatTensorMul (first = X[R,Kd], second = F[Kd,J,D]); which = 1: TensorMatMul (first = F[R,Kd,D], second = Y[Kd,J])
template <typename T>
gf_status tensormul_forward_host(gf_ctx *ctx, int which, const T *first, const T *second, T *Out, int R, int Kd, int J, int D) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!first || !second || !Out || R <= 0 || Kd <= 0 || J <= 0 || D <= 0)
        return fail(ctx, GF_ERR_INVALID, "tensormul_forward_host: bad argument");
    const size_t n1 = which == 0 ? (size_t)R * Kd : (size_t)R * Kd * D, n2 = which == 0 ? (size_t)Kd * J * D : (size_t)Kd * J;
    HostStaging<T> s(ctx);
    const int a = s.in(first, n1), b = s.in(second, n2), c = s.out(Out, (size_t)R * J * D);
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    st = which == 0 ? gf_mattensormul_forward_f32(ctx, s.dev(a), s.dev(b), s.dev(c), R, Kd, J, D)
                    : gf_tensormatmul_forward_f32(ctx, s.dev(a), s.dev(b), s.dev(c), R, Kd, J, D);
    if (st != GF_OK) return st;
    return s.end();
}

template <typename T>
gf_status tensormul_backward_host(gf_ctx *ctx, int which, const T *G, const T *first, const T *second, T *d1, T *d2, int R,
                                  int Kd, int J, int D) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!G || !first || !second || R <= 0 || Kd <= 0 || J <= 0 || D <= 0)
        return fail(ctx, GF_ERR_INVALID, "tensormul_backward_host: bad argument");
    const size_t n1 = which == 0 ? (size_t)R * Kd : (size_t)R * Kd * D, n2 = which == 0 ? (size_t)Kd * J * D : (size_t)Kd * J;
    HostStaging<T> s(ctx);
    const int g = s.in(G, (size_t)R * J * D), a = s.in(first, n1), b = s.in(second, n2);
    const int o1 = d1 ? s.out_add(d1, n1) : -1, o2 = d2 ? s.out_add(d2, n2) : -1;
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    float *p1 = d1 ? s.dev(o1) : nullptr, *p2 = d2 ? s.dev(o2) : nullptr;
    st = which == 0 ? gf_mattensormul_backward_f32(ctx, s.dev(g), s.dev(a), s.dev(b), p1, p2, R, Kd, J, D, 0)
                    : gf_tensormatmul_backward_f32(ctx, s.dev(g), s.dev(a), s.dev(b), p1, p2, R, Kd, J, D, 0);
    if (st != GF_OK) return st;
    return s.end();
}

template <typename T>
gf_status custommatmultensor_forward_host(gf_ctx *ctx, const T *W, const T *X, T *Out, long long rows, int V, int Kout) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!W || !X || !Out || rows <= 0 || V <= 0 || Kout <= 0)
        return fail(ctx, GF_ERR_INVALID, "custommatmultensor_forward_host: bad argument");
    HostStaging<T> s(ctx);
    const int w = s.in(W, (size_t)Kout * V), x = s.in(X, (size_t)rows * V), o = s.out(Out, (size_t)rows * Kout);
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    st = gf_custommatmultensor_forward_f32(ctx, s.dev(w), s.dev(x), s.dev(o), rows, V, Kout);
    if (st != GF_OK) return st;
    return s.end();
}

template <typename T>
gf_status custommatmultensor_backward_host(gf_ctx *ctx, const T *G, const T *W, const T *X, T *dW, T *dX, long long rows,
                                           int V, int Kout) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!G || !W || !X || rows <= 0 || V <= 0 || Kout <= 0)
        return fail(ctx, GF_ERR_INVALID, "custommatmultensor_backward_host: bad argument");
    HostStaging<T> s(ctx);
    const int g = s.in(G, (size_t)rows * Kout), w = s.in(W, (size_t)Kout * V), x = s.in(X, (size_t)rows * V);
    const int ow = dW ? s.out_add(dW, (size_t)Kout * V) : -1, ox = dX ? s.out_add(dX, (size_t)rows * V) : -1;
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    st = gf_custommatmultensor_backward_f32(ctx, s.dev(g), s.dev(w), s.dev(x), dW ? s.dev(ow) : nullptr,
                                            dX ? s.dev(ox) : nullptr, rows, V, Kout, 0);
    if (st != GF_OK) return st;
    return s.end();
}

// StackTensor3D in host mode: staging the nRows tensors into one device image IS the stack; the copy kernel makes
// it contiguous, then one D2H.  Backward: upload G once, scatter-add into the nRows gradient buffers on return.
template <typename T>
gf_status stack_forward_host(gf_ctx *ctx, const T *const *tensors, T *out, int nRows, size_t per) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!tensors || !out || nRows <= 0) return fail(ctx, GF_ERR_INVALID, "stack_forward_host: bad argument");
    HostStaging<T> s(ctx);
    std::vector<int> ids(nRows);
    for (int r = 0; r < nRows; ++r) {
        if (!tensors[r]) return fail(ctx, GF_ERR_INVALID, "tensors[%d] is null", r);
        ids[r] = s.in(tensors[r], per);
    }
    const int o = s.out(out, (size_t)nRows * per);
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    for (int r = 0; r < nRows; ++r)
        GF_HIP_TRY(ctx, hipMemcpyAsync(s.dev(o) + (size_t)r * per, s.dev(ids[r]), sizeof(float) * per,
                                       hipMemcpyDeviceToDevice, ctx->stream));
    return s.end();
}
template <typename T>
gf_status stack_backward_host(gf_ctx *ctx, const T *G, T *const *grads, int nRows, size_t per) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!G || !grads || nRows <= 0) return fail(ctx, GF_ERR_INVALID, "stack_backward_host: bad argument");
    HostStaging<T> s(ctx);
    const int g = s.in(G, (size_t)nRows * per);
    std::vector<int> ids(nRows);
    for (int r = 0; r < nRows; ++r) {
        if (!grads[r]) return fail(ctx, GF_ERR_INVALID, "grads[%d] is null", r);
        ids[r] = s.out_add(grads[r], per);
    }
    gf_status st = s.begin();
    if (st != GF_OK) return st;
    for (int r = 0; r < nRows; ++r)
        GF_HIP_TRY(ctx, hipMemcpyAsync(s.dev(ids[r]), s.dev(g) + (size_t)r * per, sizeof(float) * per,
                                       hipMemcpyDeviceToDevice, ctx->stream));
    return s.end();
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
    if (!gf::dist_poisoned(ctx)) (void)hipStreamSynchronize(ctx->stream);   // (poisoned: abort the communicator first, see dist_teardown)
    gf::dist_teardown(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->repack) (void)hipFree(ctx->repack);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)gf::resolve_timers(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GF_OK;
}

gf_status gf_ctx_set_option(gf_ctx *ctx, int option, int value) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    switch (option) {
        case GF_OPT_R18_GENERIC_KERNELS: ctx->r18_generic = value ? 1 : 0; return GF_OK;
        case GF_OPT_SMP_FP32_PRODUCTS: ctx->fp32_products = value ? 1 : 0; return GF_OK;
        default: return gf::fail(ctx, GF_ERR_INVALID, "gf_ctx_set_option: unknown option %d", option);
    }
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

gf_status gf_ctx_set_timing_filter(gf_ctx *ctx, const char *kernel_name) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    std::snprintf(ctx->timing_filter, sizeof ctx->timing_filter, "%s", kernel_name ? kernel_name : "");
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

gf_status gf_hbm_copy_probe_f32(gf_ctx *ctx, float *dst, const float *src, size_t n, int mode, int iters, double *ms_per_copy) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    if (!dst || !src || !ms_per_copy || n == 0 || n % 4 != 0 || iters < 1 || (mode != 0 && mode != 1))
        return gf::fail(ctx, GF_ERR_INVALID, "gf_hbm_copy_probe_f32: bad argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    GF_HIP_TRY(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return gf::fail(ctx, GF_ERR_HIP, "gf_hbm_copy_probe_f32: hipEventCreate failed");
    }
    const size_t n4 = n / 4;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const unsigned blocks = (unsigned)std::min<size_t>((n4 + 1023) / 1024, (size_t)cus * 16);
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(gf::copy_probe_f4<false>, dim3(blocks), dim3(256), 0, ctx->stream, (const float4 *)src, (float4 *)dst, n4);
        else hipLaunchKernelGGL(gf::copy_probe_f4<true>, dim3(blocks), dim3(256), 0, ctx->stream, (const float4 *)src, (float4 *)dst, n4);
    };
    launch();   // (warm: page tables, clocks)
    (void)hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < iters; ++i) launch();
    (void)hipEventRecord(e1, ctx->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return gf::fail(ctx, GF_ERR_HIP, "gf_hbm_copy_probe_f32: %s", hipGetErrorString(e));
    *ms_per_copy = (double)ms / iters;
    return GF_OK;
}

const char *gf_last_error(gf_ctx *ctx) { return gf::error_snapshot(ctx); }

size_t gf_contract_workspace_bytes(int K, int N, int C, int batch) {
    if (N <= 0 || C <= 0 || batch <= 0) return 0;
    switch (K) {
        case 18: return gf::r18_workspace_bytes(N, C, batch);
        case 4: case 10: case 50: return gf::family_workspace_bytes(K, N, C, batch);
        default: return 0;
    }
}

}  // extern "C"

// ---- channel counts that are not a multiple of four (the reference's own tests run nChanels = 10) ---------------------------------
// The slab kernels of RisiContraction_18 move 16 bytes per lane over the channel axis; at C % 4 != 0 the entry points used to fall to
// the thread-per-element kernels (N = 32, C = 10, batch 256: 13.0 ms forward + backward, against 0.57 ms at C = 12).  Now the operands are REPACKED: rows of C floats -> rows of C4 = 4 ceil(C / 4) floats with zero fill (the
// contractions act channel by channel, a zero channel stays zero), the family's kernels run at C4 on the copies, and the result is cropped
// back (+= for the accumulating backward).  Two extra passes over operand and result, in the context's own scratch.
// GF_OPT_R18_GENERIC_KERNELS keeps RisiContraction_18 on the generic kernels at any C (the parity tests' second implementation).
namespace gf {
__global__ void repack_channels(const float *__restrict__ src, float *__restrict__ dst, size_t rows, int C, int C4) {   // dst[row][C4] <- src[row][C] | 0
    const size_t total = rows * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C4;
        const int c = (int)(i - row * C4);
        dst[i] = c < C ? src[row * C + c] : 0.f;
    }
}
__global__ void crop_channels(const float *__restrict__ src, float *__restrict__ dst, size_t rows, int C, int C4, int accumulate) {   // dst[row][C] (+)= src[row][C4]
    const size_t total = rows * (size_t)C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C;
        const int c = (int)(i - row * C);
        const float v = src[row * C4 + c];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
// (RisiContraction_18 only: the table kernels of _10 / _50 and the thread kernels of _4 lose less at scalar channel loads than the two
//  extra passes cost -- SMP_2D_ver6 / ver7 steps at C = 10 measured 10.1 -> 12.0 and 24.3 -> 28.2 ms with their operands repacked)
static bool repack_wanted(const gf_ctx *ctx, int K, int C) { return K == 18 && C % 4 != 0 && !ctx->r18_generic; }
// in: [rows_in][C] operand, out: [rows_out][C] result; run(in4, out4) computes at C4 on the repacked copies
template <typename Run>
static gf_status with_repacked_channels(gf_ctx *ctx, const float *in, size_t rows_in, float *out, size_t rows_out, int C, int accumulate, Run run) {
    const int C4 = (C + 3) & ~3;
    const size_t n_in = align_up(rows_in * (size_t)C4, 64), n_out = align_up(rows_out * (size_t)C4, 64);
    gf_status st = ensure_repack(ctx, sizeof(float) * (n_in + n_out) + 256);
    if (st != GF_OK) return st;
    float *in4 = static_cast<float *>(ctx->repack), *out4 = in4 + n_in;
    GF_LAUNCH(ctx, "repack_channels", repack_channels, dim3(cast_grid(rows_in * (size_t)C4)), dim3(256), 0, in, in4, rows_in, C, C4);
    st = run(in4, out4, C4);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "crop_channels", crop_channels, dim3(cast_grid(rows_out * (size_t)C)), dim3(256), 0, out4, out, rows_out, C, C4, accumulate ? 1 : 0);
    return GF_OK;
}
}  // namespace gf

extern "C" {

gf_status gf_contract_forward_f32(gf_ctx *ctx, int K, const float *P, const float *A, float *Out, int N, int C,
                                  int batch) {
    gf_status st = gf::check_contract_args(ctx, K, P, A, Out, N, C, batch);
    if (st != GF_OK || batch == 0) return st;
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (gf::repack_wanted(ctx, K, C))
        return gf::with_repacked_channels(ctx, P, (size_t)batch * N * N * N, Out, (size_t)batch * N * N * K, C, 0,
                                          [&](const float *P4, float *Out4, int C4) { return gf_contract_forward_f32(ctx, K, P4, A, Out4, N, C4, batch); });
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
    if (gf::repack_wanted(ctx, K, C))   // (write-only at C4, then dP (+)= the cropped result)
        return gf::with_repacked_channels(ctx, G, (size_t)batch * N * N * K, dP, (size_t)batch * N * N * N, C, accumulate,
                                          [&](const float *G4, float *dP4, int C4) { return gf_contract_backward_f32(ctx, K, G4, A, dP4, N, C4, batch, 0); });
    switch (K) {
        case 18: return gf::r18_backward(ctx, G, A, dP, N, C, batch, accumulate);
        default: return gf::family_backward(ctx, K, G, A, dP, N, C, batch, accumulate);
    }
}

gf_status gf_contract18_dropout_forward_f32(gf_ctx *ctx, unsigned keep_mask, float scale, const float *P, const float *A,
                                            float *Out, int N, int C, int batch) {
    gf_status st = gf_contract_forward_f32(ctx, 18, P, A, Out, N, C, batch);
    if (st != GF_OK || batch == 0) return st;
    if ((keep_mask & 0x3ffffu) == 0x3ffffu && scale == 1.f) return GF_OK;
    const size_t n = (size_t)batch * N * N * 18 * C;
    GF_LAUNCH(ctx, "slice_scale", gf::slice_scale, dim3(gf::cast_grid(n)), dim3(256), 0, Out, Out, n, C,
              gf::slice_factors(keep_mask, scale));
    return GF_OK;
}

gf_status gf_contract18_dropout_backward_f32(gf_ctx *ctx, unsigned keep_mask, const float *G, const float *A, float *dP,
                                             int N, int C, int batch, int accumulate) {
    gf_status st = gf::check_contract_args(ctx, 18, G, A, dP, N, C, batch);
    if (st != GF_OK || batch == 0) return st;
    if ((keep_mask & 0x3ffffu) == 0x3ffffu) return gf_contract_backward_f32(ctx, 18, G, A, dP, N, C, batch, accumulate);
    // dropped slices contribute nothing (RisiContraction_18_dropout.h:498-783 skips them): masked copy of G, then the plain vjp
    const size_t n = (size_t)batch * N * N * 18 * C;
    st = gf::ensure_stage(ctx, n * sizeof(float));
    if (st != GF_OK) return st;
    float *Gm = static_cast<float *>(ctx->stage);
    GF_LAUNCH(ctx, "slice_scale", gf::slice_scale, dim3(gf::cast_grid(n)), dim3(256), 0, G, Gm, n, C,
              gf::slice_factors(keep_mask, 1.f));
    return gf_contract_backward_f32(ctx, 18, Gm, A, dP, N, C, batch, accumulate);
}

gf_status gf_contract18_dropout_forward_host_f64(gf_ctx *ctx, unsigned keep_mask, double scale, const double *const *tensors,
                                                 const double *A, double *out_value, int N, int C) {
    return gf::dropout_forward_host<double>(ctx, keep_mask, scale, tensors, A, out_value, N, C);
}
gf_status gf_contract18_dropout_backward_host_f64(gf_ctx *ctx, unsigned keep_mask, const double *out_gradient, const double *A,
                                                  double *const *grads, int N, int C) {
    return gf::dropout_backward_host<double>(ctx, keep_mask, out_gradient, A, grads, N, C);
}
gf_status gf_contract18_dropout_forward_host_f32(gf_ctx *ctx, unsigned keep_mask, double scale, const float *const *tensors,
                                                 const float *A, float *out_value, int N, int C) {
    return gf::dropout_forward_host<float>(ctx, keep_mask, scale, tensors, A, out_value, N, C);
}
gf_status gf_contract18_dropout_backward_host_f32(gf_ctx *ctx, unsigned keep_mask, const float *out_gradient, const float *A,
                                                  float *const *grads, int N, int C) {
    return gf::dropout_backward_host<float>(ctx, keep_mask, out_gradient, A, grads, N, C);
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

#define GF_HOST_MIXERS(SFX, T)                                                                                      \
    gf_status gf_matmul_forward_host_##SFX(gf_ctx *ctx, const T *A, const T *B, T *C, int M, int K, int N) {           \
        return gf::matmul_forward_host<T>(ctx, A, B, C, M, K, N);                                                       \
    }                                                                                                                   \
    gf_status gf_matmul_backward_host_##SFX(gf_ctx *ctx, const T *dC, const T *A, const T *B, T *dA, T *dB, int M,     \
                                            int K, int N) {                                                             \
        return gf::matmul_backward_host<T>(ctx, dC, A, B, dA, dB, M, K, N);                                             \
    }                                                                                                                   \
    gf_status gf_mattensormul_forward_host_##SFX(gf_ctx *ctx, const T *X, const T *F, T *Out, int R, int Kd, int J,    \
                                                 int D) {                                                               \
        return gf::tensormul_forward_host<T>(ctx, 0, X, F, Out, R, Kd, J, D);                                           \
    }                                                                                                                   \
    gf_status gf_mattensormul_backward_host_##SFX(gf_ctx *ctx, const T *G, const T *X, const T *F, T *dX, T *dF, int R, \
                                                  int Kd, int J, int D) {                                               \
        return gf::tensormul_backward_host<T>(ctx, 0, G, X, F, dX, dF, R, Kd, J, D);                                    \
    }                                                                                                                   \
    gf_status gf_tensormatmul_forward_host_##SFX(gf_ctx *ctx, const T *F, const T *Y, T *Out, int R, int Kd, int J,    \
                                                 int D) {                                                               \
        return gf::tensormul_forward_host<T>(ctx, 1, F, Y, Out, R, Kd, J, D);                                           \
    }                                                                                                                   \
    gf_status gf_tensormatmul_backward_host_##SFX(gf_ctx *ctx, const T *G, const T *F, const T *Y, T *dF, T *dY, int R, \
                                                  int Kd, int J, int D) {                                               \
        return gf::tensormul_backward_host<T>(ctx, 1, G, F, Y, dF, dY, R, Kd, J, D);                                    \
    }                                                                                                                   \
    gf_status gf_custommatmultensor_forward_host_##SFX(gf_ctx *ctx, const T *W, const T *X, T *Out, long long rows,     \
                                                       int V, int Kout) {                                               \
        return gf::custommatmultensor_forward_host<T>(ctx, W, X, Out, rows, V, Kout);                                   \
    }                                                                                                                   \
    gf_status gf_custommatmultensor_backward_host_##SFX(gf_ctx *ctx, const T *G, const T *W, const T *X, T *dW, T *dX, \
                                                        long long rows, int V, int Kout) {                              \
        return gf::custommatmultensor_backward_host<T>(ctx, G, W, X, dW, dX, rows, V, Kout);                            \
    }                                                                                                                   \
    gf_status gf_stack_forward_host_##SFX(gf_ctx *ctx, const T *const *tensors, T *out, int nRows, size_t per_tensor) { \
        return gf::stack_forward_host<T>(ctx, tensors, out, nRows, per_tensor);                                         \
    }                                                                                                                   \
    gf_status gf_stack_backward_host_##SFX(gf_ctx *ctx, const T *G, T *const *grads, int nRows, size_t per_tensor) {    \
        return gf::stack_backward_host<T>(ctx, G, grads, nRows, per_tensor);                                            \
    }
GF_HOST_MIXERS(f64, double)
GF_HOST_MIXERS(f32, float)


}  // extern "C"
