// gf_runtime.h -- the thin layer between the Entity-style op classes and the C ABI (include/gf_hip.h):
// a per-thread default context, scalar-type overloads of the host-pointer entry points, and the error policy.
//
// Error policy mirrors the reference: its ops assert() and abort (SURVEY.md 8b "Errors").  Ours print the
// library's message and abort -- there is no CPU fallback to fall back to (contrast the CPU switch of
// GraphFlow_gpu/RisiContraction_18_gpu.h:961-968, which this port deliberately does not reproduce).
#ifndef GF_RUNTIME_H_INCLUDED
#define GF_RUNTIME_H_INCLUDED

#include <cstdio>
#include <cstdlib>

#include <sys/syscall.h>
#include <unistd.h>

#include "gf_hip.h"

namespace gfhost {

inline void die(gf_ctx *ctx, const char *where, gf_status st) {
    std::fprintf(stderr, "graphflow_amd: %s failed (status %d): %s\n", where, (int)st, gf_last_error(ctx));
    std::abort();
}

// One context per host thread, created on first use on device GF_DEVICE (default 0): the reference's threading
// rule is one model clone per worker thread with no sharing (SMP_omega.h:115-129).
// Destroyed when the thread exits (workspace, staging and pinned buffers go with it): models and ops of the thread must not
// outlive the thread, as in the reference, where a worker's model clone is used by that worker only.
struct ThreadContext {
    gf_ctx *ctx;
    ThreadContext() : ctx(NULL) {}
    ~ThreadContext() {
        // (not on the main thread: its thread-locals die BEFORE objects of static storage duration, and the reference's drivers
        //  keep their models at file scope -- `SMP_omega train_network(...)`; the process is exiting there anyway)
        if (ctx && (long)syscall(SYS_gettid) != (long)getpid()) gf_ctx_destroy(ctx);
    }
};
inline gf_ctx *default_context() {
    static thread_local ThreadContext holder;
    if (!holder.ctx) {
        const char *dev = std::getenv("GF_DEVICE");
        gf_status st = gf_ctx_create(&holder.ctx, dev ? std::atoi(dev) : 0, NULL);
        if (st != GF_OK) die(NULL, "gf_ctx_create", st);
    }
    return holder.ctx;
}

inline gf_status contract_forward_host(gf_ctx *c, int K, const double *const *t, const double *A, double *out, int N, int C) {
    return gf_contract_forward_host_f64(c, K, t, A, out, N, C);
}
inline gf_status contract_forward_host(gf_ctx *c, int K, const float *const *t, const float *A, float *out, int N, int C) {
    return gf_contract_forward_host_f32(c, K, t, A, out, N, C);
}
inline gf_status contract_backward_host(gf_ctx *c, int K, const double *g, const double *A, double *const *d, int N, int C) {
    return gf_contract_backward_host_f64(c, K, g, A, d, N, C);
}
inline gf_status contract_backward_host(gf_ctx *c, int K, const float *g, const float *A, float *const *d, int N, int C) {
    return gf_contract_backward_host_f32(c, K, g, A, d, N, C);
}

inline gf_status contract18_dropout_forward_host(gf_ctx *c, unsigned m, double sc, const double *const *t, const double *A,
                                                 double *o, int N, int C) {
    return gf_contract18_dropout_forward_host_f64(c, m, sc, t, A, o, N, C);
}
inline gf_status contract18_dropout_forward_host(gf_ctx *c, unsigned m, double sc, const float *const *t, const float *A,
                                                 float *o, int N, int C) {
    return gf_contract18_dropout_forward_host_f32(c, m, sc, t, A, o, N, C);
}
inline gf_status contract18_dropout_backward_host(gf_ctx *c, unsigned m, const double *g, const double *A, double *const *d,
                                                  int N, int C) {
    return gf_contract18_dropout_backward_host_f64(c, m, g, A, d, N, C);
}
inline gf_status contract18_dropout_backward_host(gf_ctx *c, unsigned m, const float *g, const float *A, float *const *d, int N,
                                                  int C) {
    return gf_contract18_dropout_backward_host_f32(c, m, g, A, d, N, C);
}

#define GF_HOST_OVERLOAD2(NAME, ARGS_D, ARGS_F, CALL)                              \
    inline gf_status NAME ARGS_D { return gf_##NAME##_f64 CALL; }                   \
    inline gf_status NAME ARGS_F { return gf_##NAME##_f32 CALL; }

GF_HOST_OVERLOAD2(matmul_forward_host, (gf_ctx * c, const double *A, const double *B, double *C, int M, int K, int N),
                  (gf_ctx * c, const float *A, const float *B, float *C, int M, int K, int N), (c, A, B, C, M, K, N))
GF_HOST_OVERLOAD2(matmul_backward_host,
                  (gf_ctx * c, const double *g, const double *A, const double *B, double *dA, double *dB, int M, int K, int N),
                  (gf_ctx * c, const float *g, const float *A, const float *B, float *dA, float *dB, int M, int K, int N),
                  (c, g, A, B, dA, dB, M, K, N))
GF_HOST_OVERLOAD2(mattensormul_forward_host,
                  (gf_ctx * c, const double *X, const double *F, double *O, int R, int Kd, int J, int D),
                  (gf_ctx * c, const float *X, const float *F, float *O, int R, int Kd, int J, int D), (c, X, F, O, R, Kd, J, D))
GF_HOST_OVERLOAD2(mattensormul_backward_host,
                  (gf_ctx * c, const double *g, const double *X, const double *F, double *dX, double *dF, int R, int Kd, int J, int D),
                  (gf_ctx * c, const float *g, const float *X, const float *F, float *dX, float *dF, int R, int Kd, int J, int D),
                  (c, g, X, F, dX, dF, R, Kd, J, D))
GF_HOST_OVERLOAD2(tensormatmul_forward_host,
                  (gf_ctx * c, const double *F, const double *Y, double *O, int R, int Kd, int J, int D),
                  (gf_ctx * c, const float *F, const float *Y, float *O, int R, int Kd, int J, int D), (c, F, Y, O, R, Kd, J, D))
GF_HOST_OVERLOAD2(tensormatmul_backward_host,
                  (gf_ctx * c, const double *g, const double *F, const double *Y, double *dF, double *dY, int R, int Kd, int J, int D),
                  (gf_ctx * c, const float *g, const float *F, const float *Y, float *dF, float *dY, int R, int Kd, int J, int D),
                  (c, g, F, Y, dF, dY, R, Kd, J, D))
GF_HOST_OVERLOAD2(custommatmultensor_forward_host,
                  (gf_ctx * c, const double *W, const double *T, double *O, long long rows, int V, int Kout),
                  (gf_ctx * c, const float *W, const float *T, float *O, long long rows, int V, int Kout), (c, W, T, O, rows, V, Kout))
GF_HOST_OVERLOAD2(custommatmultensor_backward_host,
                  (gf_ctx * c, const double *g, const double *W, const double *T, double *dW, double *dT, long long rows, int V, int Kout),
                  (gf_ctx * c, const float *g, const float *W, const float *T, float *dW, float *dT, long long rows, int V, int Kout),
                  (c, g, W, T, dW, dT, rows, V, Kout))
GF_HOST_OVERLOAD2(stack_forward_host, (gf_ctx * c, const double *const *t, double *o, int n, size_t per),
                  (gf_ctx * c, const float *const *t, float *o, int n, size_t per), (c, t, o, n, per))
GF_HOST_OVERLOAD2(stack_backward_host, (gf_ctx * c, const double *g, double *const *d, int n, size_t per),
                  (gf_ctx * c, const float *g, float *const *d, int n, size_t per), (c, g, d, n, per))
#undef GF_HOST_OVERLOAD2

}  // namespace gfhost
#endif
