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

#include "gf_hip.h"

namespace gfhost {

inline void die(gf_ctx *ctx, const char *where, gf_status st) {
    std::fprintf(stderr, "graphflow_amd: %s failed (status %d): %s\n", where, (int)st, gf_last_error(ctx));
    std::abort();
}

// One context per host thread, created on first use on device GF_DEVICE (default 0): the reference's threading
// rule is one model clone per worker thread with no sharing (SMP_omega.h:115-129).
inline gf_ctx *default_context() {
    static thread_local gf_ctx *ctx = NULL;
    if (!ctx) {
        const char *dev = std::getenv("GF_DEVICE");
        gf_status st = gf_ctx_create(&ctx, dev ? std::atoi(dev) : 0, NULL);
        if (st != GF_OK) die(NULL, "gf_ctx_create", st);
    }
    return ctx;
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

}  // namespace gfhost
#endif
