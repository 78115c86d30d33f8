// mixers.hip -- the dense feature mixers of the SMP path on gfx950: MatMul, MatTensorMul, TensorMatMul, StackTensor3D.
//
// Replaces GraphFlow/MatMul.h:48-82, MatTensorMul.h:47-85, TensorMatMul.h:46-84 (naive triple loops) and
// StackTensor3D.h:54-90.  All three products are one strided-batched fp32 GEMM on the matrix cores:
//   MatMul        C[M,N]      = A[M,K] B[K,N]                  bwd  dA += dC B^T,  dB += A^T dC
//   MatTensorMul  Out[R,J*D]  = X[R,Kd] F[Kd,J*D]              bwd  dX += G F^T,   dF += X^T G
//   TensorMatMul  Out_i[J,D]  = Y^T[J,Kd] F_i[Kd,D], i < R     bwd  dF_i += Y G_i, dY += sum_i F_i G_i^T
// using v_mfma_f32_32x32x2_f32 (exact fp32, the only fp32-input matrix instruction on gfx950; there is no
// xf32/TF32).  Tile: 64 x 64 x 32 per 256-thread workgroup, 2 x 2 waves of one 32 x 32 accumulator each, operands
// staged through padded LDS, next tile's global loads issued before the current tile's MFMAs.  Reductions over a
// long K with a small output (dB of the K-projection: K = sum s^2 rows) are split over K into a workspace and
// summed by a second kernel in a fixed order, so results are deterministic (no atomics).
#include "gf_internal.h"

namespace gf {
namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int kThreads = 256;
using f16v = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    int lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int kchunk;            // K range per split (multiple of BK); splits = gridDim.z / batch
    int batch;
    int accumulate;        // C += result (only when not splitting)
};

// element (m,k) of op(A): TA ? A[k*lda + m] : A[m*lda + k];   element (k,n) of op(B): TB ? B[n*ldb + k] : B[k*ldb + n]
template <bool TA, bool TB>
__global__ __launch_bounds__(kThreads) void gemm_f32_mfma(GemmArgs g) {
    __shared__ float As[BM * (BK + 1)];
    __shared__ float Bs[BK * (BN + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;  // M tiles on x: M = sum s^2 can be millions of rows
    const int bz = blockIdx.z % g.batch, split = blockIdx.z / g.batch;
    const float *A = g.A + bz * g.sA, *B = g.B + bz * g.sB;
    const int kbeg = split * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;

    float ra[8], rb[8];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + e * kThreads;
            int m, k;
            if (TA) {
                m = idx % BM;
                k = idx / BM;
            } else {
                k = idx % BK;
                m = idx / BK;
            }
            const int gm = m0 + m, gk = k0 + k;
            const bool ok = gm < g.M && gk < kend;
            const size_t off = TA ? (size_t)gk * g.lda + gm : (size_t)gm * g.lda + gk;
            ra[e] = ok ? A[off] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + e * kThreads;
            int k, n;
            if (TB) {
                k = idx % BK;
                n = idx / BK;
            } else {
                n = idx % BN;
                k = idx / BN;
            }
            const int gn = n0 + n, gk = k0 + k;
            const bool ok = gn < g.N && gk < kend;
            const size_t off = TB ? (size_t)gn * g.ldb + gk : (size_t)gk * g.ldb + gn;
            rb[e] = ok ? B[off] : 0.f;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + e * kThreads;
            int m, k;
            if (TA) {
                m = idx % BM;
                k = idx / BM;
            } else {
                k = idx % BK;
                m = idx / BK;
            }
            As[m * (BK + 1) + k] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + e * kThreads;
            int k, n;
            if (TB) {
                k = idx % BK;
                n = idx / BK;
            } else {
                n = idx % BN;
                k = idx / BN;
            }
            Bs[k * (BN + 1) + n] = rb[e];
        }
    };

    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (kbeg < kend) {
        load_tiles(kbeg);
        store_tiles();
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            const bool more = k0 + BK < kend;
            if (more) load_tiles(k0 + BK);
            const float *ap = As + (wm * 32 + (lane & 31)) * (BK + 1) + (lane >> 5);
            const float *bp = Bs + (lane >> 5) * (BN + 1) + wn * 32 + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk * (BN + 1)], acc, 0, 0, 0);
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int splits = gridDim.z / g.batch;
    float *C = g.C + (splits > 1 ? (size_t)split * g.M * g.ldc : (size_t)0) + bz * g.sC;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                float *c = C + (size_t)row * g.ldc + col;
                *c = g.accumulate ? *c + acc[r] : acc[r];
            }
        }
    }
}

// C[i] (+)= sum_s part[s][i], fixed order
__global__ void splitk_reduce(const float *__restrict__ part, float *__restrict__ C, size_t n, int splits, int accumulate) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
        C[i] = accumulate ? C[i] + s : s;
    }
}

template <typename T>
__global__ void gather_rows(const T *const *__restrict__ src, T *__restrict__ dst, size_t per, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i / per][i % per];
}
template <typename T>
__global__ void scatter_add_rows(const T *__restrict__ G, T *const *__restrict__ dst, size_t per, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        dst[i / per][i % per] += G[i];
}

}  // namespace

// op(A)[M,K] op(B)[K,N] -> C[M,N] (ldc = row stride of C), batched with element strides; accumulate: C += .
gf_status gemm(gf_ctx *ctx, bool ta, bool tb, int M, int N, int K, const float *A, int lda, long long sA, const float *B,
               int ldb, long long sB, float *C, int ldc, long long sC, int batch, int accumulate) {
    if (M <= 0 || N <= 0 || batch <= 0) return GF_OK;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = sA; g.sB = sB; g.sC = sC; g.batch = batch; g.accumulate = accumulate;
    const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
    // split K when the output has too few tiles to fill 256 CUs and K is long (e.g. dB = A^T dC of the K-projection)
    int splits = 1;
    const long long tiles = (long long)gx * gy * batch;
    if (batch == 1 && tiles < 512 && K >= 8 * BK) {
        splits = (int)((1024 + tiles - 1) / tiles);
        const int maxs = K / (2 * BK);
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
    }
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    if (kchunk <= 0) kchunk = BK;
    splits = (K + kchunk - 1) / kchunk;
    if (splits < 1) splits = 1;
    g.kchunk = kchunk;
    float *part = nullptr;
    if (splits > 1 && ldc != N) return fail(ctx, GF_ERR_UNSUPPORTED, "split-K needs a dense C (ldc == N)");
    if (splits > 1) {
        gf_status st = ensure_ws(ctx, sizeof(float) * (size_t)splits * M * N + 256);
        if (st != GF_OK) return st;
        part = static_cast<float *>(ctx->ws);
        g.C = part;
        g.ldc = N;
        g.accumulate = 0;
    }
    const dim3 grid(gy, gx, batch * splits);
    if (!ta && !tb) GF_LAUNCH(ctx, "gemm_nn", (gemm_f32_mfma<false, false>), grid, dim3(kThreads), 0, g);
    if (!ta && tb) GF_LAUNCH(ctx, "gemm_nt", (gemm_f32_mfma<false, true>), grid, dim3(kThreads), 0, g);
    if (ta && !tb) GF_LAUNCH(ctx, "gemm_tn", (gemm_f32_mfma<true, false>), grid, dim3(kThreads), 0, g);
    if (ta && tb) GF_LAUNCH(ctx, "gemm_tt", (gemm_f32_mfma<true, true>), grid, dim3(kThreads), 0, g);
    if (splits > 1) {
        const size_t n = (size_t)M * N;
        size_t blocks = (n + 255) / 256;
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, part,
                  C, n, splits, accumulate);
    }
    return GF_OK;
}

gf_status stack_forward(gf_ctx *ctx, const float *const *tensors, float *out, int nRows, size_t per) {
    const size_t total = (size_t)nRows * per;
    size_t blocks = (total + 255) / 256;
    GF_LAUNCH(ctx, "stack_gather", gather_rows<float>, dim3((unsigned)(blocks > 65536 ? 65536 : (blocks ? blocks : 1))),
              dim3(256), 0, tensors, out, per, total);
    return GF_OK;
}

gf_status stack_backward(gf_ctx *ctx, const float *G, float *const *grads, int nRows, size_t per) {
    const size_t total = (size_t)nRows * per;
    size_t blocks = (total + 255) / 256;
    GF_LAUNCH(ctx, "stack_scatter_add", scatter_add_rows<float>,
              dim3((unsigned)(blocks > 65536 ? 65536 : (blocks ? blocks : 1))), dim3(256), 0, G, grads, per, total);
    return GF_OK;
}

}  // namespace gf

extern "C" {

static gf_status need(gf_ctx *ctx, bool ok, const char *what) {
    if (!ctx) return gf::fail(nullptr, GF_ERR_INVALID, "null context");
    if (!ok) return gf::fail(ctx, GF_ERR_INVALID, "%s", what);
    return GF_OK;
}

gf_status gf_matmul_forward_f32(gf_ctx *ctx, const float *A, const float *B, float *C, int M, int K, int N) {
    gf_status st = need(ctx, A && B && C && M > 0 && K > 0 && N > 0, "gf_matmul_forward_f32: null pointer or non-positive size");
    if (st != GF_OK) return st;
    return gf::gemm(ctx, false, false, M, N, K, A, K, 0, B, N, 0, C, N, 0, 1, 0);
}

gf_status gf_matmul_backward_f32(gf_ctx *ctx, const float *dC, const float *A, const float *B, float *dA, float *dB,
                                 int M, int K, int N, int accumulate) {
    gf_status st = need(ctx, dC && A && B && M > 0 && K > 0 && N > 0, "gf_matmul_backward_f32: null pointer or non-positive size");
    if (st != GF_OK) return st;
    if (dA) {  // dA[M,K] (+)= dC[M,N] B^T[N,K]
        st = gf::gemm(ctx, false, true, M, K, N, dC, N, 0, B, N, 0, dA, K, 0, 1, accumulate);
        if (st != GF_OK) return st;
    }
    if (dB)  // dB[K,N] (+)= A^T[K,M] dC[M,N]
        st = gf::gemm(ctx, true, false, K, N, M, A, K, 0, dC, N, 0, dB, N, 0, 1, accumulate);
    return st;
}

gf_status gf_mattensormul_forward_f32(gf_ctx *ctx, const float *X, const float *F, float *Out, int R, int Kd, int J, int D) {
    gf_status st = need(ctx, X && F && Out && R > 0 && Kd > 0 && J > 0 && D > 0, "gf_mattensormul_forward_f32: bad argument");
    if (st != GF_OK) return st;
    return gf::gemm(ctx, false, false, R, J * D, Kd, X, Kd, 0, F, J * D, 0, Out, J * D, 0, 1, 0);
}

gf_status gf_mattensormul_backward_f32(gf_ctx *ctx, const float *G, const float *X, const float *F, float *dX, float *dF,
                                       int R, int Kd, int J, int D, int accumulate) {
    gf_status st = need(ctx, G && X && F && R > 0 && Kd > 0 && J > 0 && D > 0, "gf_mattensormul_backward_f32: bad argument");
    if (st != GF_OK) return st;
    const int JD = J * D;
    if (dX) {  // dX[R,Kd] (+)= G[R,JD] F^T[JD,Kd]
        st = gf::gemm(ctx, false, true, R, Kd, JD, G, JD, 0, F, JD, 0, dX, Kd, 0, 1, accumulate);
        if (st != GF_OK) return st;
    }
    if (dF)  // dF[Kd,JD] (+)= X^T[Kd,R] G[R,JD]
        st = gf::gemm(ctx, true, false, Kd, JD, R, X, Kd, 0, G, JD, 0, dF, JD, 0, 1, accumulate);
    return st;
}

gf_status gf_tensormatmul_forward_f32(gf_ctx *ctx, const float *F, const float *Y, float *Out, int R, int Kd, int J, int D) {
    gf_status st = need(ctx, F && Y && Out && R > 0 && Kd > 0 && J > 0 && D > 0, "gf_tensormatmul_forward_f32: bad argument");
    if (st != GF_OK) return st;
    // Out_i[J,D] = Y^T[J,Kd] F_i[Kd,D]
    return gf::gemm(ctx, true, false, J, D, Kd, Y, J, 0, F, D, (long long)Kd * D, Out, D, (long long)J * D, R, 0);
}

gf_status gf_tensormatmul_backward_f32(gf_ctx *ctx, const float *G, const float *F, const float *Y, float *dF, float *dY,
                                       int R, int Kd, int J, int D, int accumulate) {
    gf_status st = need(ctx, G && F && Y && R > 0 && Kd > 0 && J > 0 && D > 0, "gf_tensormatmul_backward_f32: bad argument");
    if (st != GF_OK) return st;
    if (dF) {  // dF_i[Kd,D] (+)= Y[Kd,J] G_i[J,D]
        st = gf::gemm(ctx, false, false, Kd, D, J, Y, J, 0, G, D, (long long)J * D, dF, D, (long long)Kd * D, R, accumulate);
        if (st != GF_OK) return st;
    }
    if (dY) {  // dY[Kd,J] (+)= sum_i F_i[Kd,D] G_i^T[D,J]: one launch per i, accumulating in i order (deterministic)
        for (int i = 0; i < R; ++i) {
            st = gf::gemm(ctx, false, true, Kd, J, D, F + (size_t)i * Kd * D, D, 0, G + (size_t)i * J * D, D, 0, dY, J, 0, 1,
                          (i > 0 || accumulate) ? 1 : 0);
            if (st != GF_OK) return st;
        }
    }
    return st;
}

gf_status gf_stack_forward_f32(gf_ctx *ctx, const float *const *tensors, float *out, int nRows, size_t per_tensor) {
    gf_status st = need(ctx, tensors && out && nRows > 0, "gf_stack_forward_f32: bad argument");
    if (st != GF_OK) return st;
    return gf::stack_forward(ctx, tensors, out, nRows, per_tensor);
}

gf_status gf_stack_backward_f32(gf_ctx *ctx, const float *G, float *const *grads, int nRows, size_t per_tensor) {
    gf_status st = need(ctx, G && grads && nRows > 0, "gf_stack_backward_f32: bad argument");
    if (st != GF_OK) return st;
    return gf::stack_backward(ctx, G, grads, nRows, per_tensor);
}

}  // extern "C"
