// head.hip -- the fully-connected head of GraphFlow's `_physics` / `_pairgraphs` SMP models on the device:
//   x = concatenated level features of the tower(s)                           (ConcatVectors, SMP_omega_physics.h:590)
//   h_1 = LeakyReLU(W_1 x), ..., h_k = LeakyReLU(W_k h_{k-1})                  (MatVecMul + LeakyReLU, :592-597; two layers in
//                                                                              SMP_omega_pairgraphs.h)
//   y = <h_k, w>,  loss = (y - t)^2 / 2                                        (InnerProduct + SquaredLoss, :599-603)
// for a batch of molecules at once: every MatVecMul of the batch is one GEMM on the fp32 MFMA kernel of mixers.hip.
#include "smp_internal.h"

namespace gf {
namespace {

constexpr float kAlphaH = 0.01f;  // LeakyReLU.h default

__global__ void head_lrelu(const float *__restrict__ z, float *__restrict__ h, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        h[i] = z[i] > 0.f ? z[i] : kAlphaH * z[i];
}
// dz = dh * lrelu'(z)
__global__ void head_lrelu_bwd(const float *__restrict__ z, const float *__restrict__ dh, float *__restrict__ dz, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dz[i] = dh[i] * (z[i] > 0.f ? 1.f : kAlphaH);
}
// y[m] = <h[m], w>; loss, dy
__global__ void head_inner(const float *__restrict__ h, const float *__restrict__ w, const float *__restrict__ target,
                           float *__restrict__ y, float *__restrict__ loss, float *__restrict__ dy, int width) {
    __shared__ float red[64];
    const int m = blockIdx.x;
    float acc = 0.f;
    for (int f = threadIdx.x; f < width; f += 64) acc += h[(size_t)m * width + f] * w[f];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 64; ++k) t += red[k];
        const float tg = target ? target[m] : 0.f;
        if (y) y[m] = t;
        if (loss) loss[m] = 0.5f * (t - tg) * (t - tg);
        dy[m] = t - tg;
    }
}
// dw[f] += sum_m dy[m] h[m][f];  dh[m][f] = dy[m] w[f].  One workgroup (deterministic): thread = (column f, molecule group g), group g
// takes the molecules g, g + ng, ... and the groups' partial sums are folded in group order through LDS.  (Round 4: with one thread per
// column -- ~30 columns -- the 1024 dependent loads of a batch took 205 us of a 4.6 ms physics step.)
__global__ __launch_bounds__(256) void head_inner_bwd(const float *__restrict__ dy, const float *__restrict__ h, const float *__restrict__ w,
                                                      float *__restrict__ dw, float *__restrict__ dh, int width, int n) {
    __shared__ float part[256];
    for (int f0 = 0; f0 < width; f0 += 256) {   // (wider than 256 columns: in chunks)
        const int wc = (width - f0 < 256) ? width - f0 : 256;   // columns of this chunk
        const int ng = 256 / wc;                                // molecule groups
        const int g = threadIdx.x / wc, f = f0 + threadIdx.x % wc;
        float acc = 0.f;
        if (g < ng) {
            const float wf = w[f];
            for (int m = g; m < n; m += ng) {
                const float d = dy[m];
                acc += d * h[(size_t)m * width + f];
                dh[(size_t)m * width + f] = d * wf;
            }
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if (g == 0) {
            float t = part[threadIdx.x];
            for (int k = 1; k < ng; ++k) t += part[k * wc + threadIdx.x];
            dw[f] += t;
        }
        __syncthreads();
    }
}

unsigned hgrid(size_t n) {
    const size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}
size_t work_floats(int nLayers, const int *width, int n) {
    size_t w = 0;
    for (int i = 1; i <= nLayers; ++i) w += (size_t)width[i];
    return 2 * (size_t)n * w + 2 * (size_t)n * (size_t)std::max(1, *std::max_element(width, width + nLayers + 1)) + (size_t)n;
}

}  // namespace
}  // namespace gf

using gf::fail;

extern "C" {

size_t gf_head_param_count(int nLayers, const int *width) {
    if (nLayers < 1 || !width) return 0;
    size_t n = 0;
    for (int i = 1; i <= nLayers; ++i) n += (size_t)width[i] * width[i - 1];
    return n + (size_t)width[nLayers];
}

size_t gf_head_work_floats(int nLayers, const int *width, int n) {
    if (nLayers < 1 || !width || n < 1) return 0;
    return gf::work_floats(nLayers, width, n);
}

// work layout: z_1 .. z_k (pre-activations), h_1 .. h_k, then two scratch rows of max width per molecule and dy [n]
gf_status gf_head_forward_f32(gf_ctx *ctx, int nLayers, const int *width, const float *x, int n, const float *params,
                              const float *targets, float *predict, float *loss, float *work) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (nLayers < 1 || nLayers > 8 || !width || !x || !params || !work || n < 1) return fail(ctx, GF_ERR_INVALID, "gf_head_forward_f32: bad argument");
    size_t zoff = 0, hbase = 0;
    for (int i = 1; i <= nLayers; ++i) hbase += (size_t)n * width[i];
    const float *in = x;
    const float *W = params;
    size_t hoff = hbase;
    for (int i = 1; i <= nLayers; ++i) {
        float *z = work + zoff, *h = work + hoff;
        gf_status st = gf::gemm(ctx, false, true, n, width[i], width[i - 1], in, width[i - 1], 0, W, width[i - 1], 0, z, width[i], 0, 1, 0);
        if (st != GF_OK) return st;
        const size_t cnt = (size_t)n * width[i];
        GF_LAUNCH(ctx, "head_lrelu", gf::head_lrelu, dim3(gf::hgrid(cnt)), dim3(256), 0, z, h, cnt);
        W += (size_t)width[i] * width[i - 1];
        in = h;
        zoff += cnt;
        hoff += cnt;
    }
    float *dy = work + gf::work_floats(nLayers, width, n) - n;
    GF_LAUNCH(ctx, "head_inner", gf::head_inner, dim3(n), dim3(64), 0, in, W, targets, predict, loss, dy, width[nLayers]);
    return GF_OK;
}

// after gf_head_forward_f32 with targets on the same `work`: dx [n][width[0]] = gradient of the summed loss w.r.t. the input rows,
// dparams += the weights' gradients (summed over the batch)
gf_status gf_head_backward_f32(gf_ctx *ctx, int nLayers, const int *width, const float *x, int n, const float *params, float *work,
                               float *dx, float *dparams) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (nLayers < 1 || nLayers > 8 || !width || !x || !params || !work || !dx || !dparams || n < 1)
        return fail(ctx, GF_ERR_INVALID, "gf_head_backward_f32: bad argument");
    std::vector<size_t> zo(nLayers + 1, 0), ho(nLayers + 1, 0), wo(nLayers + 2, 0);
    size_t hbase = 0;
    for (int i = 1; i <= nLayers; ++i) hbase += (size_t)n * width[i];
    {
        size_t z = 0, h = hbase, w = 0;
        for (int i = 1; i <= nLayers; ++i) {
            zo[i] = z;
            ho[i] = h;
            wo[i] = w;
            z += (size_t)n * width[i];
            h += (size_t)n * width[i];
            w += (size_t)width[i] * width[i - 1];
        }
        wo[nLayers + 1] = w;
    }
    const int wmax = std::max(1, *std::max_element(width, width + nLayers + 1));
    const size_t total = gf::work_floats(nLayers, width, n);
    float *dy = work + total - n, *sa = work + 2 * hbase, *sb = sa + (size_t)n * wmax;
    // dh_k and dw
    GF_LAUNCH(ctx, "head_inner_bwd", gf::head_inner_bwd, dim3(1), dim3(256), 0, dy, work + ho[nLayers], params + wo[nLayers + 1],
              dparams + wo[nLayers + 1], sa, width[nLayers], n);
    float *dh = sa, *other = sb;
    for (int i = nLayers; i >= 1; --i) {
        const size_t cnt = (size_t)n * width[i];
        GF_LAUNCH(ctx, "head_lrelu_bwd", gf::head_lrelu_bwd, dim3(gf::hgrid(cnt)), dim3(256), 0, work + zo[i], dh, dh, cnt);  // dz in place
        const float *in = (i == 1) ? x : work + ho[i - 1];
        // dW_i [w_i][w_{i-1}] += dz^T in ;   d(in) [n][w_{i-1}] = dz W_i
        gf_status st = gf::gemm(ctx, true, false, width[i], width[i - 1], n, dh, width[i], 0, in, width[i - 1], 0, dparams + wo[i],
                                width[i - 1], 0, 1, 1);
        if (st != GF_OK) return st;
        float *out = (i == 1) ? dx : other;
        st = gf::gemm(ctx, false, false, n, width[i - 1], width[i], dh, width[i], 0, params + wo[i], width[i - 1], 0, out, width[i - 1], 0, 1, 0);
        if (st != GF_OK) return st;
        other = dh;
        dh = out;
    }
    return GF_OK;
}

}  // extern "C"
