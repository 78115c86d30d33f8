// smp_level_ops.hip -- the block products of the fused SMP level at C = 64 as stand-alone operators of the C ABI
// (gf_smp_level_products_f32 / gf_smp_level_wgrad_f32, include/gf_hip.h): the same kernels gf_smp_forward / gf_smp_backward launch
// on a level's rows (smp_level_c64.hip on the fp32 matrix pipe, smp_level_c64_split.hip on the f16 pipe with two-half operands),
// on caller-supplied matrices.  They replace, for the rows of one level, the K-projection MatMul of the reference
// (GraphFlow/SMP_omega.h:654-657 forward, MatMul.h:69-82 backward) in its regrouped form (smp_fused.hip header) -- and they are
// what the parity suite uses to hold the split-operand arithmetic to the fp64 product per output ROW and per weight-gradient ROW,
// on operands whose dynamic range the whole-network tests cannot steer (tests/test_level_ops_gpu.py).
#include <cstring>

#include "smp_internal.h"

namespace gf {
namespace {

// largest magnitudes (float bits) of T's four 64-column blocks and dO's two, into words [0, 6) of copy 0 of a blkmax table
// (the layout the producers of a level keep, smp_internal.h: kBlkCopies copies, kBlkStride words apart), and of the two columns of
// the row-factor table into words [6, 8)
__global__ __launch_bounds__(256) void level_block_maxima(const float *__restrict__ T, const float *__restrict__ dO,
                                                          const float *__restrict__ rs, int rows, unsigned *__restrict__ out) {
    __shared__ unsigned red[8];
    if (threadIdx.x < 8) red[threadIdx.x] = 0u;
    __syncthreads();
    unsigned m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto up = [](unsigned &a, float v) {
        const unsigned b = __float_as_uint(v) & 0x7fffffffu;
        a = b > a ? b : a;
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * 256; i += (size_t)gridDim.x * blockDim.x)
        up(m[(i & 255) >> 6], T[i]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * 128; i += (size_t)gridDim.x * blockDim.x)
        up(m[4 + ((i & 127) >> 6)], dO[i]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * 2; i += (size_t)gridDim.x * blockDim.x)
        up(m[6 + (i & 1)], rs[i]);
    for (int k = 0; k < 8; ++k) atomicMax(&red[k], m[k]);
    __syncthreads();
    if (threadIdx.x < 8) atomicMax(&out[threadIdx.x], red[threadIdx.x]);
}

}  // namespace
}  // namespace gf

using gf::fail;

extern "C" {

gf_status gf_smp_level_products_f32(gf_ctx *ctx, int backward, int rows, const float *A, const float *rowscale, const float *Wst,
                                    const int *trow, float *Out) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (rows < 0 || (rows > 0 && (!A || !rowscale || !Wst || !trow || !Out)))
        return fail(ctx, GF_ERR_INVALID, "gf_smp_level_products_f32: null argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return gf::smp_rowpanel_products_c64(ctx, backward == 0, A, rowscale, Wst, Out, rows, trow);
}

gf_status gf_smp_level_wgrad_f32(gf_ctx *ctx, int rows, const float *T, const float *dO, const float *rowscale, const int *trow,
                                 float *dWst) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (rows < 1 || !T || !dO || !rowscale || !trow || !dWst) return fail(ctx, GF_ERR_INVALID, "gf_smp_level_wgrad_f32: bad argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t total = 8 * 4096, maxima_words = (size_t)gf::kBlkCopies * gf::kBlkStride;
    // workspace: up to 256 + 8 partial images, then the block-maxima table
    const size_t img_floats = (size_t)(264 + 16) * total;
    gf_status st = gf::ensure_ws(ctx, sizeof(float) * (img_floats + maxima_words) + 256);
    if (st != GF_OK) return st;
    float *ws = static_cast<float *>(ctx->ws);
    unsigned *bm = reinterpret_cast<unsigned *>(ws + img_floats);
    GF_HIP_TRY(ctx, hipMemsetAsync(bm, 0, sizeof(unsigned) * maxima_words, ctx->stream));
    GF_LAUNCH(ctx, "level_block_maxima", gf::level_block_maxima, dim3(256), dim3(256), 0, T, dO, rowscale, rows, bm);
    unsigned host_max[8];
    GF_HIP_TRY(ctx, hipMemcpyAsync(host_max, bm, sizeof host_max, hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float max_tot, max_tr;
    std::memcpy(&max_tot, &host_max[6], 4);
    std::memcpy(&max_tr, &host_max[7], 4);
    GF_HIP_TRY(ctx, hipMemsetAsync(bm + 6, 0, 2 * sizeof(unsigned), ctx->stream));  // (words [6, 8) are not part of the table)
    gf::FoldGroup fg;
    st = gf::smp_wgrad_partials_c64(ctx, T, dO, rowscale, rows, ws, (size_t)264 * total, &fg, trow, bm, max_tot, max_tr);
    if (st != GF_OK) return st;
    return gf::splitk_fold(ctx, fg.part, dWst, total, fg.splits, 0);
}

}  // extern "C"
