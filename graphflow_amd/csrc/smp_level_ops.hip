// smp_level_ops.hip -- the block products of the fused SMP level at C = 64 as stand-alone operators of the C ABI
// (gf_smp_level_products_f32 / gf_smp_level_wgrad_f32, include/gf_hip.h): the same kernels gf_smp_forward / gf_smp_backward launch
// on a level's rows (smp_level_c64.hip on the fp32 matrix pipe, smp_level_c64_split.hip on the f16 pipe with two-half operands),
// on caller-supplied matrices.  They replace, for the rows of one level, the K-projection MatMul of the reference
// (GraphFlow/SMP_omega.h:654-657 forward, MatMul.h:69-82 backward) in its regrouped form (smp_fused.hip header) -- and they are
// what the parity suite uses to hold the split-operand arithmetic to the fp64 product per output ROW and per weight-gradient ROW,
// on operands whose dynamic range the whole-network tests cannot steer (tests/test_level_ops_gpu.py).
#include <cstring>

#include "smp_internal.h"

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
    const size_t total = 8 * 4096, bound_words = gf::smp_wgrad_bound_words_exact();
    // workspace: up to 256 + 8 partial images, then the column bounds of the operands (exact maxima: there is no level behind them)
    const size_t img_floats = (size_t)(264 + 16) * total;
    gf_status st = gf::ensure_ws(ctx, sizeof(float) * (img_floats + bound_words) + 256);
    if (st != GF_OK) return st;
    float *ws = static_cast<float *>(ctx->ws);
    unsigned *bw = reinterpret_cast<unsigned *>(ws + img_floats);
    st = gf::smp_wgrad_column_bounds_exact(ctx, T, dO, rowscale, rows, bw);
    if (st != GF_OK) return st;
    gf::FoldGroup fg;
    gf::WgradScales sc;
    sc.cmax = bw + 512;
    st = gf::smp_wgrad_partials_c64(ctx, T, dO, rowscale, rows, ws, (size_t)264 * total, &fg, trow, sc);
    if (st != GF_OK) return st;
    return gf::splitk_fold(ctx, fg.part, dWst, total, fg.splits, 0);
}

}  // extern "C"
