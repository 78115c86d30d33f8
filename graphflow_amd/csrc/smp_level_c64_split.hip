// smp_level_c64_split.hip -- the row-panel block products of the fused SMP level at C = 64 (compact O layout) on the f16 matrix
// pipe with fp32-grade operands: every fp32 operand x is carried as TWO halves
//     x 2^k = h + l,   h = rn_f16(x 2^k),  l = rn_f16(x 2^k - h)          (22 significant bits, k a per-row / per-block exponent)
// and a product a b is evaluated as  ah bh + ah bl + al bh  (three v_mfma_f32_32x32x16_f16 with fp32 accumulation; products of
// f16 values are exact in fp32, the dropped al bl is 2^-22 of the term).  Against an fp64 product of the same operands the result
// is as close as the fp32 MFMA's (both are dominated by the fp32 accumulation; measured in tests/test_split_numerics_gpu.py and
// held to the same 1e-5 bar by every SMP parity test), while 64 columns of reduction cost 12 MFMAs of 8 passes instead of 32 of
// 16: the three product kernels of a level stop being bound by the fp32 matrix pipe (0.71 of its 157 TF/s peak, 0.83 of what the
// sustained clock allows -- no headroom) and become HBM streams.
//
// Same work decomposition as smp_rowpanel_c64 (smp_level_c64.hip): the eight 64 x 64 weight blocks live in LDS for the life of
// the workgroup -- here as two f16 fragment images (h and l, 64 KB each, scaled by a per-block power of two) -- and every wave
// walks 32-row panels alone: a lane owns half a row of a 64-column block, finds the row's exponent, splits its 32 values in
// registers, and runs the panel's products out of registers and LDS.  Row factors (tot / tr of the node, the row's and the
// block's exponents) multiply the product's 32 x 64 result on its way into the output accumulator.
#include <cstdlib>

#include "gemm_lds.h"
#include "gf_internal.h"
#include "smp_internal.h"

namespace gf {
namespace {

using lds_image::f16v;
using lds_image::f4v;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int kSpThreads = 512;            // two waves per SIMD, 256 registers each
constexpr int kSpImg = 8 * 2 * 4 * 64;     // 16-byte fragment entries per image: [pos][column half][k chunk][lane]
constexpr size_t kSpLds = 2 * (size_t)kSpImg * 16 + 16 * sizeof(float) + (kSpThreads / 64) * 32 * sizeof(float);

// power-of-two scale that puts a magnitude with biased exponent e into [2^13, 2^14), and its inverse (e clamped: magnitudes below
// 2^-113 are flushed by the f16 conversion, which is what the fp32 product of such operands underflows to as well)
__device__ __forceinline__ void pow2_scale(unsigned maxbits, float *s, float *inv) {
    unsigned e = maxbits >> 23;
    e = e < 14u ? 14u : e;
    *s = __uint_as_float((267u - e) << 23);
    *inv = __uint_as_float((e - 13u) << 23);
}
// (a, b) 2^k -> halves
__device__ __forceinline__ void split_pair(float a, float b, float s, h2 *h, h2 *l) {
    const f2v x = {a * s, b * s};
    *h = __builtin_convertvector(x, h2);
    const f2v r = x - __builtin_convertvector(*h, f2v);
    *l = __builtin_convertvector(r, h2);
}

template <bool FWD>
__global__ __launch_bounds__(kSpThreads, 1) void smp_rowpanel_split(const float *__restrict__ A, const float *__restrict__ rs,
                                                                     const float *__restrict__ Wst, float *__restrict__ Out, int rows,
                                                                     const int *__restrict__ trow) {
    constexpr int LDA = FWD ? 256 : 128, LDOUT = FWD ? 128 : 256;
    extern __shared__ __attribute__((aligned(16))) uint4 sp_smem[];
    uint4 *imgH = sp_smem, *imgL = sp_smem + kSpImg;
    float *winv = reinterpret_cast<float *>(sp_smem + 2 * kSpImg);  // [8] 2^-k of the weight blocks
    unsigned *wmax = reinterpret_cast<unsigned *>(winv + 8);        // [8]
    float *facs = winv + 16;                                        // [waves][32]: row factors on their way to the C layout
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6;

    // ---- weight images: B fragment (pos, nh, c) gives lane (n, g) the eight values B[k = 32 g + 8 c + j][32 nh + n], j < 8, where
    // B = W_pos (forward) or W_pos^T (backward) -- the k order a lane's half row of A supplies (see split_blk)
    if (tid < 8) wmax[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int pos = 0; pos < 8; ++pos) {
        unsigned m = 0u;
#pragma unroll
        for (int i = 0; i < 4096 / kSpThreads; ++i) {
            const unsigned b = __float_as_uint(Wst[pos * 4096 + i * kSpThreads + tid]) & 0x7fffffffu;
            m = b > m ? b : m;
        }
        atomicMax(&wmax[pos], m);
    }
    __syncthreads();
    for (int t = tid; t < kSpImg; t += kSpThreads) {
        const int ln = t & 63, c = (t >> 6) & 3, nh = (t >> 8) & 1, pos = t >> 9;
        const int n = 32 * nh + (ln & 31), k0 = 32 * (ln >> 5) + 8 * c;
        float s, inv;
        pow2_scale(wmax[pos], &s, &inv);
        if (ln == 0 && c == 0 && nh == 0) winv[pos] = inv;
        const float *w = Wst + pos * 4096;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = FWD ? w[(k0 + j) * 64 + n] : w[n * 64 + k0 + j];
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h2 h, l;
            split_pair(v[2 * j], v[2 * j + 1], s, &h, &l);
            hw[j] = __builtin_bit_cast(unsigned, h);
            lw[j] = __builtin_bit_cast(unsigned, l);
        }
        imgH[t] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        imgL[t] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    __syncthreads();

    const int npanels = (rows + 31) / 32;
    const int nwaves = gridDim.x * (kSpThreads / 64);
    float *myfac = facs + wave * 32;

    struct Raw {
        f4v a[8];
    };
    struct Spl {
        uint4 h[4], l[4];  // eight f16 each
    };
    // columns [64 blk + 32 lh, +32) of row `li` of panel p (at the node's transposed row if `tr`).  Rows past the end read the last
    // row instead (unconditional loads: no branch per request); what is computed from them is never stored.
    auto load_raw = [&](Raw &R, int p, int blk, bool tr) {
        __builtin_amdgcn_sched_barrier(0);  // (requests stay where the schedule below puts them: hoisted to the top of the panel
                                            //  they would all be live at once)
        int row = p * 32 + li;
        row = row < rows ? row : rows - 1;
        const int src_row = tr ? trow[row] : row;
        const float *src = A + (size_t)src_row * LDA + blk * 64 + 32 * lh;
#pragma unroll
        for (int q = 0; q < 8; ++q) R.a[q] = *reinterpret_cast<const f4v *>(src + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_scale = [&](int p) {
        int row = p * 32 + li;
        row = row < rows ? row : rows - 1;
        return *reinterpret_cast<const float2 *>(rs + (size_t)row * 2);
    };
    // raw block -> halves at the row's exponent (one exponent for the 64 columns of the row: both lane halves agree on it);
    // MFMA c takes the lane's columns [8 c, 8 c + 8) as k = 8 (lane >> 5) + j
    auto split_blk = [&](const Raw &R, Spl &S, float &inv) {
        __builtin_amdgcn_sched_barrier(0);  // (not earlier than written: a block split ahead of time is 32 more live registers)
        unsigned m = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned b = __float_as_uint(R.a[q][j]) & 0x7fffffffu;
                m = b > m ? b : m;
            }
        const unsigned mo = (unsigned)__shfl_xor((int)m, 32);
        m = mo > m ? mo : m;
        float s;
        pow2_scale(m, &s, &inv);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h2 h, l;
                const f4v v = R.a[2 * c + (j >> 1)];
                split_pair(v[2 * (j & 1)], v[2 * (j & 1) + 1], s, &h, &l);
                hw[j] = __builtin_bit_cast(unsigned, h);
                lw[j] = __builtin_bit_cast(unsigned, l);
            }
            S.h[c] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            S.l[c] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // acc (two column halves) += rowfac * (S x block wpos).  `rowfac` is the lane's row factor (row li); the C/D layout of the
    // 32 x 32 MFMA wants it at rows (r & 3) + 8 (r >> 2) + 4 lh: through the wave's 32 floats of LDS (a wave's DS operations
    // execute in order).
    auto prod = [&](const Spl &S, float rowfac, int wpos, f16v &acc0, f16v &acc1) {
        __builtin_amdgcn_wave_barrier();
        myfac[li] = rowfac * winv[wpos];  // (both lane halves hold the row's factor: same value, same address, no branch)
        __builtin_amdgcn_wave_barrier();
        const uint4 *bh = imgH + (size_t)(wpos * 8) * 64 + lane, *bl = imgL + (size_t)(wpos * 8) * 64 + lane;
        // one column half at a time (sixteen registers of products in flight, not thirty-two: the panel's operand blocks and the
        // requests behind them take the rest of the wave's 256)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
            f16v t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const h8 bhc = __builtin_bit_cast(h8, bh[(4 * nh + c) * 64]), blc = __builtin_bit_cast(h8, bl[(4 * nh + c) * 64]);
                const h8 ah = __builtin_bit_cast(h8, S.h[c]), al = __builtin_bit_cast(h8, S.l[c]);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhc, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blc, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhc, t, 0, 0, 0);
            }
            f16v &acc = nh ? acc1 : acc0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4v fac = *reinterpret_cast<const f4v *>(myfac + 8 * g + 4 * lh);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * g + j] += t[4 * g + j] * fac[j];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto clear = [&](f16v &acc0, f16v &acc1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    };
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    auto store_out = [&](int p, int o, const f16v &acc0, const f16v &acc1) {
        const int r0 = p * 32;
        float *out = Out + (size_t)(r0 + 4 * lh) * LDOUT + o * 64 + li;
        if (r0 + 32 <= rows) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                out[(size_t)rr * LDOUT] = acc0[r];
                out[(size_t)rr * LDOUT + 32] = acc1[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (r0 + 4 * lh + rr < rows) {
                    out[(size_t)rr * LDOUT] = acc0[r];
                    out[(size_t)rr * LDOUT + 32] = acc1[r];
                }
            }
        }
    };

    // One panel.  On entry Ra and Rb hold (requests for) the panel's first two blocks; every other block is requested as soon as
    // a raw buffer has been split, one to three products (0.4 - 1 us) ahead of its use, and the first two blocks of the wave's
    // next panel go out behind the panel's last ones.
    float2 sc = load_scale(blockIdx.x * (kSpThreads / 64) + wave), scn = make_float2(0.f, 0.f);
    auto panel = [&](int p, Raw &Ra, Raw &Rb) {
        const int pn = p + nwaves;
        f16v acc0, acc1;
        Spl X, Y, Z;
        float iX, iY, iZ;
        if (FWD) {  // T blocks: 0 S_ab, 1 S_bc, 2 T6, 3 T10; outputs: 0 O_loc, 1 U.  Entry: Ra = S_ab, Rb = S_ab at the transposed rows
            split_blk(Ra, X, iX);
            load_raw(Ra, p, 1, false);               // S_bc
            split_blk(Rb, Z, iZ);
            load_raw(Rb, p, 2, false);               // T6
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, iX, 5, acc0, acc1);
            prod(Z, iZ, 7, acc0, acc1);
            split_blk(Ra, Y, iY);
            prod(Y, iY, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, iX * sc.x, 0, acc0, acc1);
            prod(X, iX * sc.y, 2, acc0, acc1);
            load_raw(Ra, p, 3, false);               // T10 (once S_ab's registers are free: X, Y and two requests do not fit)
            prod(Y, iY * sc.x, 1, acc0, acc1);
            split_blk(Rb, Z, iZ);
            load_raw(Rb, pn, 0, true);               // S_ab of the next panel at its transposed rows
            prod(Z, iZ, 3, acc0, acc1);
            split_blk(Ra, Z, iZ);
            load_raw(Ra, pn, 0, false);              // S_ab of the next panel
            prod(Z, iZ, 4, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        } else {    // dO blocks: 0 L, 1 dU; outputs: 0 dS_ab, 1 dS_bc, 2 dT6, 3 dT10.  Entry: Ra = L, Rb = dU
            split_blk(Ra, X, iX);
            load_raw(Ra, p, 1, true);                // dU at the transposed rows
            split_blk(Rb, Y, iY);
            load_raw(Rb, pn, 1, false);              // dU of the next panel
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, iX, 3, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, iX, 4, acc0, acc1);
            store_out(p, 3, acc0, acc1);
            clear(acc0, acc1);
            prod(X, iX * sc.x, 1, acc0, acc1);
            prod(Y, iY, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, iX * sc.x, 0, acc0, acc1);
            prod(X, iX * sc.y, 2, acc0, acc1);
            prod(Y, iY, 5, acc0, acc1);
            split_blk(Ra, Z, iZ);
            load_raw(Ra, pn, 0, false);              // L of the next panel
            prod(Z, iZ, 7, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        }
        sc = scn;
    };
    Raw R0, R1;
    int p = blockIdx.x * (kSpThreads / 64) + wave;
    load_raw(R0, p, 0, false);
    load_raw(R1, p, FWD ? 0 : 1, FWD);
    for (; p < npanels; p += nwaves) panel(p, R0, R1);
}

}  // namespace

bool smp_split_products() {  // (read per call: the parity tests switch it)
    const char *e = std::getenv("GF_SMP_SPLIT");
    return !(e && e[0] == '0');
}

// Row-panel products of a fused SMP level at C = 64, compact layout (O = [O_loc | U]; trow = the transposed-row table of the
// level): forward O from T = [S_ab|S_bc|T6|T10], or backward dT from dO.  Every output element is produced by one wave in a
// fixed order: results do not depend on the grid size.
gf_status smp_rowpanel_split_c64(gf_ctx *ctx, bool forward, const float *A, const float *rowscale, const float *Wst, float *Out,
                                 int rows, const int *trow, int cus) {
    const int per = kSpThreads / 64;
    const int npanels = (rows + 31) / 32;
    const int want = (npanels + per - 1) / per;
    const int grid = want < cus ? want : cus;  // one persistent workgroup per CU (the weight images take 128 KB of LDS)
    if (forward) {
        gf_status st = opt_in_lds(ctx, smp_rowpanel_split<true>, kSpLds);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smpf_products_fwd", (smp_rowpanel_split<true>), dim3((unsigned)grid), dim3(kSpThreads), kSpLds, A, rowscale, Wst,
                  Out, rows, trow);
    } else {
        gf_status st = opt_in_lds(ctx, smp_rowpanel_split<false>, kSpLds);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smpf_products_bwd", (smp_rowpanel_split<false>), dim3((unsigned)grid), dim3(kSpThreads), kSpLds, A, rowscale, Wst,
                  Out, rows, trow);
    }
    return GF_OK;
}

}  // namespace gf
