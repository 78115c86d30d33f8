// smp_level_c64_fwd.hip -- combine-forward of the fused SMP level at C = 64 on row panels (compact projected matrix O = [O_loc | U]):
//
//   f_l[x, y] = LeakyReLU( b + O_loc[x, y] + sum_e A+[y, e] U'[x, e] + r[y] V[x] + A+[x, y] S )          (smp_fused.hip, combine)
//   U' = U + G15[x, e] + G16[e, x]      (O_loc and U come from smp_rowpanel_split<true>, smp_level_c64_split.hip)
//
// A panel is a run of whole (node, x) row groups (s rows each, consecutive x of ONE node, at most 32 rows: table `pan`, built on the
// device at prepare time), so that everything combine needs across rows is inside one wave's registers.  In the C/D layout of the
// 32 x 32 MFMA a lane holds column (lane & 31) of sixteen rows; register r of the two lane halves holds exactly the two k-rows that
// step r of v_mfma_f32_32x32x2_f32 wants as its B operand, so  M += A' U'  (A' = the panel's block-diagonal gated adjacency, 32 x 32)
// is sixteen fp32 MFMAs per column half straight from registers loaded in that layout -- no transpose, no LDS.  The rank-one terms
// r[y] V[x], A+[x, y] S and the bias are nine more steps of the same instruction.  Exact fp32 products, fp32 accumulation.
//
// Rounds 3 and 4 also built the products AND this epilogue as ONE kernel (O never written or read back: 2 of the level's ~11.5
// row-blocks of forward traffic).  It is not kept: a panel's live state -- U and M accumulators (64 registers), one operand block as
// two f16 halves (32), the product in flight (16), two raw blocks on their way from HBM (64), the epilogue's gathered operands
// (16 + 32 + 32) -- is ~290 registers at its peak (the gather / A' U' stage); at two waves per SIMD (256) hipcc spills 150 - 220
// registers inside the panel loop whatever the order of the panel (round 4 tried: every block consumed before the next is split;
// U finished and folded into M before the O_loc products, with S_ab / S_bc read twice), and a scratch reload waits for the in-order
// memory queue, prefetches included; at one wave per SIMD (512) the round-3 build ran 2.4x slower than the two kernels.  DESIGN.md 9.
// Reference: GraphFlow/SMP_omega.h:654-669 (MatMul + VectorAddTensor + LeakyReLU3D of a level), regrouped as in smp_fused.hip.
#include <cstdlib>
#include <type_traits>

#include "gemm_lds.h"
#include "gf_internal.h"
#include "smp_internal.h"

namespace gf {
namespace {

using lds_image::f16v;
using lds_image::f4v;
constexpr float kAlphaFf = 0.01f;
constexpr int kFfOor = 0x40000000;  // a lane offset past every buffer this kernel addresses through a descriptor: the load returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ff_rsrc(const void *base, size_t bytes) {
    const unsigned n = bytes > 0xfffffffcull ? 0xfffffffcu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}
// (the value goes through a scalar first: __builtin_bit_cast applied directly to an element of an ext_vector -- bit_cast(unsigned, v[r]) --
//  compiled to element 0 for every r with this hipcc: round 5, every row of a tile stored register 0)
__device__ __forceinline__ void ff_st1(__amdgpu_buffer_rsrc_t r, int voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, 0);
}
__device__ __forceinline__ float ff_ld1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
// the streamed operands (O, f_l, df_l: read once) and results (f_l, dO: written once) -- aux bit 1 = nt (gf_internal.h: GF_NT_SITES 8 / 16)
__device__ __forceinline__ float ff_ld1s(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, (GF_NT_SITES & 8) ? 2 : 0));
}
__device__ __forceinline__ void ff_st1s(__amdgpu_buffer_rsrc_t r, int voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, (GF_NT_SITES & 16) ? 2 : 0);
}

// pan[p] = {first row of the panel, rows | s << 8 | groups << 16 | x0 << 24, first row of the node, first pair of the node}
// (x0 = index x of the panel's first row group inside its node); pan_node[p] = the node
// ---------------------------------------------------------------------------------------------------------------
// combine-forward on the same row panels, as a kernel of its own (default at C = 64, compact O): the epilogue of the kernel above
// with O_loc and U read from the projected matrix instead of computed.  One WAVE per panel, no LDS, no barrier: every operand of
// the panel -- its rows of O, the 64 gathered compact products, the lane's half row of the block-diagonal adjacency, the
// rank-one operands -- is requested up front (one round trip), the adjacency product and the rank-one terms are fp32 MFMAs on
// registers laid out as the MFMA's C/D tile, f_l leaves as 128-byte row segments.  Replaces smp_combine_fwd<16> (workgroup per
// (node, four x): adjacency image in LDS, a barrier between its two phases, a quarter of its waves idle on ragged quads).
// ---------------------------------------------------------------------------------------------------------------
// CB = channels: 64 (two 32-column halves per row) or 32 (one; round 4)
template <int CB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void smp_combine_fwd_panels(
    const float *__restrict__ O, float *__restrict__ F, const int4 *__restrict__ pan, const int *__restrict__ pan_node, int npanels,
    int rows, const int2 *__restrict__ goff, const float *__restrict__ Gc, long long gc_rows, const float *__restrict__ adj,
    const float *__restrict__ rsum, const float *__restrict__ Vout, long long pairs, const float *__restrict__ Sout,
    const float *__restrict__ bias, float *__restrict__ psum,   // psum (top level, or null): [npanels][64] column sums of the panel's
    // rows of f -- the readout's per-node sums (ShrinkTensor, SMP_omega.h:671-676) then read 22 MB of partials instead of f_L again
    float *__restrict__ pmax,  // pmax (or null): [npanels][64] largest |f| per column of the panel's rows (the level above scales the
    // columns of its weight-gradient operands with the level's per-channel maxima: smp_wgrad_column_bounds)
    const float *__restrict__ nodefac) {  // (or null) slice dropout: [nodes][18] factors; the compact products G15 / G16 take theirs here
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned blk;
    {
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = blockIdx.x % 8;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    }
    const int p = __builtin_amdgcn_readfirstlane((int)(blk * 4 + (threadIdx.x >> 6)));
    if (p >= npanels) return;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int4 Pv = pan[p];
    const int4 P = make_int4(rfl(Pv.x), rfl(Pv.y), rfl(Pv.z), rfl(Pv.w));
    const int node = rfl(pan_node[p]);
    const int nrows = P.y & 0xff, s = (P.y >> 8) & 0xff, G = (P.y >> 16) & 0xff, x0 = (P.y >> 24) & 0xff;
    const bool rowok = li < nrows;
    const int row = P.x + (rowok ? li : nrows - 1);
    const int2 go = goff[row];
    constexpr bool TWO = CB == 64;   // a second column half
    constexpr int OB = 2 * CB * 4, FB = CB * 4;   // bytes of a row of O = [O_loc | U] and of Gc = [G15 | G16], of f and of Vout
    const __amdgpu_buffer_rsrc_t rO = ff_rsrc(O, (size_t)rows * 2 * CB * sizeof(float));
    const __amdgpu_buffer_rsrc_t rGc = ff_rsrc(Gc, (size_t)gc_rows * 2 * CB * sizeof(float));
    const __amdgpu_buffer_rsrc_t rAdj = ff_rsrc(adj, (size_t)rows * sizeof(float));
    const __amdgpu_buffer_rsrc_t rV = ff_rsrc(Vout, (size_t)pairs * CB * sizeof(float));
    const __amdgpu_buffer_rsrc_t rF = ff_rsrc(F, (size_t)rows * CB * sizeof(float));
    const int gs = (int)((li + 0.5f) * __builtin_amdgcn_rcpf((float)s));
    const int y_li = li - gs * s;
    // byte offset of the lane's column inside a row block; CB = 16 (round 5): the lanes of columns 16..31 address nothing
    const int colb = (CB >= 32 || li < CB) ? li * 4 : kFfOor;
    // operands that do not depend on the gather indices first
    f16v m0, m1, u0, u1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * OB + colb : kFfOor;
        m0[r] = ff_ld1s(rO, vo);
        u0[r] = ff_ld1s(rO, vo + FB);
        if constexpr (TWO) {
            m1[r] = ff_ld1s(rO, vo + 128);
            u1[r] = ff_ld1s(rO, vo + FB + 128);
        } else {
            m1[r] = u1[r] = 0.f;
        }
    }
    float aval[16], vb0[5], vb1[5];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int er = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int e = er - gs * s;
        const bool in = rowok && e >= 0 && e < s && er < nrows;
        aval[r] = ff_ld1(rAdj, in ? (P.z + y_li * s + e) * 4 : kFfOor);
    }
    {
        const float *p0 = lh ? bias : Sout + (size_t)node * CB;
        vb0[0] = (CB >= 32 || li < CB) ? p0[CB >= 32 ? li : (li & (CB - 1))] : 0.f;
        vb1[0] = TWO ? p0[(TWO ? 32 : 0) + li] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int g = 2 * t + lh;
            const int vo = g < G ? (int)(((long long)P.w + x0 + g) * FB) + colb : kFfOor;
            vb0[t + 1] = ff_ld1(rV, vo);
            vb1[t + 1] = TWO ? ff_ld1(rV, vo + 128) : 0.f;
        }
    }
    const float axy = rowok ? adj[row] : 0.f;
    const float r_y = rowok ? rsum[(size_t)P.w + y_li] : 0.f;
    // the compact products G15[x, e] + G16[e, x] of the panel's rows
    float g0[16], g1[16], g2[16], g3[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        const int a15 = __builtin_amdgcn_readlane(go.x, rr), b15 = __builtin_amdgcn_readlane(go.x, rr + 4);
        const int a16 = __builtin_amdgcn_readlane(go.y, rr), b16 = __builtin_amdgcn_readlane(go.y, rr + 4);
        const int i15 = lh ? b15 : a15, i16 = lh ? b16 : a16;
        const int v15 = i15 < 0 ? kFfOor : i15 * OB + colb, v16 = i16 < 0 ? kFfOor : i16 * OB + FB + colb;
        g0[r] = ff_ld1(rGc, v15);
        g1[r] = ff_ld1(rGc, v16);
        g2[r] = TWO ? ff_ld1(rGc, v15 + 128) : 0.f;
        g3[r] = TWO ? ff_ld1(rGc, v16 + 128) : 0.f;
    }
    // (a panel is one node's.  Round 5: with these two run-time factors on the gathered operands the compiler stopped folding the gathers
    //  into the sums as they arrive and keeps all 64 in registers: 180 + 16 registers instead of 132 + 16, two waves per SIMD instead of
    //  three -- and the kernel went from 0.66 to 0.585 ms per cfg3 step, every request of a panel now being in flight at once.)
    const float c15 = nodefac ? nodefac[(size_t)node * 18 + 15] : 1.f, c16 = nodefac ? nodefac[(size_t)node * 18 + 16] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u0[r] += c15 * g0[r] + c16 * g1[r];
        u1[r] += c15 * g2[r] + c16 * g3[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float a = aval[r] > 0.f ? aval[r] : 0.f;  // the gate of RisiContraction_18 (RisiContraction_18.h:90)
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, u0[r], m0, 0, 0, 0);
        if constexpr (TWO) m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, u1[r], m1, 0, 0, 0);
    }
    {
        const float a0 = lh ? 1.f : (axy > 0.f ? axy : 0.f);
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb0[0], m0, 0, 0, 0);
        if constexpr (TWO) m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb1[0], m1, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (2 * t < G) {  // (uniform; nothing is requested in here)
                const float a = (2 * t + lh) == gs ? r_y : 0.f;
                m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb0[t + 1], m0, 0, 0, 0);
                if constexpr (TWO) m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb1[t + 1], m1, 0, 0, 0);
            }
        }
    }
    float c0 = 0.f, c1 = 0.f, x0m = 0.f, x1m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * FB + colb : kFfOor;
        const float z0 = m0[r], z1 = m1[r];
        const float f0 = z0 > 0.f ? z0 : kAlphaFf * z0, f1 = z1 > 0.f ? z1 : kAlphaFf * z1;
        ff_st1s(rF, vo, f0);
        if constexpr (TWO) ff_st1s(rF, vo + 128, f1);
        if (rr < nrows) {   // (rows in register order: a fixed order)
            c0 += f0, c1 += f1;
            x0m = fmaxf(x0m, fabsf(f0)), x1m = fmaxf(x1m, fabsf(f1));
        }
    }
    if (psum) {  // (uniform)
        c0 += __shfl_xor(c0, 32);
        c1 += __shfl_xor(c1, 32);
        if (TWO || lane < CB) psum[(size_t)p * CB + lane] = lh ? c1 : c0;   // lane = 32 lh + li: columns li | 32 + li
    }
    if (pmax) {  // (uniform)
        x0m = fmaxf(x0m, __shfl_xor(x0m, 32));
        x1m = fmaxf(x1m, __shfl_xor(x1m, 32));
        if (TWO || lane < CB) pmax[(size_t)p * CB + lane] = lh ? x1m : x0m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// combine-backward on the same row panels (round 5; it was a workgroup per (node, four x) with the adjacency image and the panel's dz
// in LDS behind two barriers: 4.0 TB/s on its bytes).  One WAVE per panel, every request up front, C/D-tile registers throughout:
//   dz[x, y]  = dF[x, y] * LeakyReLU'(f[x, y])                                   -> block L of dO  (and the level's bias / weight gradients)
//   dU[x, e]  = sum_y A+[y, e] dz[x, y]                                           -> block dU of dO : sixteen fp32 MFMA steps per column
//               half with A'^T (the panel's block-diagonal gated adjacency, transposed) as the A operand and dz as the B operand
//   dVout[x]  = sum_y r[y] dz[x, y]     dSpart[x] = sum_y A+[x, y] dz[x, y]     dbpart[x] = sum_y dz[x, y]
//               one more MFMA chain: output row (type t, group g) of a 32 x 32 tile takes weight  [row e in group g] * {r[y_e], A+[x_g, y_e], 1};
//               the three weights of an input row are what ITS lane holds for the forward pass (r[y], A+[x, y]), read with v_readlane
// and the per-column maxima of |dz| over the panel (the weight gradients' column exponents).  Same sums per output element as
// smp_combine_bwd (fixed order: the MFMA's k order), held to it by the parity tests (GF_SMP_COMBINE_BWD_PANELS=0 selects it).
// ---------------------------------------------------------------------------------------------------------------
template <int CB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void smp_combine_bwd_panels(
    const float *__restrict__ F, const float *__restrict__ dF,   // dF null: the gradient is node_dF's vector at every row of the node
    const float *__restrict__ node_dF,                            // [nodes][CB] or null; with dF: both are added (a tower's level below the top)
    float *__restrict__ dO, const int4 *__restrict__ pan, const int *__restrict__ pan_node, int npanels, int rows,
    const float *__restrict__ adj, const float *__restrict__ rsum, long long pairs, float *__restrict__ dVout, float *__restrict__ dSpart,
    float *__restrict__ dbpart, float *__restrict__ dzmax) {   // dzmax (or null): [npanels][CB]
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned blk;
    {
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = blockIdx.x % 8;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    }
    const int p = __builtin_amdgcn_readfirstlane((int)(blk * 4 + (threadIdx.x >> 6)));
    if (p >= npanels) return;
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int4 Pv = pan[p];
    const int4 P = make_int4(rfl(Pv.x), rfl(Pv.y), rfl(Pv.z), rfl(Pv.w));
    const int node = rfl(pan_node[p]);
    const int nrows = P.y & 0xff, s = (P.y >> 8) & 0xff, G = (P.y >> 16) & 0xff, x0 = (P.y >> 24) & 0xff;
    const bool rowok = li < nrows;
    const int row = P.x + (rowok ? li : nrows - 1);
    constexpr bool TWO = CB == 64;
    constexpr int OB = 2 * CB * 4, FB = CB * 4;
    const __amdgpu_buffer_rsrc_t rF = ff_rsrc(F, (size_t)rows * CB * sizeof(float));
    const __amdgpu_buffer_rsrc_t rG = ff_rsrc(dF ? dF : F, dF ? (size_t)rows * CB * sizeof(float) : 0);   // (no dF: every load returns 0)
    const __amdgpu_buffer_rsrc_t rAdj = ff_rsrc(adj, (size_t)rows * sizeof(float));
    const __amdgpu_buffer_rsrc_t rO = ff_rsrc(dO, (size_t)rows * 2 * CB * sizeof(float));
    const int gs = (int)((li + 0.5f) * __builtin_amdgcn_rcpf((float)s));
    const int y_li = li - gs * s;
    const bool colok = CB >= 32 || li < CB;   // (CB = 16: the lanes of columns 16..31 address nothing)
    const int colb = colok ? li * 4 : kFfOor;
    // every operand of the panel, requested up front
    f16v f0, f1, g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * FB + colb : kFfOor;
        f0[r] = ff_ld1s(rF, vo);
        g0[r] = ff_ld1s(rG, vo);
        if constexpr (TWO) {
            f1[r] = ff_ld1s(rF, vo + 128);
            g1[r] = ff_ld1s(rG, vo + 128);
        } else {
            f1[r] = g1[r] = 0.f;
        }
    }
    float avT[16];   // A+[y_e, y_li] of the lane's group: the lane's half row of A'^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int er = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int e = er - gs * s;
        const bool in = rowok && e >= 0 && e < s && er < nrows;
        avT[r] = ff_ld1(rAdj, in ? (P.z + e * s + y_li) * 4 : kFfOor);
    }
    const float gn0 = (node_dF && colok) ? node_dF[(size_t)node * CB + (CB >= 32 ? li : (li & (CB - 1)))] : 0.f;
    const float gn1 = (TWO && node_dF) ? node_dF[(size_t)node * CB + 32 + li] : 0.f;
    const float axy_raw = rowok ? adj[row] : 0.f;
    const float axy = axy_raw > 0.f ? axy_raw : 0.f;                  // A+[x, y] of the lane's own row
    const float r_y = rowok ? rsum[(size_t)P.w + y_li] : 0.f;           // r[y] of the lane's own row
    // dz in the C/D layout; rows past the panel are zeroed (a node-vector gradient would otherwise leak into them)
    f16v z0, z1;
    float x0m = 0.f, x1m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = rr < nrows;
        const float a = (g0[r] + gn0) * (f0[r] > 0.f ? 1.f : kAlphaFf), b = (g1[r] + gn1) * (f1[r] > 0.f ? 1.f : kAlphaFf);
        z0[r] = ok ? a : 0.f;
        z1[r] = ok ? b : 0.f;
        x0m = fmaxf(x0m, fabsf(z0[r]));
        x1m = fmaxf(x1m, fabsf(z1[r]));
        const int vo = ok ? (P.x + rr) * OB + colb : kFfOor;
        ff_st1s(rO, vo, z0[r]);
        if constexpr (TWO) ff_st1s(rO, vo + 128, z1[r]);
    }
    // dU = A'^T dz
    f16v u0, u1;
#pragma unroll
    for (int r = 0; r < 16; ++r) u0[r] = u1[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float a = avT[r] > 0.f ? avT[r] : 0.f;   // the gate of RisiContraction_18 (RisiContraction_18.h:345)
        u0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z0[r], u0, 0, 0, 0);
        if constexpr (TWO) u1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z1[r], u1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * OB + FB + colb : kFfOor;
        ff_st1s(rO, vo, u0[r]);
        if constexpr (TWO) ff_st1s(rO, vo + 128, u1[r]);
    }
    // the per-(node, x) partials: tile row (t, g) = 8 t + g, weight of input row e: [group of e == g] * {r[y_e], A+[x_g, y_e], 1}
    f16v q0, q1;
#pragma unroll
    for (int r = 0; r < 16; ++r) q0[r] = q1[r] = 0.f;
    {
        const int tt = li >> 3, tg = li & 7;   // the lane's output row as the A operand: (type, group)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ea = (r & 3) + 8 * (r >> 2);   // input row of the low lane half; the high half supplies row ea + 4
            const int ga = __builtin_amdgcn_readlane(gs, ea), gb = __builtin_amdgcn_readlane(gs, ea + 4);
            const float ra = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r_y), ea));
            const float rb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r_y), ea + 4));
            const float aa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, axy), ea));
            const float ab = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, axy), ea + 4));
            const int ge = lh ? gb : ga;
            const float rv = lh ? rb : ra, av = lh ? ab : aa;
            const bool hit = tg == ge && (ea + 4 * lh) < nrows && tg < G;
            const float w = !hit ? 0.f : tt == 0 ? rv : tt == 1 ? av : tt == 2 ? 1.f : 0.f;
            q0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, z0[r], q0, 0, 0, 0);
            if constexpr (TWO) q1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, z1[r], q1, 0, 0, 0);
        }
    }
    {
        const __amdgpu_buffer_rsrc_t rV = ff_rsrc(dVout, (size_t)pairs * CB * sizeof(float));
        const __amdgpu_buffer_rsrc_t rS = ff_rsrc(dSpart, (size_t)pairs * CB * sizeof(float));
        const __amdgpu_buffer_rsrc_t rB = ff_rsrc(dbpart, (size_t)pairs * CB * sizeof(float));
#pragma unroll
        for (int r = 0; r < 12; ++r) {   // tile rows (r & 3) + 8 (r >> 2) + 4 lh: type r >> 2, group (r & 3) + 4 lh
            const int g = (r & 3) + 4 * lh;
            const int vo = g < G ? (int)(((long long)P.w + x0 + g) * FB) + colb : kFfOor;
            const __amdgpu_buffer_rsrc_t rd = (r >> 2) == 0 ? rV : (r >> 2) == 1 ? rS : rB;
            ff_st1(rd, vo, q0[r]);
            if constexpr (TWO) ff_st1(rd, vo + 128, q1[r]);
        }
    }
    if (dzmax) {  // (uniform)
        x0m = fmaxf(x0m, __shfl_xor(x0m, 32));
        x1m = fmaxf(x1m, __shfl_xor(x1m, 32));
        if (TWO || lane < CB) dzmax[(size_t)p * CB + lane] = lh ? x1m : x0m;
    }
}

// ---- per-level tables of the kernel above, built on the device at prepare time
// node_panel[n] = first panel of node n; a node of size s has ceil(s / gpp) panels of gpp = max(1, 32 / s) row groups
__global__ void build_fwd_panels(const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                 const long long *__restrict__ node_pair, const int *__restrict__ node_panel, int4 *__restrict__ pan,
                                 int *__restrict__ pan_node) {
    const int n = blockIdx.x, s = node_s[n];
    if (s > 32) return;   // (no panels: smp_prep.cpp)
    const int gpp = s >= 32 ? 1 : (32 / s > 8 ? 8 : 32 / s), np = (s + gpp - 1) / gpp;  // (at most eight groups: the rank-one slots)
    for (int j = threadIdx.x; j < np; j += blockDim.x) {
        const int x0 = j * gpp, g = (s - x0 < gpp) ? s - x0 : gpp;
        const int p = node_panel[n] + j;
        pan[p] = make_int4((int)(node_row[n] + (long long)x0 * s), (g * s) | (s << 8) | (g << 16) | (x0 << 24), (int)node_row[n], (int)node_pair[n]);
        pan_node[p] = n;
    }
}
// goff[row of (x, e)] = {Gc row of G15[x, e], Gc row of G16[e, x]} (pair index of level l-1, -1 = structurally absent)
__global__ void build_fwd_goff(const int *__restrict__ node_s, const long long *__restrict__ node_row, const long long *__restrict__ node_pair,
                               const long long *__restrict__ pair_src_pair, const short *__restrict__ pi, int2 *__restrict__ goff) {
    const int n = blockIdx.x, s = node_s[n];
    const long long r0 = node_row[n], p0 = node_pair[n];
    for (int i = threadIdx.x; i < s * s; i += blockDim.x) {
        const int x = i / s, e = i - x * s;
        const int pxe = pi[r0 + (long long)x * s + e], pex = pi[r0 + (long long)e * s + x];
        goff[r0 + i] = make_int2(pxe >= 0 ? (int)(pair_src_pair[p0 + x] + pxe) : -1, pex >= 0 ? (int)(pair_src_pair[p0 + e] + pex) : -1);
    }
}

}  // namespace

// gather_offsets = false: goff is in place already (build_node_tables, smp.hip)
gf_status smp_fwd_fused_build_tables(gf_smp *s, int l, hipStream_t stream, bool gather_offsets) {
    gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    if (!d.fwd_pan || h.nNodes == 0) return GF_OK;
    hipLaunchKernelGGL(build_fwd_panels, dim3((unsigned)h.nNodes), dim3(64), 0, stream, d.node_s, d.node_row, d.node_pair, d.node_panel,
                       d.fwd_pan, d.fwd_pan_node);
    if (gather_offsets)
        hipLaunchKernelGGL(build_fwd_goff, dim3((unsigned)h.nNodes), dim3(64), 0, stream, d.node_s, d.node_row, d.node_pair, d.pair_src_pair,
                           d.pi, d.fwd_goff);
    GF_LAUNCH_CHECK(s->ctx, "build_fwd_panels");
    return GF_OK;
}

// f_l from the projected matrix O = [O_loc | U] (compact layout) of a fused level at C = 64: smp_combine_fwd_panels
gf_status smp_combine_fwd_panels_c64(gf_smp *s, int l, const float *O, const float *bias, float *psum, float *pmax, const float *nodefac) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int npanels = d.fwd_npanels;
    if (npanels < 1) return GF_OK;
#define GF_CFP_LAUNCH(CBv)                                                                                                               \
    GF_LAUNCH(ctx, "smpf_combine_fwd", (smp_combine_fwd_panels<CBv>), dim3((unsigned)((npanels + 3) / 4)), dim3(256), 0, O, d.f, d.fwd_pan,  \
              d.fwd_pan_node, npanels, (int)h.rows, d.fwd_goff, d.Gc, (long long)s->lay.level[l - 1].pairs, d.adj, d.rsum, d.Vout,          \
              (long long)h.pairs, d.Sout, bias, psum, pmax, nodefac)
    if (s->cfg.nChanels == 64) GF_CFP_LAUNCH(64);
    else if (s->cfg.nChanels == 16) GF_CFP_LAUNCH(16);
    else GF_CFP_LAUNCH(32);
#undef GF_CFP_LAUNCH
    return GF_OK;
}

// dO = [L | dU] and the per-(node, x) partials of a fused level from df_l (rows, a per-node vector, or both): smp_combine_bwd_panels
gf_status smp_combine_bwd_panels_c64(gf_smp *s, int l, const float *dfrows, const float *node_df, float *dO, float *dzmax) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int npanels = d.fwd_npanels;
    if (npanels < 1) return GF_OK;
#define GF_CBP_LAUNCH(CBv)                                                                                                              \
    GF_LAUNCH(ctx, "smpf_combine_bwd", (smp_combine_bwd_panels<CBv>), dim3((unsigned)((npanels + 3) / 4)), dim3(256), 0, d.f, dfrows, node_df, dO, \
              d.fwd_pan, d.fwd_pan_node, npanels, (int)h.rows, d.adj, d.rsum, (long long)h.pairs, d.dVout, d.dSpart, d.dbpart, dzmax)
    if (s->cfg.nChanels == 64) GF_CBP_LAUNCH(64);
    else if (s->cfg.nChanels == 16) GF_CBP_LAUNCH(16);
    else GF_CBP_LAUNCH(32);
#undef GF_CBP_LAUNCH
    return GF_OK;
}

}  // namespace gf
