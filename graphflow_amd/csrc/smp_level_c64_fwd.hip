// smp_level_c64_fwd.hip -- the forward block products of the fused SMP level at C = 64 WITH combine-forward folded in: the
// projected matrix O = [O_loc | U] is never written to (or read back from) HBM.
//
//   f_l[x, y] = LeakyReLU( b + O_loc[x, y] + sum_e A+[y, e] U'[x, e] + r[y] V[x] + A+[x, y] S )          (smp_fused.hip, combine)
//   O_loc = tot (S_ab W0 + S_bc W1) + tr S_ab W2 + T6 W3 + T10 W4,   U' = S_ab W5 + S_bc W6 + S_ab[trow] W7 + G15[x, e] + G16[e, x]
//
// Same decomposition as smp_rowpanel_split<true> (smp_level_c64_split.hip: the eight weight blocks as two f16 fragment images in
// LDS for the life of the workgroup, every wave alone on a panel of up to 32 rows, operands split into two f16 halves in
// registers, three v_mfma_f32_32x32x16_f16 per product term), with two differences:
//   * a panel is a run of whole (node, x) row groups (s rows each, consecutive x of ONE node, at most 32 rows: gfsmp-independent
//     table `pan`, built at prepare time), so that everything combine needs across rows is inside the wave's accumulators;
//   * the epilogue runs on the accumulators where they are.  In the C/D layout of the 32 x 32 MFMA a lane holds column (lane & 31)
//     of sixteen rows; register r of the two lane halves holds exactly the two k-rows that step r of v_mfma_f32_32x32x2_f32 wants
//     as its B operand, so   M += A' U'   (A' = the panel's block-diagonal gated adjacency, 32 x 32)   is sixteen fp32 MFMAs per
//     column half straight from the U accumulators -- no transpose, no LDS.  The rank-one terms r[y] V[x], A+[x, y] S and the bias
//     are nine more steps of the same instruction (k = group index / S / bias).  Exact fp32 products, fp32 accumulation.
// What it saves per level: O written (2C per row) and read back, the combine-forward launch (its adjacency images, barriers, LDS).
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
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr int kFfThreads = 256;            // ONE wave per SIMD with all 512 registers: the whole next panel is in flight (see the loop)
constexpr int kFfImg = 8 * 2 * 4 * 64;     // 16-byte fragment entries per image: [pos][column half][k chunk][lane]
constexpr size_t kFfLds = 2 * (size_t)kFfImg * 16 + 16 * sizeof(float) + (kFfThreads / 64) * 32 * sizeof(float);
constexpr float kAlphaFf = 0.01f;
constexpr int kFfOor = 0x40000000;  // a lane offset past every buffer this kernel addresses through a descriptor: the load returns 0

__device__ __forceinline__ void ff_pow2_scale(unsigned maxbits, float *s, float *inv) {
    unsigned e = maxbits >> 23;
    e = e < 14u ? 14u : e;
    *s = __uint_as_float((267u - e) << 23);
    *inv = __uint_as_float((e - 13u) << 23);
}
__device__ __forceinline__ void ff_split_pair(float a, float b, float s, h2 *h, h2 *l) {
    const f2v x = {a * s, b * s};
    *h = __builtin_convertvector(x, h2);
    const f2v r = x - __builtin_convertvector(*h, f2v);
    *l = __builtin_convertvector(r, h2);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ff_rsrc(const void *base, size_t bytes) {
    const unsigned n = bytes > 0xfffffffcull ? 0xfffffffcu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float ff_ld1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// pan[p] = {first row of the panel, rows | s << 8 | groups << 16 | x0 << 24, first row of the node, first pair of the node}
// (x0 = index x of the panel's first row group inside its node); pan_node[p] = the node
__global__ __launch_bounds__(kFfThreads, 1) void smp_level_fwd_fused(
    const float *__restrict__ T, const float *__restrict__ rs, const float *__restrict__ Wst, float *__restrict__ F,
    const int *__restrict__ trow, const int4 *__restrict__ pan, const int *__restrict__ pan_node, int npanels, int rows,
    const int2 *__restrict__ goff, const float *__restrict__ Gc, long long gc_rows, const float *__restrict__ adj,
    const float *__restrict__ rsum, const float *__restrict__ Vout, long long pairs, const float *__restrict__ Sout,
    const float *__restrict__ bias) {
    constexpr int LDA = 256;
    extern __shared__ __attribute__((aligned(16))) uint4 ff_smem[];
    uint4 *imgH = ff_smem, *imgL = ff_smem + kFfImg;
    float *winv = reinterpret_cast<float *>(ff_smem + 2 * kFfImg);  // [8] 2^-k of the weight blocks
    unsigned *wmax = reinterpret_cast<unsigned *>(winv + 8);        // [8]
    float *facs = winv + 16;                                        // [waves][32]: row factors on their way to the C layout
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6;

    // ---- weight images (as smp_rowpanel_split<true>)
    if (tid < 8) wmax[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int pos = 0; pos < 8; ++pos) {
        unsigned m = 0u;
#pragma unroll
        for (int i = 0; i < 4096 / kFfThreads; ++i) {
            const unsigned b = __float_as_uint(Wst[pos * 4096 + i * kFfThreads + tid]) & 0x7fffffffu;
            m = b > m ? b : m;
        }
        atomicMax(&wmax[pos], m);
    }
    __syncthreads();
    for (int t = tid; t < kFfImg; t += kFfThreads) {
        const int ln = t & 63, c = (t >> 6) & 3, nh = (t >> 8) & 1, pos = t >> 9;
        const int n = 32 * nh + (ln & 31), k0 = 32 * (ln >> 5) + 8 * c;
        float s, inv;
        ff_pow2_scale(wmax[pos], &s, &inv);
        if (ln == 0 && c == 0 && nh == 0) winv[pos] = inv;
        const float *w = Wst + pos * 4096;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[(k0 + j) * 64 + n];
        unsigned hw[4], lw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h2 h, l;
            ff_split_pair(v[2 * j], v[2 * j + 1], s, &h, &l);
            hw[j] = __builtin_bit_cast(unsigned, h);
            lw[j] = __builtin_bit_cast(unsigned, l);
        }
        imgH[t] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        imgL[t] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
    __syncthreads();

    const int nwaves = gridDim.x * (kFfThreads / 64);
    float *myfac = facs + wave * 32;
    const __amdgpu_buffer_rsrc_t rGc = ff_rsrc(Gc, (size_t)gc_rows * 128 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rAdj = ff_rsrc(adj, (size_t)rows * sizeof(float));
    const __amdgpu_buffer_rsrc_t rV = ff_rsrc(Vout, (size_t)pairs * 64 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rF = ff_rsrc(F, (size_t)rows * 64 * sizeof(float));

    struct Raw {
        f4v a[8];
    };
    struct Spl {
        uint4 h[4], l[4];  // eight f16 each
    };
    // the lane's row of a panel (rows past the panel's end repeat its last row: unconditional loads, results never stored)
    auto row_of = [&](const int4 &P) {
        const int n = P.y & 0xff;
        return P.x + (li < n ? li : n - 1);
    };
    auto load_raw_at = [&](Raw &R, int src_row, int blk) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(src_row));
        const float *src = T + (size_t)src_row * LDA + blk * 64 + 32 * lh;
#pragma unroll
        for (int q = 0; q < 8; ++q) R.a[q] = *reinterpret_cast<const f4v *>(src + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto split_blk = [&](const Raw &R, Spl &S, float &inv) {
        __builtin_amdgcn_sched_barrier(0);
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
        ff_pow2_scale(m, &s, &inv);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned hw[4], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h2 h, l;
                const f4v v = R.a[2 * c + (j >> 1)];
                ff_split_pair(v[2 * (j & 1)], v[2 * (j & 1) + 1], s, &h, &l);
                hw[j] = __builtin_bit_cast(unsigned, h);
                lw[j] = __builtin_bit_cast(unsigned, l);
            }
            S.h[c] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            S.l[c] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // acc (two column halves) += rowfac * (S x block wpos)
    auto prod = [&](const Spl &S, float rowfac, int wpos, f16v &acc0, f16v &acc1) {
        __builtin_amdgcn_wave_barrier();
        myfac[li] = rowfac * winv[wpos];
        __builtin_amdgcn_wave_barrier();
        const uint4 *bh = imgH + (size_t)(wpos * 8) * 64 + lane, *bl = imgL + (size_t)(wpos * 8) * 64 + lane;
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
    auto pan_at = [&](int p) { return pan[p < npanels ? p : npanels - 1]; };  // (past the end: the last panel again, never stored)

    // One panel per iteration, ONE wave per SIMD.  The wave's 512 registers hold the five operand blocks of a panel as raw fp32
    // (S_ab, S_ab at the transposed rows, S_bc, T6, T10: 160 registers); as soon as a block has been split into its f16 halves its
    // registers take the request for the SAME block of the wave's next panel, so every block is in flight for a whole panel's
    // worth of products -- the latency cover two waves per SIMD gave the unfused kernel, without its 256-register ceiling (U and M
    // live together, the epilogue's 100 gathered values in flight beside them).
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto pan_u = [&](int q) {  // a panel record in scalar registers
        const int4 v = pan_at(q);
        return make_int4(rfl(v.x), rfl(v.y), rfl(v.z), rfl(v.w));
    };
    int p = blockIdx.x * (kFfThreads / 64) + wave;
    if (p >= npanels) return;
    int4 P = pan_u(p), Pn = pan_u(p + nwaves);
    int tnext = trow[row_of(Pn)];
    int2 go = goff[row_of(P)];  // Gc rows of G15[x, e] / G16[e, x] of the lane's row, -1 = none (requested a panel ahead)
    Raw R0, R1, R2, R3, R4;
    {
        const int row = row_of(P);
        load_raw_at(R0, row, 0);
        load_raw_at(R1, trow[row], 0);
        load_raw_at(R2, row, 1);
        load_raw_at(R3, row, 2);
        load_raw_at(R4, row, 3);
    }
    for (; p < npanels; p += nwaves) {
        const int4 Pnn = pan_u(p + 2 * nwaves);
        const int row = row_of(P), rown = row_of(Pn);
        const int tnn = trow[row_of(Pnn)];
        const int2 gon = goff[rown];
        const int nrows = P.y & 0xff, s = (P.y >> 8) & 0xff, G = (P.y >> 16) & 0xff, x0 = (P.y >> 24) & 0xff;
        const int node = rfl(pan_node[p]);
        const float2 sc = *reinterpret_cast<const float2 *>(rs + (size_t)row * 2);
        const bool rowok = li < nrows;
        const int gs = (int)((li + 0.5f) * __builtin_amdgcn_rcpf((float)s));  // group of row li inside the panel (li < 32: exact)
        const int y_li = li - gs * s;                                          // its position y inside the group
        // ---- every operand of the epilogue is requested NOW (the gather indices came a panel ago).
        //      G15[x, e] + G16[e, x] of the 32 rows: register r of lane half lh is row (r & 3) + 8 (r >> 2) + 4 lh of the panel;
        //      that row's indices sit in lane (row) of go.
        float g0[16], g1[16];
        int v15[16], v16[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const int a15 = __builtin_amdgcn_readlane(go.x, rr), b15 = __builtin_amdgcn_readlane(go.x, rr + 4);
            const int a16 = __builtin_amdgcn_readlane(go.y, rr), b16 = __builtin_amdgcn_readlane(go.y, rr + 4);
            const int i15 = lh ? b15 : a15, i16 = lh ? b16 : a16;
            v15[r] = i15 < 0 ? kFfOor : i15 * 512 + li * 4;
            v16[r] = i16 < 0 ? kFfOor : i16 * 512 + 256 + li * 4;
            g0[r] = ff_ld1(rGc, v15[r]);
            g1[r] = ff_ld1(rGc, v16[r]);
        }
        //      the lane's half row of the panel's block-diagonal gated adjacency (A operand of M = A' U'), and the rank-one terms'
        //      operands (k = 0: A+[x, y] S, k = 1: bias, k = 2 + g: r[y] V[x0 + g] for the rows of group g; lane half lh supplies
        //      k = 2 t + lh at step t)
        float aval[16], vb0[5], vb1[5];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int er = (r & 3) + 8 * (r >> 2) + 4 * lh;  // the k-row this lane's A operand multiplies at step r
            const int e = er - gs * s;                        // its position inside the lane's own group, if it is in it
            const bool in = rowok && e >= 0 && e < s && er < nrows;
            aval[r] = ff_ld1(rAdj, in ? (P.z + y_li * s + e) * 4 : kFfOor);  // A[y][e] of the node (0 outside the group)
        }
        {
            const float *p0 = lh ? bias : Sout + (size_t)node * 64;
            vb0[0] = p0[li];
            vb1[0] = p0[32 + li];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int g = 2 * t + lh;
                const int vo = g < G ? (int)(((long long)P.w + x0 + g) * 256) + li * 4 : kFfOor;
                vb0[t + 1] = ff_ld1(rV, vo);
                vb1[t + 1] = ff_ld1(rV, vo + 128);
            }
        }
        const float axy = rowok ? adj[row] : 0.f;                          // A[x][y]: the adjacency entry AT the row's own index
        const float r_y = rowok ? rsum[(size_t)P.w + y_li] : 0.f;          // r[y]
        f16v u0, u1, m0, m1;
        Spl X, Y, Z;
        float iX, iY, iZ;
        // ---- U = S_ab W5 + S_ab[trow] W7 + S_bc W6
        split_blk(R0, X, iX);
        load_raw_at(R0, rown, 0);                // (the next panel's S_ab)
        split_blk(R1, Z, iZ);
        load_raw_at(R1, tnext, 0);               // (its S_ab at the transposed rows)
        clear(u0, u1);
        prod(X, iX, 5, u0, u1);
        prod(Z, iZ, 7, u0, u1);
        split_blk(R2, Y, iY);
        load_raw_at(R2, rown, 1);                // (its S_bc)
        prod(Y, iY, 6, u0, u1);
        // ---- U' = U + G15 + G16 (the second column half's gathers are requested when the first half's have been added)
#pragma unroll
        for (int r = 0; r < 16; ++r) u0[r] += g0[r] + g1[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g0[r] = ff_ld1(rGc, v15[r] + 128);
            g1[r] = ff_ld1(rGc, v16[r] + 128);
        }
        // ---- M = A' U'  (fp32 MFMA, the U accumulators as the B operand where they are)
        clear(m0, m1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            aval[r] = aval[r] > 0.f ? aval[r] : 0.f;  // the gate of RisiContraction_18 (RisiContraction_18.h:90)
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aval[r], u0[r], m0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) u1[r] += g0[r] + g1[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aval[r], u1[r], m1, 0, 0, 0);
        // ---- M += the rank-one terms
        {
            const float a0 = lh ? 1.f : (axy > 0.f ? axy : 0.f);
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb0[0], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb1[0], m1, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {  // (a panel holds at most eight row groups: build_fwd_panels)
                if (2 * t < G) {  // (uniform; nothing is requested in here)
                    const float a = (2 * t + lh) == gs ? r_y : 0.f;
                    m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb0[t + 1], m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb1[t + 1], m1, 0, 0, 0);
                }
            }
        }
        // ---- M += O_loc = tot (S_ab W0 + S_bc W1) + tr S_ab W2 + T6 W3 + T10 W4
        prod(X, iX * sc.x, 0, m0, m1);
        prod(X, iX * sc.y, 2, m0, m1);
        prod(Y, iY * sc.x, 1, m0, m1);
        split_blk(R3, Z, iZ);
        load_raw_at(R3, rown, 2);                // (its T6)
        prod(Z, iZ, 3, m0, m1);
        split_blk(R4, Z, iZ);
        load_raw_at(R4, rown, 3);                // (its T10)
        prod(Z, iZ, 4, m0, m1);
        // ---- LeakyReLU and the store of f_l (rows past the panel's end go out of range: dropped)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int vo = rr < nrows ? (P.x + rr) * 256 + li * 4 : kFfOor;
            const float z0 = m0[r], z1 = m1[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z0 > 0.f ? z0 : kAlphaFf * z0), rF, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z1 > 0.f ? z1 : kAlphaFf * z1), rF, vo + 128, 0, 0);
        }
        P = Pn;
        Pn = Pnn;
        tnext = tnn;
        go = gon;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// combine-forward on the same row panels, as a kernel of its own (default at C = 64, compact O): the epilogue of the kernel above
// with O_loc and U read from the projected matrix instead of computed.  One WAVE per panel, no LDS, no barrier: every operand of
// the panel -- its rows of O, the 64 gathered compact products, the lane's half row of the block-diagonal adjacency, the
// rank-one operands -- is requested up front (one round trip), the adjacency product and the rank-one terms are fp32 MFMAs on
// registers laid out as the MFMA's C/D tile, f_l leaves as 128-byte row segments.  Replaces smp_combine_fwd<16> (workgroup per
// (node, four x): adjacency image in LDS, a barrier between its two phases, a quarter of its waves idle on ragged quads).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void smp_combine_fwd_panels(
    const float *__restrict__ O, float *__restrict__ F, const int4 *__restrict__ pan, const int *__restrict__ pan_node, int npanels,
    int rows, const int2 *__restrict__ goff, const float *__restrict__ Gc, long long gc_rows, const float *__restrict__ adj,
    const float *__restrict__ rsum, const float *__restrict__ Vout, long long pairs, const float *__restrict__ Sout,
    const float *__restrict__ bias, float *__restrict__ psum) {  // psum (top level, or null): [npanels][64] column sums of the panel's
    // rows of f -- the readout's per-node sums (ShrinkTensor, SMP_omega.h:671-676) then read 22 MB of partials instead of f_L again
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
    const __amdgpu_buffer_rsrc_t rO = ff_rsrc(O, (size_t)rows * 128 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rGc = ff_rsrc(Gc, (size_t)gc_rows * 128 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rAdj = ff_rsrc(adj, (size_t)rows * sizeof(float));
    const __amdgpu_buffer_rsrc_t rV = ff_rsrc(Vout, (size_t)pairs * 64 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rF = ff_rsrc(F, (size_t)rows * 64 * sizeof(float));
    const int gs = (int)((li + 0.5f) * __builtin_amdgcn_rcpf((float)s));
    const int y_li = li - gs * s;
    // operands that do not depend on the gather indices first
    f16v m0, m1, u0, u1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * 512 + li * 4 : kFfOor;
        m0[r] = ff_ld1(rO, vo);
        m1[r] = ff_ld1(rO, vo + 128);
        u0[r] = ff_ld1(rO, vo + 256);
        u1[r] = ff_ld1(rO, vo + 384);
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
        const float *p0 = lh ? bias : Sout + (size_t)node * 64;
        vb0[0] = p0[li];
        vb1[0] = p0[32 + li];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int g = 2 * t + lh;
            const int vo = g < G ? (int)(((long long)P.w + x0 + g) * 256) + li * 4 : kFfOor;
            vb0[t + 1] = ff_ld1(rV, vo);
            vb1[t + 1] = ff_ld1(rV, vo + 128);
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
        const int v15 = i15 < 0 ? kFfOor : i15 * 512 + li * 4, v16 = i16 < 0 ? kFfOor : i16 * 512 + 256 + li * 4;
        g0[r] = ff_ld1(rGc, v15);
        g1[r] = ff_ld1(rGc, v16);
        g2[r] = ff_ld1(rGc, v15 + 128);
        g3[r] = ff_ld1(rGc, v16 + 128);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u0[r] += g0[r] + g1[r];
        u1[r] += g2[r] + g3[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float a = aval[r] > 0.f ? aval[r] : 0.f;  // the gate of RisiContraction_18 (RisiContraction_18.h:90)
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, u0[r], m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, u1[r], m1, 0, 0, 0);
    }
    {
        const float a0 = lh ? 1.f : (axy > 0.f ? axy : 0.f);
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb0[0], m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, vb1[0], m1, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (2 * t < G) {  // (uniform; nothing is requested in here)
                const float a = (2 * t + lh) == gs ? r_y : 0.f;
                m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb0[t + 1], m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vb1[t + 1], m1, 0, 0, 0);
            }
        }
    }
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int vo = rr < nrows ? (P.x + rr) * 256 + li * 4 : kFfOor;
        const float z0 = m0[r], z1 = m1[r];
        const float f0 = z0 > 0.f ? z0 : kAlphaFf * z0, f1 = z1 > 0.f ? z1 : kAlphaFf * z1;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f0), rF, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f1), rF, vo + 128, 0, 0);
        if (rr < nrows) c0 += f0, c1 += f1;   // (rows in register order: a fixed order)
    }
    if (psum) {  // (uniform)
        c0 += __shfl_xor(c0, 32);
        c1 += __shfl_xor(c1, 32);
        psum[(size_t)p * 64 + lane] = lh ? c1 : c0;   // lane = 32 lh + li: columns li | 32 + li
    }
}

// ---- per-level tables of the kernel above, built on the device at prepare time
// node_panel[n] = first panel of node n; a node of size s has ceil(s / gpp) panels of gpp = max(1, 32 / s) row groups
__global__ void build_fwd_panels(const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                 const long long *__restrict__ node_pair, const int *__restrict__ node_panel, int4 *__restrict__ pan,
                                 int *__restrict__ pan_node) {
    const int n = blockIdx.x, s = node_s[n];
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

gf_status smp_fwd_fused_build_tables(gf_smp *s, int l, hipStream_t stream) {
    gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    if (!d.fwd_pan || h.nNodes == 0) return GF_OK;
    hipLaunchKernelGGL(build_fwd_panels, dim3((unsigned)h.nNodes), dim3(64), 0, stream, d.node_s, d.node_row, d.node_pair, d.node_panel,
                       d.fwd_pan, d.fwd_pan_node);
    hipLaunchKernelGGL(build_fwd_goff, dim3((unsigned)h.nNodes), dim3(64), 0, stream, d.node_s, d.node_row, d.node_pair, d.pair_src_pair,
                       d.pi, d.fwd_goff);
    GF_LAUNCH_CHECK(s->ctx, "build_fwd_panels");
    return GF_OK;
}

// f_l from the projected matrix O = [O_loc | U] (compact layout) of a fused level at C = 64: smp_combine_fwd_panels
gf_status smp_combine_fwd_panels_c64(gf_smp *s, int l, const float *O, const float *bias, float *psum) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int npanels = d.fwd_npanels;
    if (npanels < 1) return GF_OK;
    GF_LAUNCH(ctx, "smpf_combine_fwd", smp_combine_fwd_panels, dim3((unsigned)((npanels + 3) / 4)), dim3(256), 0, O, d.f, d.fwd_pan,
              d.fwd_pan_node, npanels, (int)h.rows, d.fwd_goff, d.Gc, (long long)s->lay.level[l - 1].pairs, d.adj, d.rsum, d.Vout,
              (long long)h.pairs, d.Sout, bias, psum);
    return GF_OK;
}

// f_l of a fused level at C = 64 from its tables T, with the projected matrix kept in registers (see the head of this file)
gf_status smp_level_fwd_fused_c64(gf_smp *s, int l, const float *T, const float *bias, int cus) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int npanels = d.fwd_npanels;
    if (npanels < 1) return GF_OK;
    const int per = kFfThreads / 64, want = (npanels + per - 1) / per;
    const int grid = want < cus ? want : cus;  // one persistent workgroup per CU (the weight images take 128 KB of LDS)
    gf_status st = opt_in_lds(ctx, smp_level_fwd_fused, kFfLds);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "smpf_level_fwd", smp_level_fwd_fused, dim3((unsigned)grid), dim3(kFfThreads), kFfLds, T, d.rowscale, d.Wst, d.f, d.trow,
              d.fwd_pan, d.fwd_pan_node, npanels, (int)h.rows, d.fwd_goff, d.Gc, (long long)s->lay.level[l - 1].pairs, d.adj, d.rsum, d.Vout,
              (long long)h.pairs, d.Sout, bias);
    return GF_OK;
}

}  // namespace gf
