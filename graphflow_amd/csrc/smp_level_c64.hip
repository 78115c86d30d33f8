// smp_level_c64.hip -- the three block-product directions of the fused SMP level (smp_fused.hip) at C = 64 channels:
//   smp_rowpanel_c64<true>   O  from T   (forward)            weights resident in LDS, 32-row panels in registers
//   smp_rowpanel_c64<false>  dT from dO  (backward, tables)   same kernel, transposed weight image
//   smp_wgrad_c64            dW from T and dO (backward, weights)   output-stationary split over the rows
// Other channel counts run the same products as grouped launches of the general tiled GEMM (mixers.hip).
#include <algorithm>
#include <cstdlib>

#include "gemm_lds.h"
#include "gf_internal.h"
#include "smp_internal.h"

namespace gf {
namespace {

using namespace lds_image;

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of the fused SMP level at C = 64, output-stationary.  The eight row block products of a level,
//   dWst[p] = A_p^T B_p summed over the level's rows,   A_p in T = [S_ab|S_bc|T6|T10],  B_p in {dO_loc, tot dO_loc, tr dO_loc, dZ, dZ'}
// (stack positions 0..7 of smp_fused.hip), share their operands: S_ab feeds four of them, dO_loc five.  As five groups of
// the grouped split-K launch each product streams its own copy (7.5 GB fetched for 5.1 GB of T and dO at cfg3, PMC).
// Here a workgroup of eight waves owns a row range, stages each 32-row slice of T (4C) and dO (3C) in LDS ONCE (two stages,
// one barrier per slice), and wave p accumulates product p as a 64 x 64 register tile (2 x 2 MFMA 32x32 accumulators); the
// tot / tr factors are applied to the dO_loc fragments in registers.  Every operand byte is read from HBM once per step,
// 64 MFMAs per wave between barriers.
// Partial images go through the same two-pass ordered reduction as every split-K launch (fixed order, deterministic).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kWgThreads = 1024, kWgARows = 256, kWgBRows = 192;
constexpr int kWgStage = (kWgARows + kWgBRows) * LDS_ROW + 2 * BK;  // floats: A image, B image, the slice's tot / tr factors
__constant__ int c_wg_ablk[8] = {0, 1, 0, 2, 3, 0, 1, 0};    // S_ab, S_bc, S_ab, T6, T10, S_ab, S_bc, S_ab
__constant__ int c_wg_bblk[8] = {0, 0, 0, 0, 0, 1, 1, 2};    // dO_loc (x tot, x tot, x tr, plain, plain), dZ, dZ, dZ'
__constant__ int c_wg_scale[8] = {0, 0, 1, -1, -1, -1, -1, -1};

// sixteen waves: wave w owns columns [32 (w & 1), +32) of product w >> 1 as two 32 x 32 accumulators (the wave tile of the
// general kernel); four waves per SIMD hide each other's LDS and barrier waits
// trow == nullptr: dO = [L | dZ | dZ'] (192 columns).  trow != nullptr (compact layout): dO = [L | dU] (128 columns) and the third
// block of the B image is dU at the TRANSPOSED row of the node, trow[row] = the row of (e, x) for row (x, e):
//   sum_rows S_ab[T(row)]^T dU[row] = sum_rows S_ab[row]^T dU[T(row)]     (the weight gradient of K11)
__global__ __launch_bounds__(kWgThreads, 1) void smp_wgrad_c64(const float *__restrict__ T, const float *__restrict__ dO,
                                                               const float *__restrict__ rs, int rows, int kchunk,
                                                               float *__restrict__ part, const int *__restrict__ trow) {
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];  // two stages: slice i is multiplied while slice i + 1 lands
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prod = wave >> 1, nh = wave & 1;
    const int kbeg = blockIdx.x * kchunk;
    const int kend = (kbeg + kchunk < rows) ? kbeg + kchunk : rows;
    constexpr int LDT = 256;
    const int LDO = trow ? 128 : 192;
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};

    // staging: A = two float4 per thread (transposing store, ascat_mk on each 128-row half), B = 1536 float4 over 1024 threads
    f4v va[2], vb[2];
    float2 sc;
    int am[2], ak[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + e * kWgThreads;  // e = 128-row half
        int m, k;
        ascat_mk(idx & 1023, &m, &k);
        am[e] = (idx >> 10) * 128 + m;
        ak[e] = k;
    }
    // B: float4 id = tid (blocks 0, 1) and 1024 + tid for tid < 512 (block 2); inside a block: 16 float4 per k row
    const int bsub = tid & 511, bn = (bsub % 16) * 4, bk = bscat_k(bsub / 16), bblk0 = tid >> 9;
    const bool b2 = tid < 512;
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int gk = k0 + ak[e];
            va[e] = gk < kend ? *reinterpret_cast<const f4v *>(T + (size_t)gk * LDT + am[e]) : zero4;
        }
        const int gk = k0 + bk;
        const bool in = gk < kend;
        vb[0] = in ? *reinterpret_cast<const f4v *>(dO + (size_t)gk * LDO + bblk0 * 64 + bn) : zero4;
        if (b2) vb[1] = in ? *reinterpret_cast<const f4v *>(trow ? dO + (size_t)trow[gk] * LDO + 64 + bn : dO + (size_t)gk * LDO + 128 + bn) : zero4;
        if (tid < 512 && bn == 0) sc = in ? *reinterpret_cast<const float2 *>(rs + (size_t)gk * 2) : make_float2(0.f, 0.f);
    };
    // one of four pieces of the slice's LDS image (issued between the MFMAs of the running slice, or all at once)
    auto store_piece = [&](float *stage, int pc) {
        float *As = stage, *Bs = stage + kWgARows * LDS_ROW, *Ss = Bs + kWgBRows * LDS_ROW;
        if (pc < 2) {
            const int kp = kpos(ak[pc]);
#pragma unroll
            for (int j = 0; j < 4; ++j) As[lds_at(am[pc] + j, kp)] = va[pc][j];
        } else if (pc == 2) {
            const int kp = kpos(bk);
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[lds_at(bblk0 * 64 + bn + j, kp)] = vb[0][j];
        } else {
            const int kp = kpos(bk);
            if (b2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[lds_at(128 + bn + j, kp)] = vb[1][j];
                if (bn == 0) {  // the slice's row factors, in the lanes' k order (kpos): lane half lh reads 16 consecutive ones
                    Ss[kp] = sc.x;
                    Ss[BK + kp] = sc.y;
                }
            }
        }
    };

    f16v acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;

    if (kbeg < kend) {
        const int a0 = c_wg_ablk[prod] * 64, b0 = c_wg_bblk[prod] * 64 + 32 * nh, skind = c_wg_scale[prod];
        const int aoff0 = (a0 + li) * LDS_ROW + lh * (BK / 2), aoff1 = aoff0 + 32 * LDS_ROW;
        const int boff = (kWgARows + b0 + li) * LDS_ROW + lh * (BK / 2);
        const int za0 = lds_swz(a0 + li), za1 = lds_swz(a0 + 32 + li), zb = lds_swz(b0 + li);
        const int soff = (kWgARows + kWgBRows) * LDS_ROW + (skind > 0 ? BK : 0) + lh * (BK / 2);
        load_tiles(kbeg);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) store_piece(wg_smem, pc);
        __syncthreads();
        int cur = 0;
        for (int k0 = kbeg;;) {
            const int kn = k0 + BK;
            const bool more = kn < kend;
            if (more) load_tiles(kn);
            const float *stage = wg_smem + cur * kWgStage;
            float *next = wg_smem + (cur ^ 1) * kWgStage;
            // LDS stores of the next slice are issued BETWEEN the MFMAs of the second half (the matrix pipe runs each MFMA for
            // 64 cycles while the wave goes on); they have had the first half to arrive from HBM
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f4v fa0[2], fa1[2], fb[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fa0[q] = *reinterpret_cast<const f4v *>(stage + aoff0 + 4 * ((2 * h + q) ^ za0));
                    fa1[q] = *reinterpret_cast<const f4v *>(stage + aoff1 + 4 * ((2 * h + q) ^ za1));
                    fb[q] = *reinterpret_cast<const f4v *>(stage + boff + 4 * ((2 * h + q) ^ zb));
                }
                if (skind >= 0) {  // (wave-uniform) B = tot dO_loc or tr dO_loc: the factor of row k on the fragment's element k
#pragma unroll
                    for (int q = 0; q < 2; ++q) fb[q] *= *reinterpret_cast<const f4v *>(stage + soff + 4 * (2 * h + q));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[j >> 2][j & 3], fb[j >> 2][j & 3], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[j >> 2][j & 3], fb[j >> 2][j & 3], acc1, 0, 0, 0);
                    if (more && h == 1 && (j & 1) == 0) store_piece(next, j >> 1);
                }
            }
            __syncthreads();
            if (!more) break;
            cur ^= 1;
            k0 = kn;
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = part + ((size_t)blockIdx.x * 8 + prod) * 4096 + 32 * nh + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[row * 64] = acc0[r];
        out[(32 + row) * 64] = acc1[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Row-panel products of the fused SMP level at C = 64 with the WEIGHTS resident in LDS and the row operand in registers.
//   forward  (FWD):  O[rows][3C]  = [O_loc | Z | Z']       from T  = [S_ab | S_bc | T6 | T10]
//                    O_loc = tot S_ab W0 + tr S_ab W2 + tot S_bc W1 + T6 W3 + T10 W4     Z = S_ab W5 + S_bc W6     Z' = S_ab W7
//   backward (!FWD): dT[rows][4C] = [dS_ab|dS_bc|dT6|dT10] from dO = [L | Z | Z']
//                    dS_ab = tot L W0^T + tr L W2^T + Z W5^T + Z' W7^T     dS_bc = tot L W1^T + Z W6^T     dT6 = L W3^T    dT10 = L W4^T
// As tiled GEMMs these are reductions of 2..10 k-steps whose prologues, epilogues and barriers cost half the time.  Here the
// eight 64 x 64 weight blocks (128 KB) are copied to LDS ONCE per workgroup in B-fragment order, and after that single
// barrier every wave works alone: it takes a 32-row panel, reads its rows straight from global memory into A fragments --
// lane (row i, half h) holds columns [32 h, 32 h + 32) of its row, i.e. the MFMA's k order is permuted consistently on
// both operands -- and runs all sixteen 32 x 32 x 64 products of the panel (512 MFMAs) out of registers and LDS, one output
// block (two accumulators) at a time.  No operand staging, no barriers, sixteen independent waves per CU.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRpWRow = 36;

// TH = threads per workgroup: 1024 (four waves per SIMD, 128 registers each) or 512 (two waves per SIMD, 256 registers: no
// spills, at half the latency-hiding waves)
// COMPACT (TH = 512 only): the projected matrix is O = [O_loc | U] with U = Z + Z'^T formed by the product itself -- the S_ab
// block of the K11 product is read at the TRANSPOSED row of the node (trow[row] = row of (e, x) for row (x, e)), so the
// consumer needs no transposed read and O / dO are 2C wide instead of 3C (0.73 GB less written and read per direction at cfg3).
//   forward  U    = S_ab W5 + S_bc W6 + S_ab[trow] W7
//   backward dS_ab = tot L W0^T + tr L W2^T + dU W5^T + dU[trow] W7^T        dS_bc = tot L W1^T + dU W6^T
template <bool FWD, int TH, bool COMPACT>
__global__ __launch_bounds__(TH, 1) void smp_rowpanel_c64(const float *__restrict__ A, const float *__restrict__ rs,
                                                                  const float *__restrict__ Wst, float *__restrict__ Out, int rows,
                                                                  const int *__restrict__ trow) {
    constexpr int OC = COMPACT ? 128 : 192;
    constexpr int LDA = FWD ? 256 : OC, LDOUT = FWD ? OC : 256;
    extern __shared__ __attribute__((aligned(16))) float rp_smem[];  // [8 pos][2 column halves][2 k halves][32 lanes][36]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6;
    // B fragment order: row ((pos 2 + nh) 2 + kh) 32 + i holds B[k = 32 kh + j][n = 32 nh + i] for j = 0..31, where
    // B = W_pos (forward: W[k][n]) or W_pos^T (backward: W[n][k])
    for (int e = tid; e < 8 * 4096; e += TH) {
        const int pos = e >> 12, r = (e >> 6) & 63, c = e & 63;  // W_pos[r][c]
        const int k = FWD ? r : c, n = FWD ? c : r;
        rp_smem[((((pos * 2 + (n >> 5)) * 2 + (k >> 5)) * 32) + (n & 31)) * kRpWRow + (k & 31)] = Wst[e];
    }
    __syncthreads();
    const int npanels = (rows + 31) / 32;
    const int nwaves = gridDim.x * (TH / 64);
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const float *bbase = rp_smem + (lh * 32 + li) * kRpWRow;

    struct Blk {
        f4v a[8];
    };
    // columns [64 blk + 32 lh, +32) of row `li` of panel p -> A fragments (zeros past the end)
    auto load_blk = [&](Blk &B, int p, int blk) {
        const int row = p * 32 + li;
        const bool ok = p < npanels && row < rows;
        const float *src = A + (size_t)(ok ? row : 0) * LDA + blk * 64 + 32 * lh;
#pragma unroll
        for (int q = 0; q < 8; ++q) B.a[q] = ok ? *reinterpret_cast<const f4v *>(src + 4 * q) : zero4;
    };
    // the same block read at the transposed rows of the panel's rows (COMPACT)
    auto load_blk_t = [&](Blk &B, int p, int blk) {
        const int row = p * 32 + li;
        const bool ok = p < npanels && row < rows;
        const float *src = A + (size_t)(ok ? trow[row] : 0) * LDA + blk * 64 + 32 * lh;
#pragma unroll
        for (int q = 0; q < 8; ++q) B.a[q] = ok ? *reinterpret_cast<const f4v *>(src + 4 * q) : zero4;
    };
    auto load_scale = [&](int p) {
        const int row = p * 32 + li;
        return (p < npanels && row < rows) ? *reinterpret_cast<const float2 *>(rs + (size_t)row * 2) : make_float2(0.f, 0.f);
    };
    // acc (two column halves) += (sc * B.a) x block wpos: 64 MFMAs, B fragments from the LDS image
    auto prod = [&](const Blk &B, bool scaled, float sc, int wpos, f16v &acc0, f16v &acc1) {
        const float *b0 = bbase + (size_t)(wpos * 4) * 32 * kRpWRow, *b1 = b0 + 2 * 32 * kRpWRow;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v bq0 = *reinterpret_cast<const f4v *>(b0 + 4 * q), bq1 = *reinterpret_cast<const f4v *>(b1 + 4 * q);
            f4v av = B.a[q];
            if (scaled) av *= sc;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bq0[c], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], bq1[c], acc1, 0, 0, 0);
            }
        }
    };
    auto clear = [&](f16v &acc0, f16v &acc1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    };
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    auto store_out = [&](int p, int o, const f16v &acc0, const f16v &acc1) {
        const int r0 = p * 32;
        float *out = Out + (size_t)(r0 + 4 * lh) * LDOUT + o * 64 + li;
        if (r0 + 32 <= rows) {  // (uniform) unconditional stores: the compiler can count what it has in flight
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

    // One panel.  X holds the panel's block 0 on entry (fetched during the previous panel); Y is free.  The other blocks are
    // requested one product (64 MFMAs) before their use, and BEFORE the output stores that precede that use in program
    // order: vmcnt counts loads and stores together and in order, so a load issued after a store cannot be waited for
    // without waiting for the store as well.  (Alternating the two buffers so that Z' is also requested early costs
    // registers -- 160 B of scratch -- and measured the same.)
    float2 sc = load_scale(blockIdx.x * (TH / 64) + wave), scn = make_float2(0.f, 0.f);
    auto panel = [&](int p, Blk &X, Blk &Y) {
        const int pn = p + nwaves;
        f16v acc0, acc1;
        if (FWD) {  // blocks: 0 S_ab, 1 S_bc, 2 T6, 3 T10; outputs: 0 O_loc, 1 Z, 2 Z'.  X = S_ab stays the resident block.
            load_blk(Y, p, 1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 7, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 5, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            load_blk(X, p, 2);                       // X <- T6 while S_bc runs
            prod(Y, true, sc.x, 1, acc0, acc1);
            load_blk(Y, p, 3);                       // Y <- T10 while T6 runs
            scn = load_scale(pn);
            prod(X, false, 1.f, 3, acc0, acc1);
            load_blk(X, pn, 0);                      // X <- S_ab of the next panel while T10 runs
            prod(Y, false, 1.f, 4, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        } else {    // blocks: 0 L, 1 Z, 2 Z'; outputs: 0 dS_ab, 1 dS_bc, 2 dT6, 3 dT10.  X = L.
            load_blk(Y, p, 1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 3, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 4, acc0, acc1);
            store_out(p, 3, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 1, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            prod(Y, false, 1.f, 5, acc0, acc1);
            load_blk(Y, p, 2);                       // Y <- Z' (the one request of a panel that is waited for at once)
            scn = load_scale(pn);
            load_blk(X, pn, 0);                      // X <- L of the next panel while Z' runs
            prod(Y, false, 1.f, 7, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        }
        sc = scn;
    };
    // Two waves per SIMD (TH = 512) have 256 registers each: every block of the panel is requested when the panel starts, behind
    // the first product (64 MFMAs, about 2 us: more than an HBM round trip), and block 0 of the next panel behind the last
    // one -- no request is ever waited for at once, nothing spills.
    auto panel_wide = [&](int p, Blk &X, Blk &Y, Blk &W, Blk &V) {
        const int pn = p + nwaves;
        f16v acc0, acc1;
        if (COMPACT && FWD) {  // T blocks: 0 S_ab, 1 S_bc, 2 T6, 3 T10; outputs: 0 O_loc, 1 U
            load_blk(Y, p, 1);
            load_blk_t(W, p, 0);                     // W <- S_ab at the transposed rows
            load_blk(V, p, 2);
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, false, 1.f, 5, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            prod(W, false, 1.f, 7, acc0, acc1);
            load_blk(W, p, 3);                       // W <- T10, four products ahead of its use
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            load_blk(X, pn, 0);                      // X <- S_ab of the next panel
            prod(Y, true, sc.x, 1, acc0, acc1);
            prod(V, false, 1.f, 3, acc0, acc1);
            prod(W, false, 1.f, 4, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        } else if (COMPACT) {  // dO blocks: 0 L, 1 dU; outputs: 0 dS_ab, 1 dS_bc, 2 dT6, 3 dT10
            load_blk(Y, p, 1);
            load_blk_t(W, p, 1);                     // W <- dU at the transposed rows
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, false, 1.f, 3, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 4, acc0, acc1);
            store_out(p, 3, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 1, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            load_blk(X, pn, 0);                      // X <- L of the next panel
            prod(Y, false, 1.f, 5, acc0, acc1);
            prod(W, false, 1.f, 7, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        } else if (FWD) {
            load_blk(Y, p, 1);
            load_blk(W, p, 2);
            load_blk(V, p, 3);
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, false, 1.f, 7, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 5, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            load_blk(X, pn, 0);                      // X <- S_ab of the next panel, three products ahead of its use
            prod(Y, true, sc.x, 1, acc0, acc1);
            prod(W, false, 1.f, 3, acc0, acc1);
            prod(V, false, 1.f, 4, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        } else {
            load_blk(Y, p, 1);
            load_blk(W, p, 2);
            scn = load_scale(pn);
            clear(acc0, acc1);
            prod(X, false, 1.f, 3, acc0, acc1);
            store_out(p, 2, acc0, acc1);
            clear(acc0, acc1);
            prod(X, false, 1.f, 4, acc0, acc1);
            store_out(p, 3, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 1, acc0, acc1);
            prod(Y, false, 1.f, 6, acc0, acc1);
            store_out(p, 1, acc0, acc1);
            clear(acc0, acc1);
            prod(X, true, sc.x, 0, acc0, acc1);
            prod(X, true, sc.y, 2, acc0, acc1);
            load_blk(X, pn, 0);                      // X <- L of the next panel, two products ahead of its use
            prod(Y, false, 1.f, 5, acc0, acc1);
            prod(W, false, 1.f, 7, acc0, acc1);
            store_out(p, 0, acc0, acc1);
        }
        sc = scn;
    };
    Blk B0, B1;
    int p = blockIdx.x * (TH / 64) + wave;
    load_blk(B0, p, 0);
    if constexpr (TH == 512) {
        Blk B2, B3;
        for (; p < npanels; p += nwaves) panel_wide(p, B0, B1, B2, B3);
    } else {
        for (; p < npanels; p += nwaves) panel(p, B0, B1);  // (block 0 of the next panel is back in B0 when a panel ends)
    }
}

}  // namespace

// The eight row block products A_p^T B_p of a fused SMP level at C = 64 (see smp_wgrad_c64) as PARTIAL images: one image of
// 8 x 64 x 64 floats per row range.  T = [rows][256], dO = [rows][192], rowscale = [rows][2].  The row range per workgroup
// depends on `rows` only: results are reproducible.  The caller folds the images in order.
gf_status smp_wgrad_partials_c64(gf_ctx *ctx, const float *T, const float *dO, const float *rowscale, int rows, float *part,
                                 size_t part_floats, FoldGroup *out, const int *trow, const WgradScales &ws, const int *trowf) {
    const size_t total = 8 * 4096;
    out->part = part;
    out->n = total;
    out->splits = 0;
    if (rows < 1) return GF_OK;
    // one workgroup fits a CU (two LDS stages): aim at `target` row ranges, at least 8 slices each
    const int target = 256;
    int kchunk = ((rows + target - 1) / target + BK - 1) / BK * BK;
    if (kchunk < 8 * BK) kchunk = 8 * BK;
    const int splits = (rows + kchunk - 1) / kchunk;
    if ((size_t)splits * total > part_floats)
        return fail(ctx, GF_ERR_NOMEM, "smp_wgrad_partials_c64: %d partial images, room for %zu", splits, part_floats / total);
    out->splits = splits;
    if (trow && ws.any() && smp_split_products(ctx))
        return smp_wgrad_partials_split_c64(ctx, T, dO, rowscale, rows, kchunk, splits, part, trow, ws, trowf);
    const size_t lds = sizeof(float) * 2 * (size_t)kWgStage;
    gf_status st = opt_in_lds(ctx, smp_wgrad_c64, lds);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "smpf_wgrad", smp_wgrad_c64, dim3((unsigned)splits), dim3(kWgThreads), lds, T, dO, rowscale, rows, kchunk, part, trow);
    return GF_OK;
}

// Row-panel products of a fused SMP level at C = 64 (see smp_rowpanel_c64): forward O from T, or backward dT from dO.
// Every output element is produced by one wave in a fixed order: results do not depend on the grid size.
gf_status smp_rowpanel_products_c64(gf_ctx *ctx, bool forward, const float *A, const float *rowscale, const float *Wst, float *Out,
                                    int rows, const int *trow, const int *trowf, bool skip_zero_grads, const void *wimg, int C, int nf, int nx) {
    if (rows < 1) return GF_OK;
    static int cu_count[64] = {};
    const int di = ctx->device & 63;
    if (!cu_count[di]) {
        GF_HIP_TRY(ctx, hipDeviceGetAttribute(&cu_count[di], hipDeviceAttributeMultiprocessorCount, ctx->device));
        if (cu_count[di] < 1) cu_count[di] = 256;
    }
    const int cus = cu_count[di];
    if (trow && smp_split_products(ctx)) return smp_rowpanel_split_c64(ctx, forward, A, rowscale, Wst, Out, rows, trow, cus, trowf, skip_zero_grads, wimg, C, nf, nx);
    if (C != 64 || nf != 2 || nx != 0) return fail(ctx, GF_ERR_UNSUPPORTED, "smp_rowpanel_products: %d channels / %d row factors / %d extra products on the fp32 matrix pipe", C, nf, nx);
    const size_t lds = sizeof(float) * 8 * 2 * 2 * 32 * (size_t)kRpWRow;
    // forward: four waves per SIMD (measured equal to two); backward: two waves per SIMD with 256 registers -- the 128-register
    // build of the backward panel spills 100 B per lane and waits for one block request per panel (1.84 -> 1.66 ms at cfg3).
    // The compact layout (trow given) keeps four operand blocks in registers: two waves per SIMD in both directions.
    const int th = (forward && !trow) ? 1024 : 512;
    const int npanels = (rows + 31) / 32, per = th / 64;
    const int want = (npanels + per - 1) / per;
    const int grid = want < cus ? want : cus;  // one persistent workgroup per CU (the weight image takes 144 KB of LDS)
    const char *name = forward ? "smpf_products_fwd" : "smpf_products_bwd";
#define GF_RP_LAUNCH(F, T, Cm)                                                                                              \
    do {                                                                                                                    \
        gf_status st__ = opt_in_lds(ctx, smp_rowpanel_c64<F, T, Cm>, lds);                                                  \
        if (st__ != GF_OK) return st__;                                                                                     \
        GF_LAUNCH(ctx, name, (smp_rowpanel_c64<F, T, Cm>), dim3((unsigned)grid), dim3(T), lds, A, rowscale, Wst, Out, rows, trow); \
    } while (0)
    if (trow) {
        if (forward) GF_RP_LAUNCH(true, 512, true);
        else GF_RP_LAUNCH(false, 512, true);
    } else if (forward) {
        GF_RP_LAUNCH(true, 1024, false);
    } else {
        GF_RP_LAUNCH(false, 512, false);
    }
#undef GF_RP_LAUNCH
    return GF_OK;
}

}  // namespace gf
