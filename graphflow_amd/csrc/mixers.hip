// mixers.hip -- the dense feature mixers of the SMP path on gfx950: MatMul, MatTensorMul, TensorMatMul, StackTensor3D.
//
// Replaces GraphFlow/MatMul.h:48-82, MatTensorMul.h:47-85, TensorMatMul.h:46-84 (naive triple loops) and
// StackTensor3D.h:54-90.  All three products are one strided-batched fp32 GEMM on the matrix cores:
//   MatMul        C[M,N]      = A[M,K] B[K,N]                  bwd  dA += dC B^T,  dB += A^T dC
//   MatTensorMul  Out[R,J*D]  = X[R,Kd] F[Kd,J*D]              bwd  dX += G F^T,   dF += X^T G
//   TensorMatMul  Out_i[J,D]  = Y^T[J,Kd] F_i[Kd,D], i < R     bwd  dF_i += Y G_i, dY += sum_i F_i G_i^T
// using v_mfma_f32_32x32x2_f32 (exact fp32, the only fp32-input matrix instruction on gfx950; there is no
// xf32/TF32).  Tile: 128 x 64 x 32 per 256-thread workgroup, 2 x 2 waves of two 32 x 32 accumulators each, operands
// staged through a padded, swizzled LDS image (gemm_lds.h), next tile's global loads issued before the current tile's MFMAs.  Reductions over a
// long K with a small output (dB of the K-projection: K = sum s^2 rows) are split over K into a workspace and
// summed by a second kernel in a fixed order, so results are deterministic (no atomics).
#include <algorithm>
#include <cstdlib>

#include "gemm_lds.h"
#include "smp_internal.h"

namespace gf {
namespace {

using namespace lds_image;

struct GemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    int lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int kchunk;            // K range per split (multiple of BK); splits = gridDim.z / batch
    int nsplits;           // free-form grouped launches: this group's own split count (partial images split_stride apart)
    int batch;
    int ntiles_n;
    int accumulate;        // C += result (only when not splitting)
    // optional segmented K (grouped launches): K = sum of nseg pieces, piece s reads A + a_off[s], B + b_off[s] over
    // klen[s] (a multiple of BK); nseg == 0 means the plain contiguous K of the fields above
    long long split_stride;  // elements between consecutive split-K partial images of C
    int nseg;
    long long a_off[4], b_off[4];
    int klen[4];
    // optional scaling of op(A): element (m,k) is multiplied by rs[row * rs_ld + scol[piece]] with row = m (A stored
    // row-major, !TA) or row = k (TA, the reduction runs over the rows); scol < 0 or rs == nullptr: no scaling.
    // (The fused SMP level folds the per-node factors `total` and `trace` of the adjacency into its block products.)
    const float *rs;
    int rs_ld;
    int scol[4];
};

// Several GEMMs in ONE launch.  Tiles are ordered panel-major: all tiles of every group that belong to panel p (the
// same rows of the shared operand, or the same split-K row range) are adjacent in the grid, so they run concurrently
// and the shared operand is fetched from HBM once and hit in L2 by the others.
constexpr int kMaxGroups = 8;
struct GroupedArgs {
    GemmArgs g[kMaxGroups];
    int ngroups;
    int tiles_per_panel;          // sum over groups of tiles in one panel
    int tile_prefix[kMaxGroups];  // first tile of each group inside a panel
    int panel_is_split;           // 0: panel = M tile (tile index = N tile); 1: panel = split (tile index = M tile, N tile 0)
    // Row-panel launches order their tiles in windows of `window` panels, tile-type-major inside a window: the workgroups
    // that are co-resident on a CU (four consecutive ones of an XCD) then run the SAME kind of tile (same number of
    // k-steps) of neighbouring panels, instead of the 8/4/2/2-step tiles of one panel whose phases line up -- with four
    // tiles per panel that resonance cost 40 % of the table-gradient GEMM.  The panel's tiles stay within window * tiles
    // consecutive workgroups of one XCD, so they still share its L2.
    int npanels, window;
};

// element (m,k) of op(A): TA ? A[k*lda + m] : A[m*lda + k];   element (k,n) of op(B): TB ? B[n*ldb + k] : B[k*ldb + n]
// LDS images: As[m][kpos(k)] (128 x 36) and Bs[n][kpos(k)] (64 x 36).  2 x 2 waves; wave (wm, wn) owns rows
// [64 wm, 64 wm + 64) x cols [32 wn, 32 wn + 32) as two independent 32 x 32 accumulators, so each k-step issues two
// MFMAs that share one B fragment and hide each other's 64-cycle dependent latency.
// VEC: every operand is 16-byte aligned with leading dimensions and extents that are multiples of 4: global traffic
// moves as float4 along each operand's contiguous direction and the C tile leaves through LDS as float4 rows.
template <bool TA, bool TB, bool VEC>
__device__ __forceinline__ void gemm_tile(const GemmArgs &g, float *smem, int m0, int n0, int bz, int split, int splits) {
    float *As = smem, *Bs = smem + BM * LDS_ROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const float *A = g.A + bz * g.sA, *B = g.B + bz * g.sB;
    int kbeg = split * g.kchunk;
    int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    int seg = 0;
    if (g.nseg > 0) {  // segmented K: start on piece 0
        A += g.a_off[0];
        B += g.b_off[0];
        kbeg = 0;
        kend = g.klen[0];
    }

    constexpr int NA = VEC ? 4 : 16, NB = VEC ? 2 : 8;  // loads per thread per tile
    f4v va[VEC ? NA : 1], vb[VEC ? NB : 1];
    float ra[VEC ? 1 : NA], rb[VEC ? 1 : NB];
    float sa[VEC ? NA : 1];  // per-load row factors of op(A) (GemmArgs::rs); multiplied in when the tile goes to LDS so that
    bool scaled = false;  // the factor loads stay in flight behind the MFMAs like the operand loads do
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int k0) {
        const int sc = g.rs ? g.scol[seg] : -1;  // `seg` is the piece the tile being loaded belongs to
        scaled = sc >= 0;
        if (VEC) {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * kThreads;
                int m, k;
                if (TA) { ascat_mk(idx, &m, &k); } else { k = (idx % (BK / 4)) * 4; m = rowst_m(idx / (BK / 4)); }
                const int gm = m0 + m, gk = k0 + k;
                const size_t off = TA ? (size_t)gk * g.lda + gm : (size_t)gm * g.lda + gk;
                const bool in = gm < g.M && gk < kend;
                va[e] = in ? *reinterpret_cast<const f4v *>(A + off) : zero4;
                if (TA && sc >= 0) sa[e] = in ? g.rs[(size_t)gk * g.rs_ld + sc] : 0.f;  // applied in store_tiles
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int idx = tid + e * kThreads;
                int k, n;
                if (TB) { k = (idx % (BK / 4)) * 4; n = rowst_m(idx / (BK / 4)); } else { n = (idx % (BN / 4)) * 4; k = bscat_k(idx / (BN / 4)); }
                const int gn = n0 + n, gk = k0 + k;
                const size_t off = TB ? (size_t)gn * g.ldb + gk : (size_t)gk * g.ldb + gn;
                vb[e] = (gn < g.N && gk < kend) ? *reinterpret_cast<const f4v *>(B + off) : zero4;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * kThreads;
                int m, k;
                if (TA) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
                const int gm = m0 + m, gk = k0 + k;
                const size_t off = TA ? (size_t)gk * g.lda + gm : (size_t)gm * g.lda + gk;
                const bool in = gm < g.M && gk < kend;
                ra[e] = in ? A[off] : 0.f;
                if (sc >= 0 && in) ra[e] *= g.rs[(size_t)(TA ? gk : gm) * g.rs_ld + sc];  // (scalar path: no deferral)
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int idx = tid + e * kThreads;
                int k, n;
                if (TB) { k = idx % BK; n = idx / BK; } else { n = idx % BN; k = idx / BN; }
                const int gn = n0 + n, gk = k0 + k;
                const size_t off = TB ? (size_t)gn * g.ldb + gk : (size_t)gk * g.ldb + gn;
                rb[e] = (gn < g.N && gk < kend) ? B[off] : 0.f;
            }
        }
    };
    // !TA: a thread loads the same rows in every k-step, so its factors change only with the K piece
    auto load_row_scales = [&]() {
        if (VEC && !TA && g.rs) {
            const int sc = g.scol[seg];
#pragma unroll
            for (int e = 0; e < (VEC ? NA : 1); ++e) {
                const int gm = m0 + rowst_m((tid + e * kThreads) / (BK / 4));
                sa[e] = (sc >= 0 && gm < g.M) ? g.rs[(size_t)gm * g.rs_ld + sc] : 1.f;
            }
        }
    };
    auto store_tiles = [&]() {
        if (VEC && scaled) {
#pragma unroll
            for (int e = 0; e < (VEC ? NA : 1); ++e) va[e] *= sa[e];
        }
        if (VEC) {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * kThreads;
                if (TA) {  // four consecutive m at one k
                    int m, k;
                    ascat_mk(idx, &m, &k);
                    const int kp = kpos(k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[lds_at(m + j, kp)] = va[e][j];
                } else {   // four consecutive k of one row: even pair and odd pair are each contiguous
                    const int k = (idx % (BK / 4)) * 4, m = rowst_m(idx / (BK / 4));
                    float2 ev = make_float2(va[e][0], va[e][2]), od = make_float2(va[e][1], va[e][3]);
                    *reinterpret_cast<float2 *>(As + lds_at(m, k >> 1)) = ev;
                    *reinterpret_cast<float2 *>(As + lds_at(m, BK / 2 + (k >> 1))) = od;
                }
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int idx = tid + e * kThreads;
                if (TB) {
                    const int k = (idx % (BK / 4)) * 4, n = rowst_m(idx / (BK / 4));
                    float2 ev = make_float2(vb[e][0], vb[e][2]), od = make_float2(vb[e][1], vb[e][3]);
                    *reinterpret_cast<float2 *>(Bs + lds_at(n, k >> 1)) = ev;
                    *reinterpret_cast<float2 *>(Bs + lds_at(n, BK / 2 + (k >> 1))) = od;
                } else {
                    const int n = (idx % (BN / 4)) * 4, kp = kpos(bscat_k(idx / (BN / 4)));
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[lds_at(n + j, kp)] = vb[e][j];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NA; ++e) {
                const int idx = tid + e * kThreads;
                int m, k;
                if (TA) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
                As[lds_at(m, kpos(k))] = ra[e];
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int idx = tid + e * kThreads;
                int k, n;
                if (TB) { k = idx % BK; n = idx / BK; } else { n = idx % BN; k = idx / BN; }
                Bs[lds_at(n, kpos(k))] = rb[e];
            }
        }
    };

    f16v acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;

    if (kbeg < kend) {
        load_row_scales();
        load_tiles(kbeg);
        store_tiles();
        __syncthreads();
        // fragment q of a lane = logical slot 4 lh + q of its row = physical slot 4 lh + (q ^ swizzle(row))  (swizzle < 4)
        const float *a0p = As + (wm * 64 + li) * LDS_ROW + lh * (BK / 2);
        const float *a1p = a0p + 32 * LDS_ROW;
        const float *bp = Bs + (wn * 32 + li) * LDS_ROW + lh * (BK / 2);
        const int za0 = lds_swz(wm * 64 + li), za1 = lds_swz(wm * 64 + 32 + li), zb = lds_swz(wn * 32 + li);
        int k0 = kbeg;
        for (;;) {
            // next tile: the following BK rows of this piece, or the first tile of the next piece
            int kn = k0 + BK;
            bool more = kn < kend;
            if (!more && seg + 1 < g.nseg) {
                ++seg;
                A = g.A + bz * g.sA + g.a_off[seg];
                B = g.B + bz * g.sB + g.b_off[seg];
                kn = 0;
                kend = g.klen[seg];
                more = true;
                load_row_scales();
            }
            if (more) load_tiles(kn);
            f4v fa0[4], fa1[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fa0[q] = *reinterpret_cast<const f4v *>(a0p + 4 * (q ^ za0));
                fa1[q] = *reinterpret_cast<const f4v *>(a1p + 4 * (q ^ za1));
                fb[q] = *reinterpret_cast<const f4v *>(bp + 4 * (q ^ zb));
            }
#pragma unroll
            for (int j = 0; j < BK / 2; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[j >> 2][j & 3], fb[j >> 2][j & 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[j >> 2][j & 3], fb[j >> 2][j & 3], acc1, 0, 0, 0);
            }
            __syncthreads();
            if (!more) break;
            store_tiles();
            __syncthreads();
            k0 = kn;
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *C = g.C + (splits > 1 ? (size_t)split * g.split_stride : (size_t)0) + bz * g.sC;
    if (VEC) {
        // stage 64 rows at a time in LDS (operand tiles are dead: the k-loop ended on a barrier), then 16 B row stores
        constexpr int LDC_S = BN + 4;
        float *Cs = smem;  // 64 x 68 floats <= (128 + 64) x 36
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (wm == half) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    Cs[row * LDC_S + wn * 32 + li] = acc0[r];
                    Cs[(32 + row) * LDC_S + wn * 32 + li] = acc1[r];
                }
            }
            __syncthreads();
            const int c4 = (tid % (BN / 4)) * 4, r0 = tid / (BN / 4);
#pragma unroll
            for (int p = 0; p < 64 / (kThreads / (BN / 4)); ++p) {
                const int row = r0 + p * (kThreads / (BN / 4));
                const int gm = m0 + half * 64 + row;
                if (gm < g.M && n0 + c4 < g.N) {
                    f4v v = *reinterpret_cast<const f4v *>(Cs + row * LDC_S + c4);
                    f4v *dst = reinterpret_cast<f4v *>(C + (size_t)gm * g.ldc + n0 + c4);
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
            __syncthreads();
        }
    } else {
        const int col = n0 + wn * 32 + li;
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M) {
                    float *c = C + (size_t)row * g.ldc + col;
                    *c = g.accumulate ? *c + acc0[r] : acc0[r];
                }
                if (row + 32 < g.M) {
                    float *c = C + (size_t)(row + 32) * g.ldc + col;
                    *c = g.accumulate ? *c + acc1[r] : acc1[r];
                }
            }
        }
    }
}

// XCD-aware tile order.  The dispatcher places block b on XCD b % 8 (each XCD has its own 4 MiB L2), so tiles that share
// an operand panel must not be NEIGHBOURS in blockIdx: this bijection hands every XCD one contiguous range of logical
// tile ids, in dispatch order (block b is the (b / 8)-th block of XCD b % 8).  Speed only -- any mapping is correct.
__device__ __forceinline__ unsigned xcd_tile_id(unsigned bid, unsigned nblocks) {
    constexpr unsigned NX = 8;
    const unsigned q = nblocks / NX, r = nblocks % NX, x = bid % NX;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / NX;
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(kThreads, VEC ? 4 : 2) void gemm_f32_mfma(GemmArgs g) {  // (scalar path: 16 + 8 staged loads)
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_ROW];
    // tile order: the N tile varies fastest, so the workgroups that share one A row-panel run back to back and the
    // panel is read from HBM once (dQ = dZ K^T has 18 N tiles per panel); M tiles can be millions (M = sum s^2)
    const int ntn = g.ntiles_n;
    const unsigned tile = xcd_tile_id(blockIdx.x, gridDim.x);
    const int m0 = (int)(tile / ntn) * BM, n0 = (int)(tile % ntn) * BN;
    gemm_tile<TA, TB, VEC>(g, smem, m0, n0, blockIdx.z % g.batch, blockIdx.z / g.batch, gridDim.z / g.batch);
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(kThreads, 4) void gemm_f32_mfma_grouped(GroupedArgs ga, int nsplits) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_ROW];
    const unsigned tile = xcd_tile_id(blockIdx.x, gridDim.x);
    int panel = (int)(tile / ga.tiles_per_panel), t = (int)(tile % ga.tiles_per_panel);
    if (ga.window > 1) {
        const int per_window = ga.window * ga.tiles_per_panel;
        const int w = (int)(tile / per_window), r = (int)(tile % per_window);
        const int first = w * ga.window;
        const int wl = (ga.npanels - first < ga.window) ? ga.npanels - first : ga.window;  // the last window may be short
        t = r / wl;
        panel = first + r % wl;
    }
    int grp = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
        if (i < ga.ngroups && t >= ga.tile_prefix[i]) grp = i;
    const int q = t - ga.tile_prefix[grp];
    if (ga.panel_is_split)
        gemm_tile<TA, TB, VEC>(ga.g[grp], smem, q * BM, 0, 0, panel, nsplits);
    else
        gemm_tile<TA, TB, VEC>(ga.g[grp], smem, panel * BM, q * BN, 0, 0, 1);
}

// Free-form grouped launch: up to kMaxGroups GEMMs of UNRELATED shapes in one grid (the small per-(node,x) / per-node /
// compact products of a fused SMP level: a few tiles each, latency-bound alone, concurrent here).  tile_prefix[i] = first
// tile of group i in the grid.  !TA: a group's tiles are its (M tile, N tile) pairs, C written directly.  TA (reductions
// over the rows): tiles are (split, M tile, N tile) with the group's own split count; partial images go to g.C + split *
// split_stride and are folded by the caller in a fixed order.
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(kThreads, 4) void gemm_f32_mfma_free(GroupedArgs ga) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_ROW];
    const int tile = (int)blockIdx.x;
    int grp = 0;
#pragma unroll
    for (int i = 1; i < kMaxGroups; ++i)
        if (i < ga.ngroups && tile >= ga.tile_prefix[i]) grp = i;
    const GemmArgs &g = ga.g[grp];
    int q = tile - ga.tile_prefix[grp];
    const int ntn = g.ntiles_n, mt = (g.M + BM - 1) / BM;
    const int n0 = (q % ntn) * BN;
    q /= ntn;
    if (TA)
        gemm_tile<TA, TB, VEC>(g, smem, (q % mt) * BM, n0, 0, q / mt, 2);  // (2: "write a partial image", even for one split)
    else
        gemm_tile<TA, TB, VEC>(g, smem, q * BM, n0, 0, 0, 1);
}

// dst[c][i] = sum of part[s][i] over the c-th chunk of splits (chunk = ceil(splits / gridDim.y)); with gridDim.y == 1 and
// accumulate it is the final  C[i] (+)= sum_s part[s][i].  Two passes of this kernel fold thousands of split-K partials
// with full-chip parallelism and a fixed summation order (chunks in order, splits in order inside a chunk).
__global__ void splitk_reduce(const float *__restrict__ part, float *__restrict__ dst, size_t n, int splits, int chunk,
                              int accumulate) {
    const int c = blockIdx.y;
    const int s0 = c * chunk, s1 = (s0 + chunk < splits) ? s0 + chunk : splits;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = s0; k < s1; ++k) s += part[(size_t)k * n + i];
        float *o = dst + (size_t)c * n + i;
        *o = accumulate ? *o + s : s;
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
    return gemm_rs(ctx, ta, tb, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, batch, accumulate, nullptr, 0, -1);
}

// the same with op(A) scaled per row: rs[row * rs_ld + scol] (see GemmArgs::rs)
gf_status gemm_rs(gf_ctx *ctx, bool ta, bool tb, int M, int N, int K, const float *A, int lda, long long sA, const float *B,
                  int ldb, long long sB, float *C, int ldc, long long sC, int batch, int accumulate, const float *rs,
                  int rs_ld, int scol) {
    if (M <= 0 || N <= 0 || batch <= 0) return GF_OK;
    if (scol < 0) rs = nullptr;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA = sA; g.sB = sB; g.sC = sC; g.batch = batch; g.accumulate = accumulate;
    g.nseg = 0; g.split_stride = (long long)M * N;
    g.rs = rs; g.rs_ld = rs_ld; g.scol[0] = scol; g.scol[1] = g.scol[2] = g.scol[3] = -1;
    const int gx = (N + BN - 1) / BN, gy = (M + BM - 1) / BM;
    // split K when the output has too few tiles to fill 256 CUs and K is long (e.g. dB = A^T dC of the K-projection)
    int splits = 1;
    const long long tiles = (long long)gx * gy * batch;
    if (batch == 1 && tiles < 512 && K >= 8 * BK) {
        splits = (int)((2048 + tiles - 1) / tiles);  // about 8 workgroups per CU: each one is a latency-bound k-loop
        const int maxs = K / (8 * BK);              // at least eight k-steps per workgroup: fewer partial images to fold
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
        gf_status st = ensure_ws(ctx, sizeof(float) * ((size_t)splits + (splits + 31) / 32) * M * N + 256);
        if (st != GF_OK) return st;
        part = static_cast<float *>(ctx->ws);
        g.C = part;
        g.ldc = N;
        g.accumulate = 0;
    }
    g.ntiles_n = gx;
    const dim3 grid((unsigned)((size_t)gx * gy), 1, batch * splits);
    // vector path: 16-byte aligned operands, extents and leading dimensions in multiples of 4
    // (an extent only has to be a multiple of 4 along the directions that are actually loaded as float4:
    //  M when A is stored m-contiguous (TA), K when A or B is k-contiguous (!TA or TB), N always (B !TB and the C rows))
    const bool vec = (((uintptr_t)A | (uintptr_t)B | (uintptr_t)g.C) & 15) == 0 && (N % 4 == 0) && (!ta || M % 4 == 0) &&
                     ((ta && !tb) || K % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && (g.ldc % 4 == 0) &&
                     (sA % 4 == 0) && (sB % 4 == 0) && (sC % 4 == 0);
#define GF_GEMM_LAUNCH(TA_, TB_, NAME)                                                                         \
    do {                                                                                                       \
        if (vec)                                                                                               \
            GF_LAUNCH(ctx, NAME, (gemm_f32_mfma<TA_, TB_, true>), grid, dim3(kThreads), 0, g);                 \
        else                                                                                                   \
            GF_LAUNCH(ctx, NAME, (gemm_f32_mfma<TA_, TB_, false>), grid, dim3(kThreads), 0, g);                \
    } while (0)
    if (!ta && !tb) GF_GEMM_LAUNCH(false, false, "gemm_nn");
    if (!ta && tb) GF_GEMM_LAUNCH(false, true, "gemm_nt");
    if (ta && !tb) GF_GEMM_LAUNCH(true, false, "gemm_tn");
    if (ta && tb) GF_GEMM_LAUNCH(true, true, "gemm_tt");
#undef GF_GEMM_LAUNCH
    if (splits > 1) {
        const size_t n = (size_t)M * N;
        size_t blocks = (n + 255) / 256;
        const unsigned gx1 = (unsigned)(blocks > 4096 ? 4096 : blocks);
        if (splits > 64) {  // two passes: [splits] -> [nchunks] partials (stored behind the split partials) -> C
            const int chunk = 32, nchunks = (splits + chunk - 1) / chunk;
            float *part2 = part + (size_t)splits * n;
            GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, nchunks), dim3(256), 0, part, part2, n, splits, chunk, 0);
            GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part2, C, n, nchunks, nchunks, accumulate);
        } else {
            GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part, C, n, splits, splits, accumulate);
        }
    }
    return GF_OK;
}

// ---- grouped launches -----------------------------------------------------------------------------------------
// `specs` describe n <= kMaxGroups GEMMs (op flags shared).  rows_shared:
//   panel_is_split == 0: every group has M == rows_shared (the row panels are shared; N tiles of all groups interleave)
//   panel_is_split == 1: every group reduces over K == rows_shared (split-K row ranges are shared); the per-split
//                        partial images are laid out [split][total] with group g at offset c_off[g]; `dest` receives
//                        the ordered sum (total = sum M_g N_g contiguous floats).
static bool spec_vec_ok(const GemmSpec &s, bool ta, bool tb) {
    bool ok = (((uintptr_t)s.A | (uintptr_t)s.B | (uintptr_t)s.C) & 15) == 0 && (s.N % 4 == 0) && (!ta || s.M % 4 == 0) &&
              (s.lda % 4 == 0) && (s.ldb % 4 == 0) && (s.ldc % 4 == 0);
    if (s.nseg == 0) ok = ok && ((ta && !tb) || s.K % 4 == 0);
    for (int i = 0; i < s.nseg; ++i) ok = ok && (s.klen[i] % BK == 0) && (s.a_off[i] % 4 == 0) && (s.b_off[i] % 4 == 0);
    return ok;
}

static void fill_args(GemmArgs *g, const GemmSpec &s) {
    g->A = s.A; g->B = s.B; g->C = s.C; g->M = s.M; g->N = s.N; g->K = s.K; g->lda = s.lda; g->ldb = s.ldb; g->ldc = s.ldc;
    g->sA = g->sB = g->sC = 0; g->batch = 1; g->accumulate = 0; g->kchunk = s.K > 0 ? (s.K + BK - 1) / BK * BK : BK;
    g->ntiles_n = (s.N + BN - 1) / BN; g->split_stride = 0; g->nseg = s.nseg; g->nsplits = 1;
    for (int i = 0; i < 4; ++i) {
        g->a_off[i] = s.a_off[i];
        g->b_off[i] = s.b_off[i];
        g->klen[i] = s.klen[i];
        g->scol[i] = s.rs ? s.scol[i] : -1;
    }
    g->rs = s.rs;
    g->rs_ld = s.rs_ld;
}

template <bool TA, bool TB>
static gf_status launch_grouped(gf_ctx *ctx, const GroupedArgs &ga, unsigned grid, int nsplits, const char *name) {
    GF_LAUNCH(ctx, name, (gemm_f32_mfma_grouped<TA, TB, true>), dim3(grid), dim3(kThreads), 0, ga, nsplits);
    return GF_OK;
}

bool gemm_grouped_supported(const GemmSpec *specs, int n, bool ta, bool tb) {
    if (n < 1 || n > kMaxGroups) return false;
    for (int i = 0; i < n; ++i)
        if (!spec_vec_ok(specs[i], ta, tb)) return false;
    return true;
}

gf_status gemm_grouped_rows(gf_ctx *ctx, bool ta, bool tb, const GemmSpec *specs, int n, int rows, int accumulate) {
    GroupedArgs ga;
    ga.ngroups = n;
    ga.panel_is_split = 0;
    ga.npanels = (rows + BM - 1) / BM;
    ga.window = 8;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (specs[i].M != rows) return fail(ctx, GF_ERR_INVALID, "gemm_grouped_rows: group %d has M=%d, expected %d", i, specs[i].M, rows);
        fill_args(&ga.g[i], specs[i]);
        ga.g[i].accumulate = accumulate;   // (C += : every element of C belongs to one tile of one group)
        ga.tile_prefix[i] = tiles;
        tiles += (specs[i].N + BN - 1) / BN;
    }
    for (int i = n; i < kMaxGroups; ++i) ga.tile_prefix[i] = 1 << 30;
    ga.tiles_per_panel = tiles;
    const unsigned grid = (unsigned)((size_t)((rows + BM - 1) / BM) * tiles);
    if (!ta && !tb) return launch_grouped<false, false>(ctx, ga, grid, 1, "gemm_nn");
    if (!ta && tb) return launch_grouped<false, true>(ctx, ga, grid, 1, "gemm_nt");
    return fail(ctx, GF_ERR_UNSUPPORTED, "gemm_grouped_rows: transposed A is served by gemm_grouped_splitk");
}

gf_status gemm_grouped_splitk(gf_ctx *ctx, const GemmSpec *specs, int n, int rows, float *dest, int accumulate) {
    GroupedArgs ga;
    ga.ngroups = n;
    ga.panel_is_split = 1;
    ga.npanels = 0;   // (set to the split count below)
    ga.window = 1;
    int tiles = 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (specs[i].K != rows || specs[i].N > BN)
            return fail(ctx, GF_ERR_INVALID, "gemm_grouped_splitk: group %d must reduce over %d rows with N <= %d", i, rows, BN);
        ga.tile_prefix[i] = tiles;
        tiles += (specs[i].M + BM - 1) / BM;
        total += (size_t)specs[i].M * specs[i].N;
    }
    for (int i = n; i < kMaxGroups; ++i) ga.tile_prefix[i] = 1 << 30;
    ga.tiles_per_panel = tiles;
    int splits = (4096 + tiles - 1) / tiles;
    const int maxs = rows / (2 * BK);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int kchunk = (rows + splits - 1) / splits;
    kchunk = (kchunk + BK - 1) / BK * BK;
    splits = (rows + kchunk - 1) / kchunk;
    const int chunk = 32, nchunks = (splits + chunk - 1) / chunk;
    gf_status st = ensure_ws(ctx, sizeof(float) * ((size_t)splits + nchunks) * total + 256);
    if (st != GF_OK) return st;
    float *part = static_cast<float *>(ctx->ws);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        fill_args(&ga.g[i], specs[i]);
        ga.g[i].kchunk = kchunk;
        ga.g[i].C = part + off;        // partial image of this group inside one split's block
        ga.g[i].ldc = specs[i].N;
        ga.g[i].split_stride = (long long)total;
        off += (size_t)specs[i].M * specs[i].N;
    }
    ga.npanels = splits;
    const unsigned grid = (unsigned)((size_t)splits * tiles);
    GF_LAUNCH(ctx, "gemm_tn", (gemm_f32_mfma_grouped<true, false, true>), dim3(grid), dim3(kThreads), 0, ga, splits + 1);
    // (splits + 1: the tile function treats nsplits > 1 as "write partial images"; a single split still goes through
    //  the ordered reduction below so the destination handling stays in one place)
    size_t blocks = (total + 255) / 256;
    const unsigned gx1 = (unsigned)(blocks > 4096 ? 4096 : blocks);
    if (splits > 64) {
        float *part2 = part + (size_t)splits * total;
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, nchunks), dim3(256), 0, part, part2, total, splits, chunk, 0);
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part2, dest, total, nchunks, nchunks, accumulate);
    } else {
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part, dest, total, splits, splits, accumulate);
    }
    return GF_OK;
}

// n <= kMaxGroups unrelated products op(A_i)[M_i,K_i] B_i -> C_i in one launch (NN or NT; C written, not accumulated).
gf_status gemm_grouped_free(gf_ctx *ctx, bool tb, const GemmSpec *specs, int n, const char *name) {
    if (n < 1 || n > kMaxGroups) return fail(ctx, GF_ERR_INVALID, "gemm_grouped_free: %d groups", n);
    GroupedArgs ga;
    ga.ngroups = 0;
    ga.panel_is_split = 0;
    ga.npanels = 0;
    ga.window = 1;
    ga.tiles_per_panel = 0;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (specs[i].M <= 0 || specs[i].N <= 0) continue;
        if (!spec_vec_ok(specs[i], false, tb) || specs[i].nseg != 0)
            return fail(ctx, GF_ERR_UNSUPPORTED, "gemm_grouped_free: group %d is not 16-byte aligned / a multiple of 4", i);
        const int k = ga.ngroups++;
        fill_args(&ga.g[k], specs[i]);
        ga.g[k].nsplits = 1;
        ga.tile_prefix[k] = tiles;
        tiles += ((specs[i].M + BM - 1) / BM) * ((specs[i].N + BN - 1) / BN);
    }
    for (int i = ga.ngroups; i < kMaxGroups; ++i) ga.tile_prefix[i] = 1 << 30;
    if (tiles == 0) return GF_OK;
    if (tb)
        GF_LAUNCH(ctx, name, (gemm_f32_mfma_free<false, true, true>), dim3((unsigned)tiles), dim3(kThreads), 0, ga);
    else
        GF_LAUNCH(ctx, name, (gemm_f32_mfma_free<false, false, true>), dim3((unsigned)tiles), dim3(kThreads), 0, ga);
    return GF_OK;
}

// n <= kMaxGroups unrelated reductions A_i^T B_i (A_i [K_i, M_i], B_i [K_i, N_i]) in one launch, each split over its OWN rows
// into partial images laid out from `part` on: group i at out[i].part, out[i].splits images of out[i].n floats, back to back.
// The split count of a group depends on its row count only (results reproducible).  The caller folds the images in order.
gf_status gemm_grouped_free_tn(gf_ctx *ctx, const GemmSpec *specs, int n, float *part, size_t part_floats, FoldGroup *out,
                               const char *name) {
    if (n < 1 || n > kMaxGroups) return fail(ctx, GF_ERR_INVALID, "gemm_grouped_free_tn: %d groups", n);
    GroupedArgs ga;
    ga.ngroups = 0;
    ga.panel_is_split = 1;
    ga.npanels = 0;
    ga.window = 1;
    ga.tiles_per_panel = 0;
    int tiles = 0;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        const GemmSpec &sp = specs[i];
        out[i].part = part + off;
        out[i].n = (size_t)sp.M * sp.N;
        out[i].splits = 0;
        if (sp.M <= 0 || sp.N <= 0) continue;
        if (!spec_vec_ok(sp, true, false) || sp.nseg != 0)
            return fail(ctx, GF_ERR_UNSUPPORTED, "gemm_grouped_free_tn: group %d is not 16-byte aligned / a multiple of 4", i);
        // at least four k-steps per workgroup, at most `cap` partial images.  (64 until round 3: the per-(node,x) products of level 3
        // -- 168,000 rows -- then ran as 128 workgroups of 82 k-steps each on 256 CUs, one latency chain 140 us long)
        constexpr int cap = 256;   // (cfg3: 0.30 ms at 64, 0.23 at 128, 0.22 at 256, 0.23 at 384 + a slower fold)
        int splits = sp.K / (4 * BK);
        if (splits > cap) splits = cap;
        if (splits < 1) splits = 1;
        int kchunk = ((sp.K + splits - 1) / splits + BK - 1) / BK * BK;
        if (kchunk < BK) kchunk = BK;
        splits = sp.K > 0 ? (sp.K + kchunk - 1) / kchunk : 1;
        const int k = ga.ngroups++;
        fill_args(&ga.g[k], sp);
        ga.g[k].kchunk = kchunk;
        ga.g[k].nsplits = splits;
        ga.g[k].C = part + off;
        ga.g[k].ldc = sp.N;
        ga.g[k].split_stride = (long long)out[i].n;
        ga.tile_prefix[k] = tiles;
        tiles += splits * ((sp.M + BM - 1) / BM) * ((sp.N + BN - 1) / BN);
        out[i].splits = splits;
        off += (size_t)splits * out[i].n;
    }
    if (off > part_floats) return fail(ctx, GF_ERR_NOMEM, "gemm_grouped_free_tn: %zu floats of partial images, %zu available", off, part_floats);
    for (int i = ga.ngroups; i < kMaxGroups; ++i) ga.tile_prefix[i] = 1 << 30;
    if (tiles == 0) return GF_OK;
    GF_LAUNCH(ctx, name, (gemm_f32_mfma_free<true, false, true>), dim3((unsigned)tiles), dim3(kThreads), 0, ga);
    return GF_OK;
}

// part[s][i] (s < splits, i < total) -> dest[i] (+)= sum over s in a fixed order: one pass up to 64 partial images, else two
// (chunks of 32).  `part` must have room for splits + ceil(splits / 32) images.
gf_status splitk_fold(gf_ctx *ctx, const float *part, float *dest, size_t total, int splits, int accumulate) {
    const int chunk = 32, nchunks = (splits + chunk - 1) / chunk;
    size_t blocks = (total + 255) / 256;
    const unsigned gx1 = (unsigned)(blocks > 4096 ? 4096 : blocks);
    if (splits > 64) {
        float *part2 = const_cast<float *>(part) + (size_t)splits * total;
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, nchunks), dim3(256), 0, part, part2, total, splits, chunk, 0);
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part2, dest, total, nchunks, nchunks, accumulate);
    } else {
        GF_LAUNCH(ctx, "splitk_reduce", splitk_reduce, dim3(gx1, 1), dim3(256), 0, part, dest, total, splits, splits, accumulate);
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

gf_status gf_custommatmultensor_forward_f32(gf_ctx *ctx, const float *W, const float *T, float *Out, long long rows, int V,
                                            int Kout) {
    gf_status st = need(ctx, W && T && Out && rows > 0 && rows <= 0x7fffffffLL && V > 0 && Kout > 0,
                        "gf_custommatmultensor_forward_f32: bad argument");
    if (st != GF_OK) return st;
    // Out[rows,Kout] = T[rows,V] W^T[V,Kout]
    return gf::gemm(ctx, false, true, (int)rows, Kout, V, T, V, 0, W, V, 0, Out, Kout, 0, 1, 0);
}

gf_status gf_custommatmultensor_backward_f32(gf_ctx *ctx, const float *G, const float *W, const float *T, float *dW,
                                             float *dT, long long rows, int V, int Kout, int accumulate) {
    gf_status st = need(ctx, G && W && T && rows > 0 && rows <= 0x7fffffffLL && V > 0 && Kout > 0,
                        "gf_custommatmultensor_backward_f32: bad argument");
    if (st != GF_OK) return st;
    if (dW) {  // dW[Kout,V] (+)= G^T[Kout,rows] T[rows,V]
        st = gf::gemm(ctx, true, false, Kout, V, (int)rows, G, Kout, 0, T, V, 0, dW, V, 0, 1, accumulate);
        if (st != GF_OK) return st;
    }
    if (dT)  // dT[rows,V] (+)= G[rows,Kout] W[Kout,V]
        st = gf::gemm(ctx, false, false, (int)rows, V, Kout, G, Kout, 0, W, V, 0, dT, V, 0, 1, accumulate);
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
