// contract_families.hip -- RisiContraction_4, _10 and _50 for gfx950, factorised O(N^3 C) form.
//
// Replaces GraphFlow/RisiContraction_4.h:68-173, RisiContraction_10.h:73-225 and RisiContraction_50.h:73-802.  The
// reference walks the full 5-index space (O(N^5 C), 50 predicated updates per point for _50); here every case is
// written in its factorised form: pair marginals of P, the same marginals weighted by
// the row sums r, column sums q or diagonal dg of A, three diagonals of P, and N x N products with A.  No `A > 0`
// gate in these families (RisiContraction_50.h:83-97): A is used raw, negative entries included.
//
// Structure (K = 10 or 50; cases that a family does not have compile away):
//   forward   fam_adj     per graph: r, q, dg, tot, tr
//             fam_tables  thread per (i,j,f): 12 pair tables in one pass over the third index
//             fam_vectors thread per (i,f) / per f: single-index marginals and scalars
//             fam_forward thread per (x,y,f): all K outputs, one loop over the contracted index
//   backward  fam_bwd_scalars, fam_bwd_tables (thread per (i,j,f): X_ab, X_ac, X_bc, Z_bc, Z_ac, Z_ab),
//             fam_backward thread per (a,b,c,f): O(1) combination.
// These are "table" kernels (coalesced over the channel axis, tables re-read through L2), not the LDS-staged slab
// kernels RisiContraction_18 has; they are the correct-first implementation of the rarely used families.
#include <cstdint>

#include <cstdlib>

#include "gf_internal.h"
#include "r18_device.h"

// Non-temporal stores of the families' big results (A/B: -DGF_NT_FAM=<mask>): 1 fam50_tables_out's fifteen slices of Out,
// 2 fam50_forward_mfma's slices of Out, 4 fam_backward_rows' dP, 8 the `_4` slab kernels' streams (P, G in; Out, dP out), 16 the `_10`
// one-stream-per-graph kernels' streams.  (r18_device.h: why `nt` on the streams pays.)
#ifndef GF_NT_FAM
#define GF_NT_FAM 31
#endif
namespace gf {
namespace {

// Blocks are dispatched round-robin over the eight XCDs (block b runs on XCD b % 8, each with its own 4 MiB L2), while these
// kernels re-read per-graph tables: the logical block handed to (XCD x, k-th block of that XCD) is the k-th of the x-th
// eighth of the grid, so the threads of one graph -- consecutive logical blocks -- share an L2.  A bijection on [0, gridDim).
__device__ __forceinline__ size_t xcd_block() {
    const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = blockIdx.x % 8;
    return (size_t)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8);
}
#define GRID_STRIDE(idx, total) \
    for (size_t idx = xcd_block() * (size_t)blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

unsigned grid_for(size_t total) {
    size_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 262144 ? 262144 : (blocks == 0 ? 1 : blocks));
}

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_4 (no adjacency): k0 (a,b) sum_c | k1 (b,c) sum_a | k2 (a,c) at a==b | k3 (a,b) at b==c
// ---------------------------------------------------------------------------------------------------------------
__global__ void r4_forward(const float *__restrict__ P, float *__restrict__ Out, int N, int C, size_t total) {
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int y = t % N;
        t /= N;
        const int x = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * (size_t)N * N * N * C;
        float sab = 0.f, sbc = 0.f;
        for (int s = 0; s < N; ++s) {
            sab += Pg[(((size_t)x * N + y) * N + s) * C + f];
            sbc += Pg[(((size_t)s * N + x) * N + y) * C + f];
        }
        float *o = Out + (((size_t)g * N + x) * N + y) * (size_t)(4 * C) + f;
        o[0 * C] = sab;
        o[1 * C] = sbc;
        o[2 * C] = Pg[(((size_t)x * N + x) * N + y) * C + f];
        o[3 * C] = Pg[(((size_t)x * N + y) * N + y) * C + f];
    }
}

__global__ void r4_backward(const float *__restrict__ G, float *__restrict__ dP, int N, int C, size_t total,
                            int accumulate) {
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int c = t % N;
        t /= N;
        const int b = t % N;
        t /= N;
        const int a = t % N;
        const size_t g = t / N;
        const float *Gg = G + g * (size_t)N * N * 4 * C;
        float v = Gg[(((size_t)a * N + b) * 4 + 0) * C + f] + Gg[(((size_t)b * N + c) * 4 + 1) * C + f];
        if (a == b) v += Gg[(((size_t)a * N + c) * 4 + 2) * C + f];
        if (b == c) v += Gg[(((size_t)a * N + b) * 4 + 3) * C + f];
        if (accumulate)
            dP[idx] += v;
        else
            dP[idx] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_10 / _50
// ---------------------------------------------------------------------------------------------------------------
// channel vector width of the table kernels: 4 (16-byte loads and stores, C % 4 == 0 and aligned buffers) or 1
using vf4 = __attribute__((ext_vector_type(4))) float;
template <int VW>
struct Vec;
template <>
struct Vec<1> {
    using T = float;
    static __device__ __forceinline__ T ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, T v) { *p = v; }
    template <int SITE>
    static __device__ __forceinline__ void st_s(float *p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
};
template <>
struct Vec<4> {
    using T = vf4;
    static __device__ __forceinline__ T ld(const float *p) { return *reinterpret_cast<const vf4 *>(p); }
    static __device__ __forceinline__ void st(float *p, T v) { *reinterpret_cast<vf4 *>(p) = v; }
    // results written once and read by a later kernel (GF_NT_FAM, below): non-temporal
    template <int SITE>
    static __device__ __forceinline__ void st_s(float *p, T v) {
        if constexpr ((GF_NT_FAM & SITE) != 0) __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(p));
        else *reinterpret_cast<vf4 *>(p) = v;
    }
    template <int SITE>
    static __device__ __forceinline__ T ld_s(const float *p) {
        if constexpr ((GF_NT_FAM & SITE) != 0) return __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p));
        else return *reinterpret_cast<const vf4 *>(p);
    }
    static __device__ __forceinline__ T zero() { return vf4{0.f, 0.f, 0.f, 0.f}; }
};

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_4 as slab streams (round 4; C % 4 == 0, 4 <= C / 4 <= 64 lanes per position dividing 64, 16-byte aligned buffers).
// Every output of the family is owned by ONE middle index b: Out[a,b,0] = sum_c P[a,b,c] and Out[a,b,3] = P[a,b,b] for all a,
// Out[b,c,1] = sum_a P[a,b,c] and Out[b,c,2] = P[b,b,c] for all c -- so a workgroup per (graph, b) streams the slab P[g][:, b, :, :]
// ONCE (N rows of N C contiguous floats), with no workspace and no second pass: lane = (position group cg, channel quad fl), a wave
// load covers PPW = 64 / (C / 4) positions x C channels as 16 bytes per lane, wave w owns rows a = w, w + 4, ...; the sums over c are
// a register sum plus a butterfly over the position groups, the sums over a stay in registers per wave and meet in LDS in wave order.
// The thread-per-output kernels above read P twice through 4-byte loads (0.62 ms at N = 24, C = 32, batch 256 forward + backward).
// Backward: dP[a,b,c] = G0[a,b] + G1[b,c] + [a = b] G2[b,c] + [b = c] G3[a,b]: the (b, c)-indexed terms are lane-resident, a row costs two
// broadcast loads and N C floats of stores.
// ---------------------------------------------------------------------------------------------------------------
template <int SLOTS>   // position slots per lane: ceil(N / PPW)
__global__ __launch_bounds__(256) void r4_fwd_slab(const float *__restrict__ P, float *__restrict__ Out, int N, int C) {
    extern __shared__ __attribute__((aligned(16))) float r4_smem[];   // [4 waves][N][C] partial S_bc
    const int lpc = C >> 2, ppw = 64 / lpc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane / lpc, fl = lane % lpc;
    const size_t blk = xcd_block();
    const int b = (int)(blk % N);
    const size_t g = blk / N;
    const size_t NC = (size_t)N * C;
    const float *slab = P + g * NC * N * N + (size_t)b * NC + 4 * fl;   // row a at + a N NC
    float *outg = Out + g * (size_t)N * N * 4 * C + 4 * fl;
    vf4 sbc[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) sbc[i] = vf4{0.f, 0.f, 0.f, 0.f};
    for (int a = wave; a < N; a += 4) {
        const float *row = slab + (size_t)a * N * NC;
        vf4 v[SLOTS];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            v[i] = c < N ? Vec<4>::ld_s<8>(row + (size_t)c * C) : vf4{0.f, 0.f, 0.f, 0.f};
        }
        vf4 sab = vf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            sab += v[i];
            sbc[i] += v[i];
        }
        // sum over the position groups of the wave (lane bits >= log2(lpc))
        for (int m = lpc; m < 64; m <<= 1) {
            sab[0] += __shfl_xor(sab[0], m);
            sab[1] += __shfl_xor(sab[1], m);
            sab[2] += __shfl_xor(sab[2], m);
            sab[3] += __shfl_xor(sab[3], m);
        }
        float *o = outg + ((size_t)a * N + b) * 4 * C;
        if (cg == 0) Vec<4>::st_s<8>(o + 0 * C, sab);                       // Out[a,b,0]
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            if (c == b) Vec<4>::st_s<8>(o + 3 * C, v[i]);                    // Out[a,b,3] = P[a,b,b]
            if (a == b && c < N) Vec<4>::st_s<8>(outg + ((size_t)b * N + c) * 4 * C + 2 * C, v[i]);   // Out[b,c,2] = P[b,b,c]
        }
    }
    // S_bc[b,c] = sum_a: the four waves' partials in wave order
    vf4 *part = reinterpret_cast<vf4 *>(r4_smem);   // [4][N][lpc]
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = i * ppw + cg;
        if (c < N) part[((size_t)wave * N + c) * lpc + fl] = sbc[i];
    }
    __syncthreads();
    for (int i = tid; i < N * lpc; i += 256) {
        vf4 t = part[i];
        for (int w = 1; w < 4; ++w) t += part[(size_t)w * N * lpc + i];
        const int c = i / lpc, q = i % lpc;
        Vec<4>::st_s<8>(Out + g * (size_t)N * N * 4 * C + ((size_t)b * N + c) * 4 * C + 1 * C + 4 * q, t);   // Out[b,c,1]
    }
}

template <int SLOTS, bool ACC>
__global__ __launch_bounds__(256) void r4_bwd_slab(const float *__restrict__ G, float *__restrict__ dP, int N, int C) {
    const int lpc = C >> 2, ppw = 64 / lpc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane / lpc, fl = lane % lpc;
    const size_t blk = xcd_block();
    const int b = (int)(blk % N);
    const size_t g = blk / N;
    const size_t NC = (size_t)N * C;
    const float *Gg = G + g * (size_t)N * N * 4 * C + 4 * fl;
    float *slab = dP + g * NC * N * N + (size_t)b * NC + 4 * fl;
    vf4 y[SLOTS], y2[SLOTS];   // G1[b,c] and G1[b,c] + G2[b,c] (the row a = b)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int c = i * ppw + cg;
        const float *gc = Gg + ((size_t)b * N + (c < N ? c : 0)) * 4 * C;
        y[i] = Vec<4>::ld_s<8>(gc + 1 * C);
        y2[i] = y[i] + Vec<4>::ld_s<8>(gc + 2 * C);
    }
    for (int a = wave; a < N; a += 4) {
        const float *ga = Gg + ((size_t)a * N + b) * 4 * C;
        const vf4 g0 = Vec<4>::ld_s<8>(ga), g3 = Vec<4>::ld_s<8>(ga + 3 * C);
        float *row = slab + (size_t)a * N * NC;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            if (c < N) {
                vf4 v = g0 + (a == b ? y2[i] : y[i]);
                if (c == b) v += g3;
                if (ACC) v += Vec<4>::ld(row + (size_t)c * C);
                Vec<4>::st_s<8>(row + (size_t)c * C, v);
            }
        }
    }
}

// the slab kernels serve: C % 4 == 0, C / 4 a power of two <= 64, N <= 16 positions slots per lane, aligned buffers
static bool r4_slab_ok(int N, int C, const void *p0, const void *p1) {
    if (C % 4 != 0 || C < 4) return false;
    const int lpc = C / 4;
    if (lpc > 64 || (lpc & (lpc - 1)) != 0) return false;
    const int ppw = 64 / lpc, slots = (N + ppw - 1) / ppw;
    return slots <= 16 && ((((uintptr_t)p0) | ((uintptr_t)p1)) & 15) == 0;
}

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_10 as ONE stream per graph (round 4, second session; C % 4 == 0 with C / 4 <= 32 lanes per position, N <= 32, at most
// four position slots per lane, 16-byte aligned buffers, batches that fill the chip).  The ten slices are the "1+1+1" cases
// (RisiContraction_10.h:94-142): with S_ab, S_ac, S_bc the pair marginals of P, Va / Vb / Vc their vector marginals, s the total, r / q
// the row / column sums of A and tot their total,
//   Out[.,.,0..9] = S_ab tot | S_ac tot | Va[x] r[y] | Va[x] q[y] | S_bc tot | Vb[x] r[y] | Vb[x] q[y] | Vc[x] r[y] | Vc[x] q[y] | s A[x,y].
// No 1-D slab owns all three pair marginals, but a whole graph's do fit ONE workgroup: P[g] (N^3 C floats, 1.8 MB at N = 24, C = 32) is
// streamed once by a workgroup of ceil(N / 2) waves, wave w owning the rows b = w and w + nw of every slab P[g][a]: S_ab[a,b] is a
// register sum plus a butterfly over the wave's position groups, S_bc[b,c] accumulates in the owner's registers over a, and S_ac[a,:]
// = sum_b meets in LDS once per a (the waves' partials in wave order behind ONE barrier: double-buffered) -- where it is written out,
// folded into Vc, and forgotten.  The rows of a + 1 are requested before the sums of a.  The vector slices are written from LDS at the
// end.  The table kernels read P three times through L2 (1.48 x fetched) and wrote / re-read the marginals: 0.25 -> see DESIGN 4.2.
// Backward: dP[a,b,c] = tot (G0[a,b] + G1[a,c] + G4[b,c]) + xa[a] + xb[b] + xc[c] + s with xa[x] = sum_y G2[x,y] r[y] + G3[x,y] q[y]
// (xb from G5, G6; xc from G7, G8) and s = sum G9[x,y] A[x,y]: one pass over seven slices of G for the vectors, then dP streamed out
// with the (b, c)-indexed terms lane-resident.
// ---------------------------------------------------------------------------------------------------------------
struct R10Lds {   // float offsets into the workgroup's LDS
    int A, r, q, tot, Va, Vb, Vc, S, pva, pac, total;
};
__host__ __device__ inline R10Lds r10_lds(int N, int lpc, int nw) {
    R10Lds L;
    L.A = 0;
    L.r = N * N;
    L.q = L.r + N;
    L.tot = L.q + N;
    L.Va = (L.tot + 4 + 3) & ~3;
    L.Vb = L.Va + 4 * N * lpc;
    L.Vc = L.Vb + 4 * N * lpc;
    L.S = L.Vc + 4 * N * lpc;
    L.pva = L.S + 4 * lpc;
    L.pac = L.pva + 4 * 2 * nw * lpc;
    L.total = L.pac + 4 * 2 * nw * N * lpc;
    return L;
}
// the adjacency, its row / column sums and their total into LDS (every thread of the workgroup; ends behind a barrier)
__device__ __forceinline__ float r10_adjacency(const float *__restrict__ Ag, float *sm, const R10Lds &L, int N) {
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) sm[L.A + i] = Ag[i];
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float rs = 0.f, qs = 0.f;
        for (int j = 0; j < N; ++j) {
            rs += sm[L.A + threadIdx.x * N + j];
            qs += sm[L.A + j * N + threadIdx.x];
        }
        sm[L.r + threadIdx.x] = rs;
        sm[L.q + threadIdx.x] = qs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < N; ++i) t += sm[L.r + i];
        sm[L.tot] = t;
    }
    __syncthreads();
    return sm[L.tot];
}
__device__ __forceinline__ vf4 r10_butterfly(vf4 v, int lpc) {   // sum over the position groups of the wave (lane bits >= log2(lpc))
    for (int m = lpc; m < 64; m <<= 1) {
        v[0] += __shfl_xor(v[0], m);
        v[1] += __shfl_xor(v[1], m);
        v[2] += __shfl_xor(v[2], m);
        v[3] += __shfl_xor(v[3], m);
    }
    return v;
}

template <int SLOTS, int MAXT>   // position slots per lane: ceil(N / PPW); threads: 768 (N <= 24: 168 registers) or 1024 (128)
__global__ __launch_bounds__(MAXT) void r10_fwd_graph(const float *__restrict__ P, const float *__restrict__ A, float *__restrict__ Out, int N, int C,
                                                      int nw) {
    extern __shared__ __attribute__((aligned(16))) float r10_smem[];
    const int lpc = C >> 2, ppw = 64 / lpc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane / lpc, fl = lane % lpc;
    const size_t g = blockIdx.x;
    const size_t NC = (size_t)N * C;
    const R10Lds L = r10_lds(N, lpc, nw);
    const float tot = r10_adjacency(A + g * N * N, r10_smem, L, N);
    vf4 *sVa = reinterpret_cast<vf4 *>(r10_smem + L.Va), *sVb = reinterpret_cast<vf4 *>(r10_smem + L.Vb), *sVc = reinterpret_cast<vf4 *>(r10_smem + L.Vc);
    vf4 *sS = reinterpret_cast<vf4 *>(r10_smem + L.S), *pva = reinterpret_cast<vf4 *>(r10_smem + L.pva), *pac = reinterpret_cast<vf4 *>(r10_smem + L.pac);
    const float *sr = r10_smem + L.r, *sq = r10_smem + L.q, *sA = r10_smem + L.A;

    const int brow[2] = {wave, wave + nw};
    const bool okb[2] = {true, wave + nw < N};
    const float *Pg = P + g * NC * N * N + 4 * fl;
    float *Outg = Out + g * (size_t)N * N * 10 * C;
    const vf4 zero = vf4{0.f, 0.f, 0.f, 0.f};
    auto load_rows = [&](int a, vf4(&v)[2][SLOTS]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float *row = Pg + ((size_t)a * N + (okb[rb] ? brow[rb] : brow[0])) * NC;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const int c = i * ppw + cg;
                v[rb][i] = (okb[rb] && c < N) ? Vec<4>::ld_s<16>(row + (size_t)c * C) : zero;
            }
        }
    };
    vf4 sbc[2][SLOTS], vbacc[2] = {zero, zero}, vcacc = zero;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) sbc[rb][i] = zero;
    // (the 1024-thread build -- N > 24 -- has 128 registers per lane: no second row set, its sixteen waves hide the latency instead;
    //  with the prefetch it spilled 71 registers and ran no faster than the table kernels)
    constexpr bool PREFETCH = MAXT <= 768 || SLOTS <= 2;
    vf4 cur[2][SLOTS], nxt[PREFETCH ? 2 : 1][PREFETCH ? SLOTS : 1];
    if constexpr (PREFETCH) load_rows(0, cur);
    for (int a = 0; a < N; ++a) {
        if constexpr (PREFETCH) {
            if (a + 1 < N) load_rows(a + 1, nxt);   // (in flight over the sums and the barrier of a)
        } else {
            load_rows(a, cur);
        }
        vf4 sac[SLOTS], vap = zero;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) sac[i] = zero;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            vf4 sab = zero;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                sab += cur[rb][i];
                sbc[rb][i] += cur[rb][i];
                sac[i] += cur[rb][i];
            }
            sab = r10_butterfly(sab, lpc);
            vbacc[rb] += sab;
            vap += sab;
            if (cg == 0 && okb[rb]) Vec<4>::st_s<16>(Outg + (((size_t)a * N + brow[rb]) * 10 + 0) * C + 4 * fl, tot * sab);   // S_ab tot
        }
        const int buf = a & 1;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            if (c < N) pac[(((size_t)buf * nw + wave) * N + c) * lpc + fl] = sac[i];
        }
        if (cg == 0) pva[((size_t)buf * nw + wave) * lpc + fl] = vap;
        __syncthreads();
        if (tid < N * lpc) {   // S_ac[a, c] = sum_b: thread (c = tid / lpc, channel quad tid % lpc), the waves in order
            vf4 t = pac[((size_t)buf * nw) * N * lpc + tid];
            for (int w = 1; w < nw; ++w) t += pac[((size_t)buf * nw + w) * N * lpc + tid];
            vcacc += t;
            Vec<4>::st_s<16>(Outg + (((size_t)a * N + tid / lpc) * 10 + 1) * C + 4 * (tid % lpc), tot * t);
        }
        if (wave == nw - 1 && lane < lpc) {   // Va[a] = sum_b S_ab[a, b]
            vf4 t = pva[((size_t)buf * nw) * lpc + lane];
            for (int w = 1; w < nw; ++w) t += pva[((size_t)buf * nw + w) * lpc + lane];
            sVa[(size_t)a * lpc + lane] = t;
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < SLOTS; ++i) cur[rb][i] = nxt[rb][i];
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {   // S_bc tot, Vb
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            if (okb[rb] && c < N) Vec<4>::st_s<16>(Outg + (((size_t)brow[rb] * N + c) * 10 + 4) * C + 4 * fl, tot * sbc[rb][i]);
        }
        if (cg == 0 && okb[rb]) sVb[(size_t)brow[rb] * lpc + fl] = vbacc[rb];
    }
    if (tid < N * lpc) sVc[tid] = vcacc;
    __syncthreads();
    if (tid < lpc) {
        vf4 t = sVa[tid];
        for (int a = 1; a < N; ++a) t += sVa[(size_t)a * lpc + tid];
        sS[tid] = t;
    }
    __syncthreads();
    // the seven vector / scalar slices of every row (x, y)
    const int total = N * N * 7 * lpc;
    for (int idx = tid; idx < total; idx += blockDim.x) {
        const int q = idx % lpc, j = (idx / lpc) % 7, row = idx / (7 * lpc);
        const int x = row / N, y = row - x * N;
        const vf4 *vec = j < 2 ? sVa : j < 4 ? sVb : sVc;
        const float w = j == 6 ? sA[x * N + y] : (j & 1) ? sq[y] : sr[y];
        const vf4 v = (j == 6 ? sS[q] : vec[(size_t)x * lpc + q]) * w;
        const int k = j < 2 ? 2 + j : 3 + j;   // 2, 3 | 5, 6, 7, 8, 9
        Vec<4>::st_s<16>(Outg + ((size_t)row * 10 + k) * C + 4 * q, v);
    }
}

template <int SLOTS, bool ACC, int MAXT>
__global__ __launch_bounds__(MAXT) void r10_bwd_graph(const float *__restrict__ G, const float *__restrict__ A, float *__restrict__ dP, int N, int C,
                                                      int nw) {
    extern __shared__ __attribute__((aligned(16))) float r10_smem[];
    const int lpc = C >> 2, ppw = 64 / lpc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane / lpc, fl = lane % lpc;
    const size_t g = blockIdx.x;
    const size_t NC = (size_t)N * C;
    const R10Lds L = r10_lds(N, lpc, nw);
    const float tot = r10_adjacency(A + g * N * N, r10_smem, L, N);
    vf4 *sxa = reinterpret_cast<vf4 *>(r10_smem + L.Va), *sxb = reinterpret_cast<vf4 *>(r10_smem + L.Vb), *sxc = reinterpret_cast<vf4 *>(r10_smem + L.Vc);
    vf4 *sS = reinterpret_cast<vf4 *>(r10_smem + L.S), *ps = reinterpret_cast<vf4 *>(r10_smem + L.pva);
    const float *sr = r10_smem + L.r, *sq = r10_smem + L.q, *sA = r10_smem + L.A;
    const int brow[2] = {wave, wave + nw};
    const bool okb[2] = {true, wave + nw < N};
    const float *Gg = G + g * (size_t)N * N * 10 * C + 4 * fl;
    const vf4 zero = vf4{0.f, 0.f, 0.f, 0.f};
    {   // the vectors: rows x = the wave's two, sums over y
        vf4 sw = zero;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (!okb[rb]) continue;   // (wave-uniform)
            const int x = brow[rb];
            vf4 xa = zero, xb = zero, xc = zero, xs = zero;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const int y = i * ppw + cg;
                if (y < N) {
                    const float *row = Gg + ((size_t)x * N + y) * 10 * C;
                    const float ry = sr[y], qy = sq[y], axy = sA[x * N + y];
                    xa += Vec<4>::ld_s<16>(row + 2 * C) * ry + Vec<4>::ld_s<16>(row + 3 * C) * qy;
                    xb += Vec<4>::ld_s<16>(row + 5 * C) * ry + Vec<4>::ld_s<16>(row + 6 * C) * qy;
                    xc += Vec<4>::ld_s<16>(row + 7 * C) * ry + Vec<4>::ld_s<16>(row + 8 * C) * qy;
                    xs += Vec<4>::ld_s<16>(row + 9 * C) * axy;
                }
            }
            xa = r10_butterfly(xa, lpc), xb = r10_butterfly(xb, lpc), xc = r10_butterfly(xc, lpc), xs = r10_butterfly(xs, lpc);
            if (cg == 0) {
                sxa[(size_t)x * lpc + fl] = xa;
                sxb[(size_t)x * lpc + fl] = xb;
                sxc[(size_t)x * lpc + fl] = xc;
            }
            sw += xs;
        }
        if (cg == 0) ps[(size_t)wave * lpc + fl] = sw;
        __syncthreads();
        if (tid < lpc) {
            vf4 t = ps[tid];
            for (int w = 1; w < nw; ++w) t += ps[(size_t)w * lpc + tid];
            sS[tid] = t;
        }
        __syncthreads();
    }
    // dP streamed out: wave owns the rows b of every slab a; tot G4[b,c] + xb[b] + xc[c] + s stay in the lane
    vf4 base[2][SLOTS];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg, b = okb[rb] ? brow[rb] : brow[0];
            const int cc = c < N ? c : 0;
            base[rb][i] = tot * Vec<4>::ld_s<16>(Gg + (((size_t)b * N + cc) * 10 + 4) * C) + sxb[(size_t)b * lpc + fl] + sxc[(size_t)cc * lpc + fl] + sS[fl];
        }
    float *dPg = dP + g * NC * N * N + 4 * fl;
    auto load_a = [&](int a, vf4(&ga)[2], vf4(&g1)[SLOTS]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) ga[rb] = Vec<4>::ld_s<16>(Gg + (((size_t)a * N + (okb[rb] ? brow[rb] : brow[0])) * 10 + 0) * C);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int c = i * ppw + cg;
            g1[i] = Vec<4>::ld_s<16>(Gg + (((size_t)a * N + (c < N ? c : 0)) * 10 + 1) * C);
        }
    };
    vf4 ga[2], g1[SLOTS], gan[2], g1n[SLOTS];
    load_a(0, ga, g1);
    for (int a = 0; a < N; ++a) {
        if (a + 1 < N) load_a(a + 1, gan, g1n);
        const vf4 xa = sxa[(size_t)a * lpc + fl];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (!okb[rb]) continue;
            float *row = dPg + ((size_t)a * N + brow[rb]) * NC;
            const vf4 u = tot * ga[rb] + xa;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const int c = i * ppw + cg;
                if (c < N) {
                    vf4 v = base[rb][i] + u + tot * g1[i];
                    if (ACC) v += Vec<4>::ld(row + (size_t)c * C);
                    Vec<4>::st_s<16>(row + (size_t)c * C, v);
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) ga[rb] = gan[rb];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) g1[i] = g1n[i];
    }
}

// the per-graph kernels serve: C % 4 == 0 with C / 4 a power of two <= 32, N <= 32 with at most four position slots per lane, aligned
// buffers, a workgroup's LDS within the CU's, and batches that give most CUs a graph (GF_FAM10_GRAPH=0: the table kernels)
static bool r10_graph_ok(int N, int C, int batch, const void *p0, const void *p1, const void *p2) {
    const char *e = std::getenv("GF_FAM10_GRAPH");   // 0: the table kernels; 2: any batch size (tests)
    if (e && e[0] == '0') return false;
    if (C % 4 != 0 || C < 4 || N < 1 || N > 32 || (batch < 96 && !(e && e[0] == '2'))) return false;
    const int lpc = C / 4;
    if (lpc > 32 || (lpc & (lpc - 1)) != 0) return false;
    const int ppw = 64 / lpc, slots = (N + ppw - 1) / ppw, nw = (N + 1) / 2;
    if (slots > 4) return false;
    if ((size_t)r10_lds(N, lpc, nw).total * sizeof(float) > 160 * 1024) return false;
    return ((((uintptr_t)p0) | ((uintptr_t)p1) | ((uintptr_t)p2)) & 15) == 0;
}

// Output slot (0-based) of "case c" (1-based numbering of RisiContraction_50.h) in family K, or -1 when absent.
template <int K>
__host__ __device__ constexpr int slot(int c) {
    return (K == 50) ? c - 1 : (K == 10 && c <= 10) ? c - 1 : -1;
}

// per-graph adjacency statistics: adjs[g] = { r[N], q[N], dg[N], tot, tr }
__host__ __device__ inline size_t adjs_stride(int N) { return 3 * (size_t)N + 2; }

__global__ void fam_adj(const float *__restrict__ A, float *__restrict__ adjs, int N) {
    const size_t g = blockIdx.x;
    const float *Ag = A + g * N * N;
    float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N, *st = dg + N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float rs = 0.f, qs = 0.f;
        for (int j = 0; j < N; ++j) {
            rs += Ag[i * N + j];
            qs += Ag[j * N + i];
        }
        r[i] = rs;
        q[i] = qs;
        dg[i] = Ag[i * N + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, d = 0.f;
        for (int i = 0; i < N; ++i) {
            t += r[i];
            d += dg[i];
        }
        st[0] = t;
        st[1] = d;
    }
}

// pair tables, tab[g][12][N][N][C]:
//   0 S_ab  1 S_ac  2 S_bc                    (plain marginals)
//   3 ab.r  4 ab.q  5 ab.dg                   sum_c P[i,j,c] w[c]      (cases 11, 12, 41)
//   6 ac.r  7 ac.q  8 ac.dg                   sum_b P[i,b,j] w[b]      (cases 14, 15, 42)
//   9 bc.r 10 bc.q 11 bc.dg                   sum_a P[a,i,j] w[a]      (cases 23, 24, 45)
constexpr int kNTab = 12;

template <int K, int VW>
__global__ void fam_tables(const float *__restrict__ P, const float *__restrict__ adjs, float *__restrict__ tab, int N,
                           int C, size_t total) {
    using V = typename Vec<VW>::T;
    const size_t NNC = (size_t)N * N * C;
    const int CV = C / VW;
    GRID_STRIDE(idx, total) {
        const int f = (int)(idx % CV) * VW;
        size_t t = idx / CV;
        const int j = t % N;
        t /= N;
        const int i = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N;
        V acc[kNTab];
#pragma unroll
        for (int k = 0; k < kNTab; ++k) acc[k] = Vec<VW>::zero();
        for (int s = 0; s < N; ++s) {
            const float rs = r[s], qs = q[s], ds = dg[s];
            const V pab = Vec<VW>::ld(Pg + (((size_t)i * N + j) * N + s) * C + f);  // P[i][j][s]
            const V pac = Vec<VW>::ld(Pg + (((size_t)i * N + s) * N + j) * C + f);  // P[i][s][j]
            const V pbc = Vec<VW>::ld(Pg + (((size_t)s * N + i) * N + j) * C + f);  // P[s][i][j]
            acc[0] += pab;
            acc[1] += pac;
            acc[2] += pbc;
            if (K == 50) {
                acc[3] += pab * rs;
                acc[4] += pab * qs;
                acc[5] += pab * ds;
                acc[6] += pac * rs;
                acc[7] += pac * qs;
                acc[8] += pac * ds;
                acc[9] += pbc * rs;
                acc[10] += pbc * qs;
                acc[11] += pbc * ds;
            }
        }
        float *tg = tab + g * kNTab * NNC + ((size_t)i * N + j) * C + f;
#pragma unroll
        for (int k = 0; k < (K == 50 ? kNTab : 3); ++k) Vec<VW>::st(tg + k * NNC, acc[k]);
    }
}

// RisiContraction_50, C % 4 == 0, with fam50_forward_mfma<.., true>: the same pass over P, but
//   * only the three plain marginals go to the workspace (the adjacency products read their rows and columns); the nine weighted
//     tables ARE output slices (11, 12, 41 | 14, 15, 42 | 23, 24, 45) and the six slices S tot, S tr (1, 13 | 2, 16 | 5, 25) are one
//     multiply away: all fifteen are written straight into Out -- the workspace round trip of nine tables (0.34 GB per cfg5 step)
//     is gone and the output kernel has no table to load for its plain slices;
//   * UNR steps of the walk are loaded before any is used (3 UNR loads of 16 bytes in flight per thread), so the kernel keeps the
//     memory system busy at a LOW occupancy -- the launch asks for enough LDS to hold it to a few workgroups per CU: the three
//     roles of a graph's P (1.8 MB at cfg5) are read by consecutive workgroups of one XCD, and with about two graphs in flight
//     per XCD instead of fourteen the second and third reads are L2 hits (P was fetched 2.25 x).
// K = 10 (round 4): the same walk with the three plain marginals only -- S_ab, S_ac, S_bc into the workspace and S tot straight into
// slices 1, 2, 5 -- eight steps requested ahead, instead of fam_tables<10>'s one step at a time (0.185 -> 0.11 ms at the cfg5 shape).
template <int UNR, int K = 50>
__global__ __launch_bounds__(256) void fam50_tables_out(const float *__restrict__ P, const float *__restrict__ adjs,
                                                        float *__restrict__ tab, float *__restrict__ Out, int N, int C,
                                                        size_t total) {
    const size_t NNC = (size_t)N * N * C;
    const int CV = C / 4;
    GRID_STRIDE(idx, total) {
        const int f = (int)(idx % CV) * 4;
        size_t t = idx / CV;
        const int j = t % N;
        t /= N;
        const int i = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N + f;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N, *st = dg + N;
        vf4 acc[kNTab];
#pragma unroll
        for (int k = 0; k < kNTab; ++k) acc[k] = vf4{0.f, 0.f, 0.f, 0.f};
        const float *pab0 = Pg + ((size_t)i * N + j) * N * C;   // + s C
        const float *pac0 = Pg + (size_t)i * NNC + (size_t)j * C;   // + s N C
        const float *pbc0 = Pg + ((size_t)i * N + j) * C;       // + s N N C
        for (int s0 = 0; s0 < N; s0 += UNR) {
            vf4 pab[UNR], pac[UNR], pbc[UNR];
            float ok[UNR], rs[UNR], qs[UNR], ds[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool in = s0 + u < N;
                const int sc = in ? s0 + u : N - 1;   // (clamped re-read, weights 0)
                pab[u] = Vec<4>::ld(pab0 + (size_t)sc * C);
                pac[u] = Vec<4>::ld(pac0 + (size_t)sc * N * C);
                pbc[u] = Vec<4>::ld(pbc0 + (size_t)sc * NNC);
                ok[u] = in ? 1.f : 0.f;
                rs[u] = (K == 50 && in) ? r[sc] : 0.f;
                qs[u] = (K == 50 && in) ? q[sc] : 0.f;
                ds[u] = (K == 50 && in) ? dg[sc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                acc[0] += pab[u] * ok[u];  acc[1] += pac[u] * ok[u];  acc[2] += pbc[u] * ok[u];
                if constexpr (K == 50) {
                    acc[3] += pab[u] * rs[u];  acc[4] += pab[u] * qs[u];  acc[5] += pab[u] * ds[u];
                    acc[6] += pac[u] * rs[u];  acc[7] += pac[u] * qs[u];  acc[8] += pac[u] * ds[u];
                    acc[9] += pbc[u] * rs[u];  acc[10] += pbc[u] * qs[u]; acc[11] += pbc[u] * ds[u];
                }
            }
        }
        const float tot = st[0], tr = st[1];
        float *tg = tab + g * kNTab * NNC + ((size_t)i * N + j) * C + f;
        Vec<4>::st(tg + 0 * NNC, acc[0]);
        Vec<4>::st(tg + 1 * NNC, acc[1]);
        Vec<4>::st(tg + 2 * NNC, acc[2]);
        float *o = Out + (((size_t)g * N + i) * N + j) * (size_t)(K * C) + f;
#define OUTS(c, expr) Vec<4>::st_s<1>(o + (size_t)((c) - 1) * C, (expr))
        OUTS(1, acc[0] * tot);  OUTS(2, acc[1] * tot);  OUTS(5, acc[2] * tot);
        if constexpr (K == 50) {
            OUTS(11, acc[3]);       OUTS(12, acc[4]);       OUTS(13, acc[0] * tr);
            OUTS(14, acc[6]);       OUTS(15, acc[7]);       OUTS(16, acc[1] * tr);
            OUTS(23, acc[9]);       OUTS(24, acc[10]);      OUTS(25, acc[2] * tr);
            OUTS(41, acc[5]);       OUTS(42, acc[8]);       OUTS(45, acc[11]);
        }
        (void)tr;
#undef OUTS
    }
}

// single-index vectors and scalars, vec[g][6][N][C] then sc[g][5][C] (written by the threads with i == 0):
//   vec 0 s_a[i] = sum_b S_ab[i,b]   1 s_b[i] = sum_a S_ab[a,i]   2 s_c[i] = sum_b S_bc[b,i]
//       3 v_bb[i] = sum_b P[i,b,b]   4 v_aba[i] = sum_a P[a,i,a]  5 v_aac[i] = sum_a P[a,a,i]
//   sc  0 total   1 d1 = sum P[a,a,c]   2 d2 = sum P[a,b,a]   3 d3 = sum P[a,b,b]   4 d4 = sum P[a,a,a]
constexpr int kNVec = 6, kNSc = 5;

__global__ void fam_vectors(const float *__restrict__ P, const float *__restrict__ tab, float *__restrict__ vec,
                            float *__restrict__ sc, int N, int C, size_t total) {
    const size_t NNC = (size_t)N * N * C;
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        const int i = (idx / C) % N;
        const size_t g = idx / ((size_t)C * N);
        const float *Pg = P + g * NNC * N;
        const float *Sab = tab + g * kNTab * NNC, *Sbc = Sab + 2 * NNC;
        float sa = 0.f, sb = 0.f, s_c = 0.f, vbb = 0.f, vaba = 0.f, vaac = 0.f;
        for (int s = 0; s < N; ++s) {
            sa += Sab[((size_t)i * N + s) * C + f];
            sb += Sab[((size_t)s * N + i) * C + f];
            s_c += Sbc[((size_t)s * N + i) * C + f];
            vbb += Pg[(((size_t)i * N + s) * N + s) * C + f];
            vaba += Pg[(((size_t)s * N + i) * N + s) * C + f];
            vaac += Pg[(((size_t)s * N + s) * N + i) * C + f];
        }
        float *v = vec + g * kNVec * (size_t)N * C + (size_t)i * C + f;
        const size_t NC = (size_t)N * C;
        v[0 * NC] = sa;
        v[1 * NC] = sb;
        v[2 * NC] = s_c;
        v[3 * NC] = vbb;
        v[4 * NC] = vaba;
        v[5 * NC] = vaac;
    }
}

__global__ void fam_scalars(const float *__restrict__ P, const float *__restrict__ vec, float *__restrict__ sc, int N,
                            int C, size_t total) {
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        const size_t g = idx / C;
        const float *v = vec + g * kNVec * (size_t)N * C + f;
        const float *Pg = P + g * (size_t)N * N * N * C;
        const size_t NC = (size_t)N * C;
        float tot = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;
        for (int i = 0; i < N; ++i) {
            tot += v[0 * NC + (size_t)i * C];
            d1 += v[5 * NC + (size_t)i * C];
            d2 += v[4 * NC + (size_t)i * C];
            d3 += v[3 * NC + (size_t)i * C];
            d4 += Pg[(((size_t)i * N + i) * N + i) * C + f];
        }
        float *s = sc + g * kNSc * (size_t)C + f;
        s[0 * C] = tot;
        s[1 * C] = d1;
        s[2 * C] = d2;
        s[3 * C] = d3;
        s[4 * C] = d4;
    }
}

#define OUTC(c, expr)                                                  \
    do {                                                               \
        if (slot<K>(c) >= 0) Vec<VW>::st(o + slot<K>(c) * C, (expr)); \
    } while (0)

// Thread per (g, x, block of YB columns y, f): the nine operands of the N x N products that are indexed by (x, z) are
// loaded once per z and serve all YB columns (they do not depend on y); only two adjacency entries are per column.
// (with 16-byte channel vectors a thread keeps one column: 18 x 4 accumulators)
template <int K, int VW, bool PRODUCTS = true>
__global__ __launch_bounds__(256) void fam_forward(const float *__restrict__ P, const float *__restrict__ A,
                                                   const float *__restrict__ adjs, const float *__restrict__ tab,
                                                   const float *__restrict__ vec, const float *__restrict__ sc,
                                                   float *__restrict__ Out, int N, int C, int nyb, size_t total) {
    using V = typename Vec<VW>::T;
    constexpr int kYB = (VW == 4) ? 1 : 4;
    const size_t NNC = (size_t)N * N * C, NC = (size_t)N * C;
    const int CV = C / VW;
    GRID_STRIDE(idx, total) {
        const int f = (int)(idx % CV) * VW;
        size_t t = idx / CV;
        const int y0 = (int)(t % nyb) * kYB;
        t /= nyb;
        const int x = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *Ag = A + g * N * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
        const float tot = st[0], tr = st[1];
        const float *T = tab + g * kNTab * NNC;
        const float *v = vec + g * kNVec * NC + f;
        const float *s = sc + g * kNSc * (size_t)C + f;
#define TB(k, i, j) Vec<VW>::ld(T + (k)*NNC + ((size_t)(i) * N + (j)) * C + f)
#define PP(a, b, c) Vec<VW>::ld(Pg + (((size_t)(a) * N + (b)) * N + (c)) * C + f)
        const V va = Vec<VW>::ld(v + 0 * NC + (size_t)x * C), vb = Vec<VW>::ld(v + 1 * NC + (size_t)x * C),
                vc = Vec<VW>::ld(v + 2 * NC + (size_t)x * C);
        V vbb = Vec<VW>::zero(), vaba = Vec<VW>::zero(), vaac = Vec<VW>::zero();
        V s0 = Vec<VW>::ld(s + 0 * C), s1 = Vec<VW>::zero(), s2 = Vec<VW>::zero(), s3 = Vec<VW>::zero(), s4 = Vec<VW>::zero();
        if (K == 50) {
            vbb = Vec<VW>::ld(v + 3 * NC + (size_t)x * C);
            vaba = Vec<VW>::ld(v + 4 * NC + (size_t)x * C);
            vaac = Vec<VW>::ld(v + 5 * NC + (size_t)x * C);
            s1 = Vec<VW>::ld(s + 1 * C);
            s2 = Vec<VW>::ld(s + 2 * C);
            s3 = Vec<VW>::ld(s + 3 * C);
            s4 = Vec<VW>::ld(s + 4 * C);
        }
#pragma unroll
        for (int m = 0; m < kYB; ++m) {
            const int y = y0 + m;
            if (y >= N) break;
            float *o = Out + (((size_t)g * N + x) * N + y) * (size_t)(K * C) + f;
            const float ry = r[y], qy = q[y], axy = Ag[x * N + y];
            const V sab = TB(0, x, y), sac = TB(1, x, y), sbc = TB(2, x, y);
            // "1+1+1"
            OUTC(1, sab * tot);
            OUTC(2, sac * tot);
            OUTC(3, va * ry);
            OUTC(4, va * qy);
            OUTC(5, sbc * tot);
            OUTC(6, vb * ry);
            OUTC(7, vb * qy);
            OUTC(8, vc * ry);
            OUTC(9, vc * qy);
            OUTC(10, s0 * axy);
            if (K == 50) {
                // "1+2" that are plain table reads or outer products
                OUTC(11, TB(3, x, y));
                OUTC(12, TB(4, x, y));
                OUTC(13, sab * tr);
                OUTC(14, TB(6, x, y));
                OUTC(15, TB(7, x, y));
                OUTC(16, sac * tr);
                OUTC(17, vbb * ry);
                OUTC(20, vbb * qy);
                OUTC(23, TB(9, x, y));
                OUTC(24, TB(10, x, y));
                OUTC(25, sbc * tr);
                OUTC(26, vaba * ry);
                OUTC(29, vaba * qy);
                OUTC(32, vaac * ry);
                OUTC(35, vaac * qy);
                OUTC(38, s1 * axy);
                OUTC(39, s2 * axy);
                OUTC(40, s3 * axy);
                OUTC(41, TB(5, x, y));
                OUTC(42, TB(8, x, y));
                OUTC(45, TB(11, x, y));
                OUTC(50, s4 * axy);
            }
        }
        if (K == 50 && PRODUCTS) {
            // N x N products with A: one pass over the contracted index for the whole column block
            V mm[kYB][18];
#pragma unroll
            for (int m = 0; m < kYB; ++m)
#pragma unroll
                for (int k = 0; k < 18; ++k) mm[m][k] = Vec<VW>::zero();
            for (int z = 0; z < N; ++z) {
                const V sab_xz = TB(0, x, z), sab_zx = TB(0, z, x);
                const V sac_xz = TB(1, x, z), sac_zx = TB(1, z, x);
                const V sbc_xz = TB(2, x, z), sbc_zx = TB(2, z, x);
                const V pxzz = PP(x, z, z), pzxz = PP(z, x, z), pzzx = PP(z, z, x);
#pragma unroll
                for (int m = 0; m < kYB; ++m) {
                    const int y = (y0 + m < N) ? y0 + m : N - 1;
                    const float ayz = Ag[y * N + z], azy = Ag[z * N + y];
                    mm[m][0] += sab_xz * ayz;   // 18 (a,d) tie(b,e): sum_b S_ab[a,b] A[d,b]
                    mm[m][1] += sac_xz * ayz;   // 19 (a,d) tie(c,e)
                    mm[m][2] += sab_xz * azy;   // 21 (a,e) tie(b,d): sum_b S_ab[a,b] A[b,e]
                    mm[m][3] += sac_xz * azy;   // 22 (a,e) tie(c,d)
                    mm[m][4] += sab_zx * ayz;   // 27 (b,d) tie(a,e): sum_a S_ab[a,b] A[d,a]
                    mm[m][5] += sbc_xz * ayz;   // 28 (b,d) tie(c,e)
                    mm[m][6] += sab_zx * azy;   // 30 (b,e) tie(a,d)
                    mm[m][7] += sbc_xz * azy;   // 31 (b,e) tie(c,d)
                    mm[m][8] += sac_zx * ayz;   // 33 (c,d) tie(a,e): sum_a S_ac[a,c] A[d,a]
                    mm[m][9] += sbc_zx * ayz;   // 34 (c,d) tie(b,e)
                    mm[m][10] += sac_zx * azy;  // 36 (c,e) tie(a,d)
                    mm[m][11] += sbc_zx * azy;  // 37 (c,e) tie(b,d)
                    mm[m][12] += pxzz * ayz;    // 43 (a,d) b=c=e
                    mm[m][13] += pxzz * azy;    // 44 (a,e) b=c=d
                    mm[m][14] += pzxz * ayz;    // 46 (b,d) a=c=e
                    mm[m][15] += pzxz * azy;    // 47 (b,e) a=c=d
                    mm[m][16] += pzzx * ayz;    // 48 (c,d) a=b=e
                    mm[m][17] += pzzx * azy;    // 49 (c,e) a=b=d
                }
            }
            constexpr int cases[18] = {18, 19, 21, 22, 27, 28, 30, 31, 33, 34, 36, 37, 43, 44, 46, 47, 48, 49};
#pragma unroll
            for (int m = 0; m < kYB; ++m) {
                if (y0 + m >= N) break;
                float *o = Out + (((size_t)g * N + x) * N + y0 + m) * (size_t)(K * C) + f;
#pragma unroll
                for (int k = 0; k < 18; ++k) Vec<VW>::st(o + (size_t)(cases[k] - 1) * C, mm[m][k]);
            }
        }
#undef TB
#undef PP
    }
}
#undef OUTC

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_50 forward outputs on the matrix pipe (C % 32 == 0, N <= 32): ONE kernel writes all fifty slices of a row
// (the table kernels write them in one launch of a thread per (x, y, channel quad)).  One WAVE owns (graph g, row x, window of 32
// channels).  The eighteen adjacency products  Out_k[x, y, f] = sum_z op[z, f] M[z, y]  (M = A[y, z] or A[z, y]) are
// v_mfma_f32_32x32x2_f32 with the nine operands -- rows / columns x of S_ab, S_ac, S_bc and the three diagonals of P -- loaded
// straight from global memory in the B-operand layout (z = 2 step + lane / 32, f = lane % 32: two coalesced 128-byte rows per
// load, each operand read once and used by two products) and the adjacency in registers in the A-operand layout.  The result
// leaves the pipe as (row y = 8 (v / 4) + 4 (lane / 32) + v % 4, column f = lane % 32), the layout the thirty-two plain slices
// are evaluated in as well: every store is two 128-byte rows of Out.
// ---------------------------------------------------------------------------------------------------------------
typedef float f16acc __attribute__((ext_vector_type(16)));

// Buffer addressing for the two matrix-pipe kernels: wave-uniform descriptor + per-lane byte offset + scalar byte offset.  A lane
// that has no row / no z gets the offset kOob: its loads return 0 and its stores are dropped by the bounds check, so the kernels
// are free of branches and of 64-bit address arithmetic.
constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bst_out(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {   // slices of Out (aux bit 1 = nt)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, (GF_NT_FAM & 2) ? 2 : 0);
}

template <int NS, bool SLIM>  // N <= 2 NS; SLIM: fam50_tables_out has written the fifteen slices that are tables already
__global__ __launch_bounds__(256, 2) void fam50_forward_mfma(const float *__restrict__ P, const float *__restrict__ A,
                                                             const float *__restrict__ adjs, const float *__restrict__ tab,
                                                             const float *__restrict__ vec, const float *__restrict__ sc,
                                                             float *__restrict__ Out, int N, int C, unsigned nwaves) {
    constexpr int K = 50;
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    const unsigned wid = (unsigned)xcd_block() * 4 + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const unsigned nwin = (unsigned)C >> 5;
    const int win = (int)(wid % nwin), x = (int)((wid / nwin) % (unsigned)N);
    const size_t g = wid / nwin / (unsigned)N;
    const int f = win * 32 + m;
    const unsigned uN = (unsigned)N, uC = (unsigned)C, NC = uN * uC, NNC = uN * NC;
    const float *Ag = A + g * N * N;
    const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
    const __amdgpu_buffer_rsrc_t rT = dev::make_rsrc(tab + g * kNTab * (size_t)NNC, (size_t)kNTab * NNC * 4);
    const __amdgpu_buffer_rsrc_t rP = dev::make_rsrc(P + g * (size_t)NNC * N, (size_t)NNC * N * 4);
    const __amdgpu_buffer_rsrc_t rO = dev::make_rsrc(Out + (g * N + x) * (size_t)N * K * C, (size_t)N * K * C * 4);
    auto row_of = [&](int v) { return 8 * (v >> 2) + 4 * hi + (v & 3); };
    unsigned oY[16];   // byte offset of (column y, channel f) in the row of Out; slice cs adds (cs - 1) C floats
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const unsigned y = (unsigned)row_of(v);
        oY[v] = y < uN ? (y * (unsigned)K * uC + (unsigned)f) * 4u : kOob;
    }
    {   // the thirty-two plain slices
        const float *vv = vec + g * kNVec * (size_t)NC + (size_t)x * C + f;
        const float *ss = sc + g * kNSc * (size_t)C + f;
        const float va = vv[0 * NC], vb = vv[1 * NC], vc = vv[2 * NC], vbb = vv[3 * NC], vaba = vv[4 * NC], vaac = vv[5 * NC];
        const float s0 = ss[0 * C], s1 = ss[1 * C], s2 = ss[2 * C], s3 = ss[3 * C], s4 = ss[4 * C];
        const float tot = st[0], tr = st[1];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned y = (unsigned)row_of(v);
            const unsigned yc = y < uN ? y : 0u;
            const unsigned tY = y < uN ? (((unsigned)x * uN + y) * uC + (unsigned)f) * 4u : kOob;
            (void)tY;
            const float ry = r[yc], qy = q[yc], axy = Ag[x * N + (int)yc];
#define OUTS(c, expr) bst_out(rO, oY[v], (unsigned)((c) - 1) * uC * 4u, (expr))
            if (!SLIM) {
                float t[kNTab];
#pragma unroll
                for (int k = 0; k < kNTab; ++k) t[k] = bld(rT, tY, (unsigned)k * NNC * 4u);
                OUTS(1, t[0] * tot);  OUTS(2, t[1] * tot);  OUTS(5, t[2] * tot);
                OUTS(11, t[3]);       OUTS(12, t[4]);       OUTS(13, t[0] * tr); OUTS(14, t[6]);    OUTS(15, t[7]);
                OUTS(16, t[1] * tr);  OUTS(23, t[9]);       OUTS(24, t[10]);     OUTS(25, t[2] * tr);
                OUTS(41, t[5]);       OUTS(42, t[8]);       OUTS(45, t[11]);
            }
            OUTS(3, va * ry);     OUTS(4, va * qy);     OUTS(6, vb * ry);     OUTS(7, vb * qy);     OUTS(8, vc * ry);
            OUTS(9, vc * qy);     OUTS(10, s0 * axy);   OUTS(17, vbb * ry);   OUTS(20, vbb * qy);   OUTS(26, vaba * ry);
            OUTS(29, vaba * qy);  OUTS(32, vaac * ry);  OUTS(35, vaac * qy);  OUTS(38, s1 * axy);   OUTS(39, s2 * axy);
            OUTS(40, s3 * axy);   OUTS(50, s4 * axy);
#undef OUTS
            if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four rows of loads in flight, not sixteen)
        }
    }
    float ma[NS], mt[NS];
    bool zok[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int z = 2 * t + hi;
        zok[t] = z < N;
        const bool v = zok[t] && m < N;
        const int zc = zok[t] ? z : 0, mc = m < N ? m : 0;
        const float a0 = Ag[zc * N + mc], a1 = Ag[mc * N + zc];
        ma[t] = v ? a0 : 0.f;   // A[z, y]
        mt[t] = v ? a1 : 0.f;   // A[y, z]
    }
    // operand (z, f) at  base + z zstr  floats of descriptor rs, in the B layout; slices c_yz = sum_z op A[y, z], c_zy = sum_z op A[z, y]
    auto product = [&](__amdgpu_buffer_rsrc_t rs, unsigned base, unsigned zstr, int c_yz, int c_zy) {
        const unsigned lo = ((unsigned)hi * zstr + (unsigned)f) * 4u;
        float b[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) b[t] = bld(rs, zok[t] ? lo : kOob, (base + 2u * t * zstr) * 4u);
        f16acc d0, d1;
#pragma unroll
        for (int v = 0; v < 16; ++v) d0[v] = d1[v] = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(mt[t], b[t], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[t], b[t], d1, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            bst_out(rO, oY[v], (unsigned)(c_yz - 1) * uC * 4u, d0[v]);
            bst_out(rO, oY[v], (unsigned)(c_zy - 1) * uC * 4u, d1[v]);
        }
    };
    const unsigned xr = (unsigned)x * NC, xc = (unsigned)x * uC;
    product(rT, 0 * NNC + xr, uC, 18, 21);          // S_ab[x, z]
    product(rT, 1 * NNC + xr, uC, 19, 22);          // S_ac[x, z]
    product(rT, 0 * NNC + xc, NC, 27, 30);          // S_ab[z, x]
    product(rT, 2 * NNC + xr, uC, 28, 31);          // S_bc[x, z]
    product(rT, 1 * NNC + xc, NC, 33, 36);          // S_ac[z, x]
    product(rT, 2 * NNC + xc, NC, 34, 37);          // S_bc[z, x]
    product(rP, (unsigned)x * NNC, NC + uC, 43, 44);   // P[x, z, z]
    product(rP, xr, NNC + uC, 46, 47);              // P[z, x, z]
    product(rP, xc, NNC + NC, 48, 49);              // P[z, z, x]
}

// backward scalars, bsc[g][5][C]: u10, u38, u39, u40, u50 = sum_{d,e} G_c[d,e] A[d,e].
// Workgroup per (graph, slice): 256 threads = row groups x channel lanes; a group walks (d,e) = grp, grp + ngrp, ...,
// the groups are folded through LDS in a fixed order (deterministic).
template <int K>
__global__ __launch_bounds__(256) void fam_bwd_scalars(const float *__restrict__ G, const float *__restrict__ A,
                                                       float *__restrict__ bsc, int N, int C) {
    __shared__ float red[256];
    const size_t g = blockIdx.x / 5;
    const int j = (int)(blockIdx.x % 5);
    const int cs = (j == 0) ? 10 : (j == 1) ? 38 : (j == 2) ? 39 : (j == 3) ? 40 : 50;
    const int sl = (K == 50) ? cs - 1 : (cs <= 10 ? cs - 1 : -1);
    const int nl = C < 256 ? C : 256, ngrp = 256 / nl;
    const int fl = threadIdx.x % nl, grp = threadIdx.x / nl;
    const float *Ag = A + g * N * N;
    for (int f0 = 0; f0 < C; f0 += nl) {
        const int f = f0 + fl;
        float sum = 0.f;
        if (sl >= 0 && grp < ngrp && f < C) {
            const float *Gg = G + g * (size_t)N * N * K * C + (size_t)sl * C + f;
            int de = grp;
            for (; de + 3 * ngrp < N * N; de += 4 * ngrp) {  // four independent loads in flight
                const float x0 = Gg[(size_t)de * K * C], x1 = Gg[(size_t)(de + ngrp) * K * C];
                const float x2 = Gg[(size_t)(de + 2 * ngrp) * K * C], x3 = Gg[(size_t)(de + 3 * ngrp) * K * C];
                sum += x0 * Ag[de];
                sum += x1 * Ag[de + ngrp];
                sum += x2 * Ag[de + 2 * ngrp];
                sum += x3 * Ag[de + 3 * ngrp];
            }
            for (; de < N * N; de += ngrp) sum += Gg[(size_t)de * K * C] * Ag[de];
        }
        red[threadIdx.x] = sum;
        __syncthreads();
        if (grp == 0 && f < C) {
            float t = red[fl];
            for (int k = 1; k < ngrp; ++k) t += red[k * nl + fl];
            bsc[g * 5 * (size_t)C + (size_t)j * C + f] = t;
        }
        __syncthreads();
    }
}

// backward pair tables, btab[g][6][N][N][C]:
//   0 X_ab[a,b]  1 X_ac[a,c]  2 X_bc[b,c]  3 Z_bc[a,b] (applies at b==c)  4 Z_ac[b,a] (a==c)  5 Z_ab[c,a] (a==b)
// Thread per (g, i, block of JB columns j, f): the twenty G slices indexed by (i,z) are loaded once per z and serve
// all JB columns; only the ten slices indexed by (j,z) and four adjacency entries are per column.
constexpr int kNBTab = 6;

template <int K, int VW, int IB>
__global__ __launch_bounds__(256) void fam_bwd_tables(const float *__restrict__ G, const float *__restrict__ A,
                                                      const float *__restrict__ adjs, const float *__restrict__ bsc,
                                                      float *__restrict__ btab, int N, int C, int njb, size_t total) {
    // IB rows i x kJB columns j per thread (round 3: IB = 2 -- per z a thread loads the twenty (i,z) slices of each row and the ten
    // (j,z) slices of each column, (20 IB + 10 kJB) loads for IB kJB outputs: 13.3 per output at 1 x 6, 8.3 at 2 x 6; the kernel
    // is bound by those loads through L1)
    using V = typename Vec<VW>::T;
    constexpr int kJB = (VW == 4) ? 2 : 6;
    const size_t NNC = (size_t)N * N * C;
    const int CV = C / VW, nib = (N + IB - 1) / IB;
    GRID_STRIDE(idx, total) {
        const int f = (int)(idx % CV) * VW;
        size_t t = idx / CV;
        const int j0 = (int)(t % njb) * kJB;
        t /= njb;
        const int i0 = (int)(t % nib) * IB;
        const size_t g = t / nib;
        const float *Gg = G + g * (size_t)N * N * K * C + f;
        const float *Ag = A + g * N * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
        const float tot = st[0], tr = st[1];
        const float *u = bsc + g * 5 * (size_t)C + f;
#define GC(c, x, y) (slot<K>(c) >= 0 ? Vec<VW>::ld(Gg + (((size_t)(x) * N + (y)) * K + slot<K>(c)) * C) : Vec<VW>::zero())
        V xab[IB][kJB], xac[IB][kJB], xbc[IB][kJB], zbc[IB][kJB], zac[IB][kJB], zab[IB][kJB];
        const V u10 = Vec<VW>::ld(u + 0 * C);
        V u38 = Vec<VW>::zero(), u39 = Vec<VW>::zero(), u40 = Vec<VW>::zero();
        if (K == 50) {
            u38 = Vec<VW>::ld(u + 1 * C);
            u39 = Vec<VW>::ld(u + 2 * C);
            u40 = Vec<VW>::ld(u + 3 * C);
        }
        int jj[kJB], ii[IB];
#pragma unroll
        for (int n = 0; n < IB; ++n) ii[n] = (i0 + n < N) ? i0 + n : N - 1;  // clamped duplicates, never stored
#pragma unroll
        for (int m = 0; m < kJB; ++m) {
            jj[m] = (j0 + m < N) ? j0 + m : N - 1;
            const int j = jj[m];
#pragma unroll
            for (int n = 0; n < IB; ++n) {
                const int i = ii[n];
                xab[n][m] = tot * GC(1, i, j) + u10;
                xac[n][m] = tot * GC(2, i, j);
                xbc[n][m] = tot * GC(5, i, j);
                zbc[n][m] = zac[n][m] = zab[n][m] = Vec<VW>::zero();
                if (K == 50) {
                    xab[n][m] += tr * GC(13, i, j);
                    xac[n][m] += tr * GC(16, i, j);
                    xbc[n][m] += tr * GC(25, i, j);
                    zbc[n][m] = u40;
                    zac[n][m] = u39;
                    zab[n][m] = u38;
                }
            }
        }
        for (int z = 0; z < N; ++z) {
            const float rz = r[z], qz = q[z];
            // (i,z)-indexed slices: shared by every column of the block
            const V zz = Vec<VW>::zero();
            V i34[IB], g18[IB], g21[IB], g19[IB], g22[IB], g28[IB], g31[IB], g43[IB], g44[IB], g46[IB], g47[IB], g48[IB], g49[IB], z17[IB],
                z26[IB], z32[IB];
            float azi[IB], aiz[IB];
#pragma unroll
            for (int n = 0; n < IB; ++n) {
                const int i = ii[n];
                i34[n] = GC(3, i, z) * rz + GC(4, i, z) * qz;
                g18[n] = g21[n] = g19[n] = g22[n] = g28[n] = g31[n] = g43[n] = g44[n] = g46[n] = g47[n] = g48[n] = g49[n] = z17[n] = z26[n] =
                    z32[n] = zz;
                azi[n] = aiz[n] = 0.f;
                if (K == 50) {
                    g18[n] = GC(18, i, z); g21[n] = GC(21, i, z); g19[n] = GC(19, i, z); g22[n] = GC(22, i, z);
                    g28[n] = GC(28, i, z); g31[n] = GC(31, i, z); g43[n] = GC(43, i, z); g44[n] = GC(44, i, z);
                    g46[n] = GC(46, i, z); g47[n] = GC(47, i, z); g48[n] = GC(48, i, z); g49[n] = GC(49, i, z);
                    z17[n] = GC(17, i, z) * rz + GC(20, i, z) * qz;
                    z26[n] = GC(26, i, z) * rz + GC(29, i, z) * qz;
                    z32[n] = GC(32, i, z) * rz + GC(35, i, z) * qz;
                    azi[n] = Ag[z * N + i];
                    aiz[n] = Ag[i * N + z];
                }
            }
#pragma unroll
            for (int m = 0; m < kJB; ++m) {
                const int j = jj[m];
                // (j,z)-indexed slices: shared by every row of the block
                // outer-product cases: X_ab[a=i,b=j] takes U3[a]+U4[a]+U6[b]+U7[b]; X_ac[a=i,c=j] takes U8[c]+U9[c]
                const V j67 = GC(6, j, z) * rz + GC(7, j, z) * qz, j89 = GC(8, j, z) * rz + GC(9, j, z) * qz;
                V g27 = zz, g30 = zz, g33 = zz, g36 = zz, g34 = zz, g37 = zz;
                float azj = 0.f, ajz = 0.f;
                if (K == 50) {
                    g27 = GC(27, j, z); g30 = GC(30, j, z); g33 = GC(33, j, z); g36 = GC(36, j, z); g34 = GC(34, j, z); g37 = GC(37, j, z);
                    azj = Ag[z * N + j], ajz = Ag[j * N + z];
                }
#pragma unroll
                for (int n = 0; n < IB; ++n) {
                    xab[n][m] += i34[n] + j67;
                    xac[n][m] += j89;
                    if (K == 50) {
                        xab[n][m] += g18[n] * azj + g21[n] * ajz + g27 * azi[n] + g30 * aiz[n];   // X_ab[a=i, b=j]
                        xac[n][m] += g19[n] * azj + g22[n] * ajz + g33 * azi[n] + g36 * aiz[n];   // X_ac[a=i, c=j]
                        xbc[n][m] += g28[n] * azj + g31[n] * ajz + g34 * azi[n] + g37 * aiz[n];   // X_bc[b=i, c=j]
                        zbc[n][m] += z17[n] + g43[n] * azj + g44[n] * ajz;                        // Z_bc[a=i, b=j]  (b == c)
                        zac[n][m] += z26[n] + g46[n] * azj + g47[n] * ajz;                        // Z_ac[b=i, a=j]  (a == c)
                        zab[n][m] += z32[n] + g48[n] * azj + g49[n] * ajz;                        // Z_ab[c=i, a=j]  (a == b)
                    }
                }
            }
        }
#undef GC
#pragma unroll
        for (int n = 0; n < IB; ++n) {
#pragma unroll
            for (int m = 0; m < kJB; ++m) {
                if (j0 + m >= N || i0 + n >= N) continue;
                float *bt = btab + g * kNBTab * NNC + ((size_t)(i0 + n) * N + j0 + m) * C + f;
                Vec<VW>::st(bt + 0 * NNC, xab[n][m]);
                Vec<VW>::st(bt + 1 * NNC, xac[n][m]);
                Vec<VW>::st(bt + 2 * NNC, xbc[n][m]);
                if (K == 50) {
                    Vec<VW>::st(bt + 3 * NNC, zbc[n][m]);
                    Vec<VW>::st(bt + 4 * NNC, zac[n][m]);
                    Vec<VW>::st(bt + 5 * NNC, zab[n][m]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// RisiContraction_50 backward tables on the matrix pipe (C % 32 == 0, N <= 32): every G row is read ONCE, straight from global
// memory into the operand registers of v_mfma_f32_32x32x2_f32 -- no LDS, no re-reads through L1.
//
// All six tables are sums of  out[t, f] += sum_z G_s[x, z, f] M[z, t]  with M one of A[z, t], A[t, z] (per graph) or a weight
// vector r[z], q[z] that does not depend on t.  One WAVE owns (graph g, row x, window of 32 channels).  The G operand of a step
// is the 2 x 32 block (z = 2 step + lane / 32, f = lane % 32) -- exactly the B-operand layout of the instruction, and two fully
// coalesced 128-byte rows of the slice per load.  The adjacency operand (A-operand layout: t = lane % 32, z likewise) is loaded
// once per wave and stays in registers for all eighteen products.  The weighted sums (r, q) run on the VALU and are added to
// every row of the accumulator.  D leaves the pipe as (row t = 8 (v / 4) + 4 (lane / 32) + v % 4, column f = lane % 32): stores
// are 128-byte rows again.  Twelve MFMAs per product at N = 24 (216 per wave): a twentieth of the kernel's memory time.
//   ROLE 0 (column role, first launch): the slices indexed (j = x, z) give COLUMN x of X_ab, X_ac, X_bc: stored at [t][x].
//   ROLE 1 (row role, second launch): the slices indexed (i = x, z) give ROW x of all six tables; X_* start from what the
//   first launch left at [x][t], the direct (x, t)-indexed slices and the per-graph scalars.
// fam_bwd_tables (thread per (i, columns j, f)) re-read the (i, z) slices once per column block and the (j, z) slices once per
// row pair through L1: 0.35 ms at cfg5, the longest kernel of that step.
// ---------------------------------------------------------------------------------------------------------------

#ifndef GF_FAM_BWD_OCC
#define GF_FAM_BWD_OCC 3   // (4: 128 registers, six spills, 0.118 + 0.151 ms against 0.109 + 0.136)
#endif
#ifndef GF_FAM_BWD_SCHED_BARRIER
#define GF_FAM_BWD_SCHED_BARRIER 1   // (0: the tables of a launch interleave -- measured 0.167 ms against 0.13 ms for the row launch)
#endif
template <int NS, int ROLE>  // N <= 2 NS
__global__ __launch_bounds__(256, GF_FAM_BWD_OCC) void fam50_bwd_tables_mfma(const float *__restrict__ G, const float *__restrict__ A,
                                                                const float *__restrict__ adjs, const float *__restrict__ bsc,
                                                                float *__restrict__ btab, float *__restrict__ bpart, int N, int C,
                                                                unsigned nwaves) {
    constexpr int K = 50;
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    const unsigned wid = (unsigned)xcd_block() * 4 + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const unsigned nwin = (unsigned)C >> 5;
    const int win = (int)(wid % nwin), x = (int)((wid / nwin) % (unsigned)N);
    const size_t g = wid / nwin / (unsigned)N;
    const int f = win * 32 + m;
    const unsigned uN = (unsigned)N, uC = (unsigned)C, NC = uN * uC, NNC = uN * NC, zs = (unsigned)K * uC;
    const float *Ag = A + g * N * N;
    const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
    // row x of G: slice cs of (x, z) at  z zs + (cs - 1) C  floats
    const __amdgpu_buffer_rsrc_t rG = dev::make_rsrc(G + (g * N + x) * (size_t)N * zs, (size_t)N * zs * 4);
    const __amdgpu_buffer_rsrc_t rB = dev::make_rsrc(btab + g * kNBTab * (size_t)NNC, (size_t)kNBTab * NNC * 4);
    float ma[NS], mt[NS], rz[NS], qz[NS];
    unsigned lo[NS];   // this lane's (z, f) of step t inside a G row, or kOob
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int z = 2 * t + hi;
        const bool zv = z < N, v = zv && m < N;
        const int zc = zv ? z : 0, mc = m < N ? m : 0;
        const float a0 = Ag[zc * N + mc], a1 = Ag[mc * N + zc], r0 = r[zc], q0 = q[zc];
        ma[t] = v ? a0 : 0.f;   // A[z, t]
        mt[t] = v ? a1 : 0.f;   // A[t, z]
        rz[t] = zv ? r0 : 0.f;
        qz[t] = zv ? q0 : 0.f;
        lo[t] = zv ? ((unsigned)hi * zs + (unsigned)f) * 4u : kOob;
    }
    auto ld = [&](int cs, int t) -> float { return bld(rG, lo[t], (2u * t * zs + (unsigned)(cs - 1) * uC) * 4u); };
    // acc += G_ca . A[z, t] + G_ct . A[t, z]
    auto prod = [&](f16acc &acc, int ca, int ct) {
        float ga[NS], gt[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) ga[t] = ld(ca, t), gt[t] = ld(ct, t);
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[t], ga[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mt[t], gt[t], acc, 0, 0, 0);
        }
    };
    // sum_z G_cr[x, z] r[z] + G_cq[x, z] q[z]   (the same value in both halves of the wave)
    auto wsum = [&](int cr, int cq) -> float {
        float w = 0.f;
#pragma unroll
        for (int t = 0; t < NS; ++t) w += ld(cr, t) * rz[t] + ld(cq, t) * qz[t];
        return dev::xor32_sum(w);
    };
    auto row_of = [&](int v) { return 8 * (v >> 2) + 4 * hi + (v & 3); };
    if (ROLE == 0) {
        unsigned cY[16];   // (t, x, f) of a table
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = (unsigned)row_of(v);
            cY[v] = t < uN ? ((t * uN + (unsigned)x) * uC + (unsigned)f) * 4u : kOob;
        }
        auto col_table = [&](int tbl, float w, int ca, int ct) {
            f16acc acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = w;
            prod(acc, ca, ct);
#pragma unroll
            for (int v = 0; v < 16; ++v) bst(rB, cY[v], (unsigned)tbl * NNC * 4u, acc[v]);
#if GF_FAM_BWD_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        col_table(0, wsum(6, 7), 27, 30);
        col_table(1, wsum(8, 9), 33, 36);
        col_table(2, 0.f, 34, 37);
        // (measured: the three diagonal tables in THIS launch, their scalars added by the combination kernel -- 27 slices here, 14 in
        //  the row launch -- 0.165 + 0.098 ms against 0.108 + 0.127; this launch is the first reader after the forward's 0.9 GB of
        //  writes and runs 0.03 ms slower than its own repeat)
        // row x of the five per-graph scalars u_c = sum_{d,e} G_c[d, e] A[d, e] (cases 10, 38, 39, 40, 50): bpart[g][x][5][C], folded
        // over x by fam50_bwd_scalars_fold before the row launch (fam_bwd_scalars read these slices in a launch of its own)
        float ax[NS];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int z = 2 * t + hi;
            const float a = Ag[x * N + (z < N ? z : 0)];
            ax[t] = z < N ? a : 0.f;
        }
        constexpr int ucase[5] = {10, 38, 39, 40, 50};
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float w = 0.f;
#pragma unroll
            for (int t = 0; t < NS; ++t) w += ld(ucase[j], t) * ax[t];
            w = dev::xor32_sum(w);
            if (hi == 0) bpart[((g * N + x) * 5 + j) * (size_t)C + f] = w;
        }
    } else {
        const float tot = st[0], tr = st[1];
        const float *u = bsc + g * 5 * (size_t)C + f;
        unsigned rY[16], gY[16];   // (x, t, f) of a table; (t, f) of a G row
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const unsigned t = (unsigned)row_of(v);
            rY[v] = t < uN ? (((unsigned)x * uN + t) * uC + (unsigned)f) * 4u : kOob;
            gY[v] = t < uN ? (t * zs + (unsigned)f) * 4u : kOob;
        }
        // X tables: tot G_c1[x, t] + tr G_c2[x, t] + what the column launch left + u + weighted sums + the two products
        auto xtable = [&](int tbl, int c1, int c2, float add, int ca, int ct) {
            f16acc acc;
#pragma unroll
            for (int v = 0; v < 16; ++v)
                acc[v] = tot * bld(rG, gY[v], (unsigned)(c1 - 1) * uC * 4u) + tr * bld(rG, gY[v], (unsigned)(c2 - 1) * uC * 4u) +
                         bld(rB, rY[v], (unsigned)tbl * NNC * 4u) + add;
            prod(acc, ca, ct);
#pragma unroll
            for (int v = 0; v < 16; ++v) bst(rB, rY[v], (unsigned)tbl * NNC * 4u, acc[v]);
#if GF_FAM_BWD_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        auto ztable = [&](int tbl, float add, int ca, int ct) {
            f16acc acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = add;
            prod(acc, ca, ct);
#pragma unroll
            for (int v = 0; v < 16; ++v) bst(rB, rY[v], (unsigned)tbl * NNC * 4u, acc[v]);
#if GF_FAM_BWD_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        xtable(0, 1, 13, u[0] + wsum(3, 4), 18, 21);
        xtable(1, 2, 16, 0.f, 19, 22);
        xtable(2, 5, 25, 0.f, 28, 31);
        ztable(3, u[3 * C] + wsum(17, 20), 43, 44);
        ztable(4, u[2 * C] + wsum(26, 29), 46, 47);
        ztable(5, u[1 * C] + wsum(32, 35), 48, 49);
    }
}

template <int K>
__global__ void fam_backward(const float *__restrict__ G, const float *__restrict__ A, const float *__restrict__ adjs,
                             const float *__restrict__ bsc, const float *__restrict__ btab, float *__restrict__ dP, int N,
                             int C, size_t total, int accumulate) {
    const size_t NNC = (size_t)N * N * C;
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int c = t % N;
        t /= N;
        const int b = t % N;
        t /= N;
        const int a = t % N;
        const size_t g = t / N;
        const float *bt = btab + g * kNBTab * NNC + f;
#define BT(k, i, j) bt[(k)*NNC + ((size_t)(i) * N + (j)) * C]
        float v = BT(0, a, b) + BT(1, a, c) + BT(2, b, c);
        if (K == 50) {
            const float *Gg = G + g * (size_t)N * N * K * C + f;
            const float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N;
#define GC(cs, x, y) Gg[(((size_t)(x) * N + (y)) * K + slot<K>(cs)) * C]
            v += GC(11, a, b) * r[c] + GC(12, a, b) * q[c] + GC(41, a, b) * dg[c];
            v += GC(14, a, c) * r[b] + GC(15, a, c) * q[b] + GC(42, a, c) * dg[b];
            v += GC(23, b, c) * r[a] + GC(24, b, c) * q[a] + GC(45, b, c) * dg[a];
#undef GC
            if (b == c) v += BT(3, a, b);
            if (a == c) v += BT(4, b, a);
            if (a == b) v += BT(5, c, a);
            if (a == b && b == c) v += bsc[g * 5 * (size_t)C + 4 * C + f];
        }
#undef BT
        if (accumulate)
            dP[idx] += v;
        else
            dP[idx] = v;
    }
}

// The same combination with a workgroup per (g, AB consecutive a): the four (a,c)-indexed rows (X_ac and the G slices of cases
// 14, 15, 42) of each a are staged in LDS once and serve every b; a thread owns (b, f), keeps its (a,b)-indexed terms in
// registers and walks c.  The (b,c)-indexed terms (X_bc and the slices of cases 23, 24, 45: four global loads per output with one
// a per workgroup, N x the bytes of those tables through L2 -- the kernel's bound at cfg5) are loaded once per c and serve the AB
// rows a of the workgroup.
template <int K, int VW, int AB, bool ACC>
__global__ __launch_bounds__(256) void fam_backward_rows(const float *__restrict__ G, const float *__restrict__ adjs,
                                                         const float *__restrict__ bsc, const float *__restrict__ btab,
                                                         float *__restrict__ dP, int N, int C,
                                                         int) {
    using V = typename Vec<VW>::T;
    // [AB][5][N][C]: X_ac(a,c) | G14(a,c) | G15(a,c) | G42(a,c) | Z_ab(c,a)
    extern __shared__ __attribute__((aligned(16))) float srow[];
    constexpr int NT = 5;
    const size_t NNC = (size_t)N * N * C, NC = (size_t)N * C;
    const size_t blk = xcd_block();
    const int na = (N + AB - 1) / AB;
    const size_t g = blk / na;
    const int a0 = (int)(blk % na) * AB;
    const float *bt = btab + g * kNBTab * NNC;
    const float *Gg = G + g * (size_t)N * N * K * C;
    const float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N;
#define GCF(cs, x, y, f) Vec<VW>::ld(Gg + (((size_t)(x) * N + (y)) * K + slot<K>(cs)) * C + (f))
    const int CV = C / VW, items = N * CV;
    for (int it = threadIdx.x; it < AB * items; it += blockDim.x) {
        const int m = it / items, c = (it % items) / CV, f = (it % CV) * VW;
        const int a = (a0 + m < N) ? a0 + m : N - 1;  // (rows past the end: a clamped copy, never stored)
        float *sr = srow + (size_t)m * NT * NC + (size_t)c * C + f;
        V xac = Vec<VW>::ld(bt + 1 * NNC + ((size_t)a * N + c) * C + f);
        Vec<VW>::st(sr + 0 * NC, xac);
        if (K == 50) {
            Vec<VW>::st(sr + 1 * NC, GCF(14, a, c, f));
            Vec<VW>::st(sr + 2 * NC, GCF(15, a, c, f));
            Vec<VW>::st(sr + 3 * NC, GCF(42, a, c, f));
            Vec<VW>::st(sr + 4 * NC, Vec<VW>::ld(bt + 5 * NNC + ((size_t)c * N + a) * C + f));   // Z_ab[c, a]: applies at a == b
        }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int b = it / CV, f = (it % CV) * VW;
        V xab[AB], g11[AB], g12[AB], g41[AB], zbc[AB], zac[AB];
        float ra[AB], qa[AB], dga[AB];
        const float rb = r[b], qb = q[b], dgb = dg[b];
        V u50 = Vec<VW>::zero();
        if (K == 50) u50 = Vec<VW>::ld(bsc + g * 5 * (size_t)C + 4 * C + f);
#pragma unroll
        for (int m = 0; m < AB; ++m) {
            const int a = (a0 + m < N) ? a0 + m : N - 1;
            xab[m] = Vec<VW>::ld(bt + 0 * NNC + ((size_t)a * N + b) * C + f);
            g11[m] = g12[m] = g41[m] = zbc[m] = zac[m] = Vec<VW>::zero();
            ra[m] = r[a], qa[m] = q[a], dga[m] = dg[a];
            if (K == 50) {
                g11[m] = GCF(11, a, b, f);
                g12[m] = GCF(12, a, b, f);
                g41[m] = GCF(41, a, b, f);
                zbc[m] = Vec<VW>::ld(bt + 3 * NNC + ((size_t)a * N + b) * C + f);   // applies at c == b
                zac[m] = Vec<VW>::ld(bt + 4 * NNC + ((size_t)b * N + a) * C + f);   // applies at c == a
            }
        }
        // The (b, c)-indexed operands of step c + 1 are requested BEFORE step c is combined and stored: the walk used to be one
        // round trip per step (four requests, wait for all, two stores) on 2.4 waves per SIMD -- two thirds of the waves' cycles
        // were waits (SQ_WAIT_ANY).  No conditional request inside the walk (the diagonal table Z_ab rides in the LDS image, the
        // read of the old dP is a template parameter): the compiler counts the queue instead of draining it.
        struct BC {
            V xbc, g23, g24, g45;
        };
        auto load_bc = [&](int c) {
            BC o;
            o.xbc = Vec<VW>::ld(bt + 2 * NNC + ((size_t)b * N + c) * C + f);
            o.g23 = o.g24 = o.g45 = Vec<VW>::zero();
            if (K == 50) o.g23 = GCF(23, b, c, f), o.g24 = GCF(24, b, c, f), o.g45 = GCF(45, b, c, f);
            return o;
        };
#ifndef GF_FAM_ROWS_PF2
#define GF_FAM_ROWS_PF2 0
#endif
        BC cur = load_bc(0);
#if GF_FAM_ROWS_PF2
        BC nx1 = load_bc(1 < N ? 1 : 0);
#endif
        for (int c = 0; c < N; ++c) {
#if GF_FAM_ROWS_PF2
            const BC nxt = nx1;
            nx1 = load_bc(c + 2 < N ? c + 2 : N - 1);
#else
            const BC nxt = load_bc(c + 1 < N ? c + 1 : c);   // (the last step re-requests its own row)
#endif
            float rc = 0.f, qc = 0.f, dgc = 0.f;
            if (K == 50) rc = r[c], qc = q[c], dgc = dg[c];
#pragma unroll
            for (int m = 0; m < AB; ++m) {
                const int a = a0 + m;
                const float *sr = srow + (size_t)m * NT * NC + (size_t)c * C + f;
                V v = xab[m] + Vec<VW>::ld(sr + 0 * NC) + cur.xbc;
                if (K == 50) {
                    v += g11[m] * rc + g12[m] * qc + g41[m] * dgc;
                    v += Vec<VW>::ld(sr + 1 * NC) * rb + Vec<VW>::ld(sr + 2 * NC) * qb + Vec<VW>::ld(sr + 3 * NC) * dgb;
                    v += cur.g23 * ra[m] + cur.g24 * qa[m] + cur.g45 * dga[m];
                    if (b == c) v += zbc[m];
                    if (a == c) v += zac[m];
                    if (a == b) v += Vec<VW>::ld(sr + 4 * NC);
                    if (a == b && b == c) v += u50;
                }
                if (a < N) {
                    float *out = dP + ((g * N + a) * N + b) * (size_t)N * C + (size_t)c * C + f;
                    if (ACC) v += Vec<VW>::ld(out);
                    Vec<VW>::template st_s<4>(out, v);
                }
            }
            cur = nxt;
        }
    }
#undef GCF
}

struct FamWs {
    float *adjs, *tab, *vec, *sc;
};

size_t fam_ws_floats(int N, int C, int batch) {
    const size_t NNC = (size_t)N * N * C;
    return (size_t)batch * (adjs_stride(N) + kNTab * NNC + kNVec * (size_t)N * C + kNSc * (size_t)C) + 64;
}

FamWs carve(float *ws, int N, int C, int batch) {
    FamWs w;
    w.adjs = ws;
    w.tab = w.adjs + align_up((size_t)batch * adjs_stride(N), 4);
    w.vec = w.tab + (size_t)batch * kNTab * N * N * C;
    w.sc = w.vec + (size_t)batch * kNVec * N * C;
    return w;
}

template <int K>
gf_status fam_forward_launch(gf_ctx *ctx, const float *P, const float *A, float *Out, int N, int C, int batch) {
    gf_status st = ensure_ws(ctx, sizeof(float) * fam_ws_floats(N, C, batch) + 256);
    if (st != GF_OK) return st;
    const FamWs w = carve(static_cast<float *>(ctx->ws), N, C, batch);
    const size_t nn = (size_t)batch * N * N * C, nv = (size_t)batch * N * C, ns = (size_t)batch * C;
    GF_LAUNCH(ctx, "fam_adj", fam_adj, dim3(batch), dim3(64), 0, A, w.adjs, N);
    const bool vec = C % 4 == 0 && (((uintptr_t)P | (uintptr_t)Out | (uintptr_t)w.tab) & 15) == 0;
    bool mfma_out = false;
    if constexpr (K == 50) {
        // all fifty slices of a row from one wave, the adjacency products on the matrix pipe (GF_FAM_FWD_MFMA=0: the table kernels below,
        // which serve every other shape and _10 -- the parity tests hold the two against each other)
        const char *e = std::getenv("GF_FAM_FWD_MFMA");
        mfma_out = vec && C % 32 == 0 && N <= 32 && (size_t)batch * N * (C / 32) < 0x7fffffffu && (size_t)N * N * N * C < (1u << 29) &&
                   !(e && e[0] == '0');   // (buffer descriptors of a graph's P and tables stay under 2 GB)
    }
    if (mfma_out) {
        if constexpr (K == 50) {
            // the pass over P writes the fifteen slices that ARE tables straight into Out; occupancy held at four workgroups per CU by
            // a 40 KB LDS request (measured: 0.170 - 0.182 ms from 0 to 160 KB)
            const size_t lds = (size_t)40 * 1024;
            st = opt_in_lds(ctx, fam50_tables_out<8>, lds);
            if (st != GF_OK) return st;
            GF_LAUNCH(ctx, "fam_tables", (fam50_tables_out<8>), dim3(grid_for(nn / 4)), dim3(256), lds, P, w.adjs, w.tab, Out, N, C, nn / 4);
        }
    } else if (vec && K == 10 && (((uintptr_t)Out) & 15) == 0) {
        // the three plain marginals with eight steps of the walk in flight (fam50_tables_out<8, 10>); S tot also goes straight into slices 1, 2, 5
        GF_LAUNCH(ctx, "fam_tables", (fam50_tables_out<8, 10>), dim3(grid_for(nn / 4)), dim3(256), 0, P, w.adjs, w.tab, Out, N, C, nn / 4);
    } else if (vec)
        GF_LAUNCH(ctx, "fam_tables", (fam_tables<K, 4>), dim3(grid_for(nn / 4)), dim3(256), 0, P, w.adjs, w.tab, N, C, nn / 4);
    else
        GF_LAUNCH(ctx, "fam_tables", (fam_tables<K, 1>), dim3(grid_for(nn)), dim3(256), 0, P, w.adjs, w.tab, N, C, nn);
    GF_LAUNCH(ctx, "fam_vectors", fam_vectors, dim3(grid_for(nv)), dim3(256), 0, P, w.tab, w.vec, w.sc, N, C, nv);
    GF_LAUNCH(ctx, "fam_scalars", fam_scalars, dim3(grid_for(ns)), dim3(256), 0, P, w.vec, w.sc, N, C, ns);
    if constexpr (K == 50) {
        if (mfma_out) {
            const unsigned nw = (unsigned)((size_t)batch * N * (C / 32));
            const dim3 grid((nw + 3) / 4), block(256);
            if (N <= 16)
                GF_LAUNCH(ctx, "fam_forward", (fam50_forward_mfma<8, true>), grid, block, 0, P, A, w.adjs, w.tab, w.vec, w.sc, Out, N, C, nw);
            else if (N <= 24)
                GF_LAUNCH(ctx, "fam_forward", (fam50_forward_mfma<12, true>), grid, block, 0, P, A, w.adjs, w.tab, w.vec, w.sc, Out, N, C, nw);
            else
                GF_LAUNCH(ctx, "fam_forward", (fam50_forward_mfma<16, true>), grid, block, 0, P, A, w.adjs, w.tab, w.vec, w.sc, Out, N, C, nw);
            return GF_OK;
        }
    }
    if (vec) {
        const size_t nf = (size_t)batch * N * N * (C / 4);
        GF_LAUNCH(ctx, "fam_forward", (fam_forward<K, 4>), dim3(grid_for(nf)), dim3(256), 0, P, A, w.adjs, w.tab, w.vec, w.sc,
                  Out, N, C, N, nf);
    } else {
        const int nyb = (N + 3) / 4;
        const size_t nf = (size_t)batch * N * nyb * C;
        GF_LAUNCH(ctx, "fam_forward", (fam_forward<K, 1>), dim3(grid_for(nf)), dim3(256), 0, P, A, w.adjs, w.tab, w.vec, w.sc,
                  Out, N, C, nyb, nf);
    }
    return GF_OK;
}

// bsc[g][5][C] = sum_x bpart[g][x][5][C], in the order of x
__global__ void fam50_bwd_scalars_fold(const float *__restrict__ bpart, float *__restrict__ bsc, int N, int C, size_t total) {
    GRID_STRIDE(idx, total) {
        const size_t g = idx / (5 * (size_t)C), jf = idx % (5 * (size_t)C);
        float sum = 0.f;
        for (int x = 0; x < N; ++x) sum += bpart[(g * N + x) * 5 * (size_t)C + jf];
        bsc[idx] = sum;
    }
}

template <int K>
gf_status fam_backward_launch(gf_ctx *ctx, const float *G, const float *A, float *dP, int N, int C, int batch,
                              int accumulate) {
    gf_status st = ensure_ws(ctx, sizeof(float) * fam_ws_floats(N, C, batch) + 256);
    if (st != GF_OK) return st;
    const FamWs w = carve(static_cast<float *>(ctx->ws), N, C, batch);  // tab doubles as btab, sc as bsc
    const size_t nn = (size_t)batch * N * N * C, np = nn * N;
    const bool vec = C % 4 == 0 && (((uintptr_t)G | (uintptr_t)dP | (uintptr_t)w.tab | (uintptr_t)w.sc) & 15) == 0;
    GF_LAUNCH(ctx, "fam_adj", fam_adj, dim3(batch), dim3(64), 0, A, w.adjs, N);
    bool mfma_tables = false;
    if constexpr (K == 50) {
        // matrix-pipe tables (every G row read once, no LDS): C in whole 32-channel windows, N <= 32.  GF_FAM_BWD_MFMA=0: the
        // thread-per-(i, columns, f) kernel below (every other shape, _10; the parity tests hold the two against each other)
        const char *e = std::getenv("GF_FAM_BWD_MFMA");
        mfma_tables = C % 32 == 0 && N <= 32 && (size_t)batch * N * (C / 32) < 0x7fffffffu && (size_t)N * N * K * C < (1u << 28) &&
                      !(e && e[0] == '0');
    }
    if constexpr (K == 50) {
        if (mfma_tables) {
            const unsigned nw = (unsigned)((size_t)batch * N * (C / 32));
            const dim3 grid((nw + 3) / 4), block(256);
            const size_t nsc = (size_t)batch * 5 * C;
#define GF_FAM_MFMA(NS)                                                                                                            \
    do {                                                                                                                           \
        GF_LAUNCH(ctx, "fam_bwd_tables_col", (fam50_bwd_tables_mfma<NS, 0>), grid, block, 0, G, A, w.adjs, w.sc, w.tab, w.vec, N, C, nw); \
        GF_LAUNCH(ctx, "fam_bwd_scalars", fam50_bwd_scalars_fold, dim3(grid_for(nsc)), dim3(256), 0, w.vec, w.sc, N, C, nsc);      \
        GF_LAUNCH(ctx, "fam_bwd_tables_row", (fam50_bwd_tables_mfma<NS, 1>), grid, block, 0, G, A, w.adjs, w.sc, w.tab, w.vec, N, C, nw); \
    } while (0)
            if (N <= 16) GF_FAM_MFMA(8); else if (N <= 24) GF_FAM_MFMA(12); else GF_FAM_MFMA(16);
#undef GF_FAM_MFMA
        }
    }
    if (!mfma_tables) {   // (the matrix-pipe column launch leaves per-row partials of the scalars instead)
        GF_LAUNCH(ctx, "fam_bwd_scalars", fam_bwd_scalars<K>, dim3(batch * 5), dim3(256), 0, G, A, w.sc, N, C);
        // one channel per thread, six columns x one or two rows (two once the grid fills the part): with four channels per thread only
        // two columns fit in registers and the shared (i, z) operands are re-read three times as often (measured at cfg5)
        const int njb = (N + 5) / 6;
        if ((size_t)batch * ((N + 1) / 2) * njb * C >= 256 * 1024) {
            const size_t nt = (size_t)batch * ((N + 1) / 2) * njb * C;
            GF_LAUNCH(ctx, "fam_bwd_tables", (fam_bwd_tables<K, 1, 2>), dim3(grid_for(nt)), dim3(256), 0, G, A, w.adjs, w.sc, w.tab, N,
                      C, njb, nt);
        } else {
            const size_t nt = (size_t)batch * N * njb * C;
            GF_LAUNCH(ctx, "fam_bwd_tables", (fam_bwd_tables<K, 1, 1>), dim3(grid_for(nt)), dim3(256), 0, G, A, w.adjs, w.sc, w.tab, N,
                      C, njb, nt);
        }
    }
    const size_t row_lds = sizeof(float) * 5 * (size_t)N * C;
    if (row_lds <= 48 * 1024 && (size_t)batch * N < 0x7fffffffu) {
        // rows a per workgroup: as many of one / two / three as fit 64 KB of staged rows while the grid still fills the part
        // (cfg5, with the walk's requests a step ahead: 0.197 / 0.171 / 0.178 ms at two / three / four rows)
        int ab = (2 * row_lds <= 64 * 1024 && (size_t)batch * ((N + 1) / 2) >= 1024) ? 2 : 1;
        if (3 * row_lds <= 64 * 1024 && (size_t)batch * ((N + 2) / 3) >= 1024) ab = 3;
        const unsigned nb = (unsigned)((size_t)batch * ((N + ab - 1) / ab));
#define GF_FAM_ROWS(VW, AB)                                                                                                        \
    do {                                                                                                                           \
        if (accumulate) {                                                                                                          \
            st = opt_in_lds(ctx, fam_backward_rows<K, VW, AB, true>, AB * row_lds);                                                \
            if (st != GF_OK) return st;                                                                                            \
            GF_LAUNCH(ctx, "fam_backward", (fam_backward_rows<K, VW, AB, true>), dim3(nb), dim3(256), AB * row_lds, G, w.adjs, w.sc, w.tab, \
                      dP, N, C, 0);                                                                                                \
        } else {                                                                                                                   \
            st = opt_in_lds(ctx, fam_backward_rows<K, VW, AB, false>, AB * row_lds);                                               \
            if (st != GF_OK) return st;                                                                                            \
            GF_LAUNCH(ctx, "fam_backward", (fam_backward_rows<K, VW, AB, false>), dim3(nb), dim3(256), AB * row_lds, G, w.adjs, w.sc, w.tab, \
                      dP, N, C, 0);                                                                                                \
        }                                                                                                                          \
    } while (0)
        if (vec) {  // (16-byte lanes: 0.26 -> 0.23 ms at cfg5)
            if (ab == 3) GF_FAM_ROWS(4, 3); else if (ab == 2) GF_FAM_ROWS(4, 2); else GF_FAM_ROWS(4, 1);
        } else {
            if (ab == 3) GF_FAM_ROWS(1, 3); else if (ab == 2) GF_FAM_ROWS(1, 2); else GF_FAM_ROWS(1, 1);
        }
#undef GF_FAM_ROWS
    } else {
        GF_LAUNCH(ctx, "fam_backward", fam_backward<K>, dim3(grid_for(np)), dim3(256), 0, G, A, w.adjs, w.sc, w.tab, dP, N, C,
                  np, accumulate);
    }
    return GF_OK;
}

}  // namespace

size_t family_workspace_bytes(int K, int N, int C, int batch) {
    if (K == 4) return 0;
    return sizeof(float) * fam_ws_floats(N, C, batch) + 256;
}

gf_status family_forward(gf_ctx *ctx, int K, const float *P, const float *A, float *Out, int N, int C, int batch) {
    if (K == 4) {
        if (r4_slab_ok(N, C, P, Out) && (size_t)batch * N < 0x7fffffffu) {
            const int ppw = 64 / (C / 4), slots = (N + ppw - 1) / ppw;
            const size_t lds = sizeof(float) * 4 * (size_t)N * C;
            const dim3 grid((unsigned)((size_t)batch * N)), block(256);
#define GF_R4F(S)                                                              \
    do {                                                                       \
        gf_status st = opt_in_lds(ctx, r4_fwd_slab<S>, lds);                   \
        if (st != GF_OK) return st;                                            \
        GF_LAUNCH(ctx, "r4_forward", r4_fwd_slab<S>, grid, block, lds, P, Out, N, C); \
    } while (0)
            if (slots <= 1) GF_R4F(1);
            else if (slots <= 2) GF_R4F(2);
            else if (slots <= 4) GF_R4F(4);
            else if (slots <= 8) GF_R4F(8);
            else GF_R4F(16);
#undef GF_R4F
            return GF_OK;
        }
        const size_t total = (size_t)batch * N * N * C;
        GF_LAUNCH(ctx, "r4_forward", r4_forward, dim3(grid_for(total)), dim3(256), 0, P, Out, N, C, total);
        return GF_OK;
    }
    if (K == 10 && r10_graph_ok(N, C, batch, P, Out, A)) {
        const int lpc = C / 4, ppw = 64 / lpc, slots = (N + ppw - 1) / ppw, nw = (N + 1) / 2;
        const size_t lds = sizeof(float) * (size_t)r10_lds(N, lpc, nw).total;
        const dim3 grid((unsigned)batch), block(64 * nw);
#define GF_R10F2(S, T)                                                                            \
    do {                                                                                          \
        gf_status st = opt_in_lds(ctx, r10_fwd_graph<S, T>, lds);                                 \
        if (st != GF_OK) return st;                                                               \
        GF_LAUNCH(ctx, "r10_forward", (r10_fwd_graph<S, T>), grid, block, lds, P, A, Out, N, C, nw); \
    } while (0)
#define GF_R10F(S)                   \
    do {                             \
        if (nw <= 12) GF_R10F2(S, 768); \
        else GF_R10F2(S, 1024);      \
    } while (0)
        if (slots <= 1) GF_R10F(1);
        else if (slots <= 2) GF_R10F(2);
        else if (slots <= 3) GF_R10F(3);
        else GF_R10F(4);
#undef GF_R10F2
#undef GF_R10F
        return GF_OK;
    }
    if (K == 10) return fam_forward_launch<10>(ctx, P, A, Out, N, C, batch);
    if (K == 50) return fam_forward_launch<50>(ctx, P, A, Out, N, C, batch);
    return fail(ctx, GF_ERR_INVALID, "family_forward: K=%d", K);
}

gf_status family_backward(gf_ctx *ctx, int K, const float *G, const float *A, float *dP, int N, int C, int batch,
                          int accumulate) {
    if (K == 4) {
        if (r4_slab_ok(N, C, G, dP) && (size_t)batch * N < 0x7fffffffu) {
            const int ppw = 64 / (C / 4), slots = (N + ppw - 1) / ppw;
            const dim3 grid((unsigned)((size_t)batch * N)), block(256);
#define GF_R4B(S)                                                                                 \
    do {                                                                                          \
        if (accumulate) GF_LAUNCH(ctx, "r4_backward", (r4_bwd_slab<S, true>), grid, block, 0, G, dP, N, C); \
        else GF_LAUNCH(ctx, "r4_backward", (r4_bwd_slab<S, false>), grid, block, 0, G, dP, N, C);           \
    } while (0)
            if (slots <= 1) GF_R4B(1);
            else if (slots <= 2) GF_R4B(2);
            else if (slots <= 4) GF_R4B(4);
            else if (slots <= 8) GF_R4B(8);
            else GF_R4B(16);
#undef GF_R4B
            return GF_OK;
        }
        const size_t total = (size_t)batch * N * N * N * C;
        GF_LAUNCH(ctx, "r4_backward", r4_backward, dim3(grid_for(total)), dim3(256), 0, G, dP, N, C, total, accumulate);
        return GF_OK;
    }
    if (K == 10 && r10_graph_ok(N, C, batch, G, dP, A)) {
        const int lpc = C / 4, ppw = 64 / lpc, slots = (N + ppw - 1) / ppw, nw = (N + 1) / 2;
        const size_t lds = sizeof(float) * (size_t)r10_lds(N, lpc, nw).total;
        const dim3 grid((unsigned)batch), block(64 * nw);
#define GF_R10B2(S, T)                                                                                                   \
    do {                                                                                                                 \
        gf_status st = opt_in_lds(ctx, r10_bwd_graph<S, true, T>, lds);                                                  \
        if (st == GF_OK) st = opt_in_lds(ctx, r10_bwd_graph<S, false, T>, lds);                                          \
        if (st != GF_OK) return st;                                                                                      \
        if (accumulate) GF_LAUNCH(ctx, "r10_backward", (r10_bwd_graph<S, true, T>), grid, block, lds, G, A, dP, N, C, nw); \
        else GF_LAUNCH(ctx, "r10_backward", (r10_bwd_graph<S, false, T>), grid, block, lds, G, A, dP, N, C, nw);          \
    } while (0)
#define GF_R10B(S)                   \
    do {                             \
        if (nw <= 12) GF_R10B2(S, 768); \
        else GF_R10B2(S, 1024);      \
    } while (0)
        if (slots <= 1) GF_R10B(1);
        else if (slots <= 2) GF_R10B(2);
        else if (slots <= 3) GF_R10B(3);
        else GF_R10B(4);
#undef GF_R10B2
#undef GF_R10B
        return GF_OK;
    }
    if (K == 10) return fam_backward_launch<10>(ctx, G, A, dP, N, C, batch, accumulate);
    if (K == 50) return fam_backward_launch<50>(ctx, G, A, dP, N, C, batch, accumulate);
    return fail(ctx, GF_ERR_INVALID, "family_backward: K=%d", K);
}

}  // namespace gf
