// contract18.hip -- RisiContraction_18 forward/backward for gfx950 (MI355X), factorised O(N^3 C) form.
//
// Replaces GraphFlow/RisiContraction_18.h:73-331 (forward) and :333-560 (backward), whose loop nests cost
// O(nnz(A) N^3 C).  Every one of the 18 cases factorises into (i) one streaming pass over P[a][b][c][f] that
// builds a handful of N x N (per channel) tables and (ii) tiny N x N products with the gated adjacency
// A+ = A*[A>0].  The pass is HBM-bound; nothing here is GEMM-shaped enough for MFMA.
//
// Decomposition (per graph g, channel window of CW = 4*LPC channels):
//   forward   F1 "slab" : one workgroup per (g, b) streams the slab P[g][:, b, :, :] once (coalesced 16 B/lane over
//                         the channel axis, lanes = (c-group, channel quad)), owns every quantity indexed by b
//                         (S_ab[:,b], S_bc[b,:], the r-weighted sums, the diagonals) and writes 10 of the 18 slices
//                         directly; it leaves S_ab[:,b], P[:,b,b] and 4 partial scalars in a small workspace.
//             F2 "rows" : one workgroup per (g, a) finishes the 8 slices that need a sum over b from that workspace.
//   backward  B1 "rows" : one workgroup per (g, a) reduces the 8 (a,d)- and (d,e)-indexed gradient slices to two
//                         N x N x C tables and 4 partial scalars.
//             B2 "slab" : one workgroup per (g, b) builds the per-b tables from the other 10 slices and streams
//                         dP[g][:, b, :, :] out (write-only, or read-modify-write when accumulate != 0).
// Algorithmic HBM bytes per graph (fp32): fwd 4(N^3 C + N^2 + 18 N^2 C), bwd the same (+4 N^3 C with accumulate);
// the workspace adds 2 x 4(2 N^2 C + 4 N C) per direction (about 8 % at N=32, C=64).
//
// Shapes outside the slab kernels' reach (C % 4 != 0, misaligned pointers, N > 8 * 64/LPC) take the "generic"
// kernels at the bottom of this file: same maths, one thread per output element, no layout assumptions.
#include "gf_internal.h"
#include "r18_device.h"

// r18_bwd_slab's streaming accesses (A/B switches: -DGF_NT_BS_LD=0 / -DGF_NT_BS_ST=0)
#ifndef GF_NT_BS_LD
#define GF_NT_BS_LD 1
#endif
#ifndef GF_NT_BS_ST
#define GF_NT_BS_ST 1
#endif
#if GF_NT_BS_LD
#define BS_LD ld4_nt
#else
#define BS_LD ld4
#endif
#if GF_NT_BS_ST
#define BS_ST st4_nt
#else
#define BS_ST st4
#endif
namespace gf {
namespace {

using namespace gf::dev;

// ------------------------------------------------------------------------------------------------------------
// F1: forward slab kernel.  Workgroup (g, b).  Wave w owns rows a = w, w+4, ...; lane = (cg, fl):
// c = i*PPW + cg for load i, channels f0 + 4*fl .. +3.  Per row: NI coalesced 16 B loads per lane (+1 for the two
// diagonal elements P[a,b,b], P[a,b,a]), register accumulation over a (S_bc, T10) and shuffle reduction over c
// (S_ab, T6).  Rows are double-buffered in registers so the next row's loads are in flight while this one is reduced.
// The streaming loop is branch-free: ragged shapes (FULL == false: N < NI*PPW or C % CW != 0) clamp the load
// address into the row and multiply the value by a 0/1 lane mask instead of predicating the load.
// ------------------------------------------------------------------------------------------------------------
template <int LPC, int NI>
struct SlabLane {
    int coff[NI];    // BYTE offset of this lane's i-th load inside a row (clamped to a valid c)
    float cmask[NI];  // 1 where (c < N and channel quad in range), else 0
};

template <int LPC, int NI, bool FULL>
__device__ __forceinline__ void load_slab_row(__amdgpu_buffer_rsrc_t slab, int row_bytes, const SlabLane<LPC, NI> &ln,
                                              int dgoff_bytes, f4 (&v)[NI], f4 &dg) {
#ifndef GF_R18_DG
#define GF_R18_DG 1
#endif
    // (the two diagonal elements lie in lines the row loads fetch as well: A/B of their order / policy, GF_R18_DG)
    if (GF_R18_DG == 1) dg = buf_ld4(slab, dgoff_bytes, row_bytes);
    if (GF_R18_DG == 2) dg = buf_ld4_nt(slab, dgoff_bytes, row_bytes);
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = buf_ld4_nt(slab, ln.coff[i], row_bytes);
    if (GF_R18_DG == 0) dg = buf_ld4_nt(slab, dgoff_bytes, row_bytes);
    if (GF_R18_DG == 3) dg = buf_ld4(slab, dgoff_bytes, row_bytes);
}

template <int LPC, int NI, bool FULL>
__global__ __launch_bounds__(kThreads) void r18_fwd_slab(const float *__restrict__ P, const float *__restrict__ A,
                                                         float *__restrict__ Out, float *__restrict__ wsSab,
                                                         float *__restrict__ wsDbb, float *__restrict__ wsScal, Ragged R,
                                                         int C, int nwin) {
    constexpr int PPW = 64 / LPC;
    constexpr int CW = 4 * LPC;
    constexpr int NCP = NI * PPW;  // padded c extent (>= N)
    static_assert(PPW >= 4, "row epilogue uses four c-groups");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row base addresses stay in SGPRs
    const int cg = lane / LPC, fl = lane % LPC;
    const Where W = locate(R, nwin);
    const int N = W.N, b = W.i;
    const size_t rowbase = W.rowbase, pbase = W.pbase, pairbase = W.pairbase;
    (void)pbase;
    (void)pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = FULL || f < C;
    const int fld = fok ? f : 0;  // clamped channel offset for loads

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency<false>(smem, A + rowbase, N);
    float *sSab = smem + adj_lds_floats(N);  // [N][CW]    S_ab[e, b]
    float *sDac = sSab + N * CW;             // [N][CW]    P[e, b, e]
    float *sSbc = sDac + N * CW;             // [NCP][CW]  S_bc[b, e]
    float *sRed = sSbc + NCP * CW;           // [2][NCP][CW] cross-wave reduction buffer
    float *sMisc = sRed + 2 * NCP * CW;      // [kWaves][2][CW] diagonal sums, then [CW] colsum
    const float tot = L.st[0], tr = L.st[1];

    SlabLane<LPC, NI> ln;
    float rc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * PPW + cg;
        const bool ok = FULL || (c < N);
        ln.coff[i] = ((ok ? c : 0) * C + fld) * 4;
        ln.cmask[i] = (ok && fok) ? 1.f : 0.f;
        rc[i] = (ok && fok) ? L.r[ok ? c : 0] : 0.f;
    }
    const float dmask = (cg < 2 && fok) ? 1.f : 0.f;

    // slab descriptor: base P[g][0][b][0][0]; row a starts a*N*N*C floats further (scalar offset)
    const float *slab0 = P + pbase * C + (size_t)b * N * C;
    const __amdgpu_buffer_rsrc_t slab = make_rsrc(slab0, ((size_t)N * N * N * C - (size_t)b * N * C) * 4);
    const int rowBytes = N * N * C * 4;
    const int dgA = fld * 4 + C * 4 * b;  // lanes cg==0 read P[a,b,b]; lanes cg!=0 read P[a,b,a] (offset depends on a)

    f4 sbc[NI], t10[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) sbc[i] = t10[i] = splat(0.f);
    f4 dgsum = splat(0.f);  // cg==0: sum_a P[a,b,b]   cg==1: sum_a P[a,b,a]

    f4 cur[NI], nxt[NI], dcur, dnxt;
    {
        const int a0 = wave < N ? wave : 0;
        load_slab_row<LPC, NI, FULL>(slab, a0 * rowBytes, ln, (cg == 0) ? dgA : (fld + C * a0) * 4, cur, dcur);
    }
    for (int a = wave; a < N; a += kWaves) {
        const int an = (a + kWaves < N) ? a + kWaves : a;  // last iteration re-reads its own row (L1/L2 hit)
        load_slab_row<LPC, NI, FULL>(slab, an * rowBytes, ln, (cg == 0) ? dgA : (fld + C * an) * 4, nxt, dnxt);
        const float ra = L.r[a];
        f4 sab = splat(0.f), t6 = splat(0.f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f4 v = cur[i];
            sbc[i] += v;  // lanes with c >= N accumulate clamped duplicates that are never stored
            t10[i] += ra * v;
            if (FULL)
                sab += v;
            else
                sab += ln.cmask[i] * v;
            t6 += rc[i] * v;
        }
        sab = reduce_cgroups<LPC>(sab);
        t6 = reduce_cgroups<LPC>(t6);
        const f4 dv = dcur * dmask;
        dgsum += dv;
        // row epilogue: four c-groups each issue one 16 B store per lane (256 B segments at C=64)
        const size_t oab = (rowbase + (size_t)a * N + b) * (size_t)(kK * C) + f;  // Out[g][a][b][.][f]
        const size_t wab = (rowbase + (size_t)a * N + b) * (size_t)C + f;         // ws[g][a][b][f]
        if (cg == 0) {
            if (fok) {
                st4_nt(Out + oab + 0 * C, sab * tot);  // k0  S_ab*tot
                st4(wsDbb + wab, dv);               // P[a,b,b] for F2
            }
        } else if (cg == 1) {
            if (fok) st4_nt(Out + oab + 6 * C, sab * tr);  // k6  S_ab*tr
            st4(sDac + a * CW + 4 * fl, dv);            // P[a,b,a]
        } else if (cg == 2) {
            if (fok) st4_nt(Out + oab + 5 * C, t6);  // k5  sum_c P[a,b,c] r[c]
        } else if (cg == 3) {
            if (fok) st4(wsSab + wab, sab);
            st4(sSab + a * CW + 4 * fl, sab);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
        dcur = dnxt;
    }

    // cross-wave reduction of the a-sums: waves 1..3 hand their partials to wave 0 one at a time (deterministic
    // order).  The buffers are padded to NCP rows so no lane needs a bounds branch.
#pragma unroll 1
    for (int w = 1; w < kWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                st4(sRed + (i * PPW + cg) * CW + 4 * fl, sbc[i]);
                st4(sRed + (NCP + i * PPW + cg) * CW + 4 * fl, t10[i]);
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                sbc[i] += ld4(sRed + (i * PPW + cg) * CW + 4 * fl);
                t10[i] += ld4(sRed + (NCP + i * PPW + cg) * CW + 4 * fl);
            }
        }
        __syncthreads();
    }
    if (cg < 2) st4(sMisc + (wave * 2 + cg) * CW + 4 * fl, dgsum);
    if (wave == 0) {
        f4 cs = splat(0.f);
        float *obc = Out + (rowbase + (size_t)b * N) * (size_t)(kK * C) + f;  // Out[g][b][c][.][f]
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = i * PPW + cg;
            const f4 sv = FULL ? sbc[i] : sbc[i] * ln.cmask[i];
            cs += sv;
            st4(sSbc + c * CW + 4 * fl, sv);
            if (FULL || ln.cmask[i] != 0.f) {
                st4_nt(obc + (size_t)c * (kK * C) + 2 * C, sbc[i] * tot);  // k2  S_bc*tot
                st4_nt(obc + (size_t)c * (kK * C) + 9 * C, t10[i]);        // k9  sum_a r[a] P[a,b,c]
            }
        }
        cs = reduce_cgroups<LPC>(cs);
        if (cg == 0) st4(sMisc + kWaves * 2 * CW + 4 * fl, cs);  // colsum_b = sum_{a,c} P[a,b,c]
    }
    __syncthreads();

    // (b,d)-indexed slices: three N x N products with A+ plus two outer products with r.
    // Table-building role: thread = (row group grp, channel quad fl); LPC divides 64 so tid % LPC == fl.
    constexpr int NGRP = kThreads / LPC;
    const int grp = tid / LPC;
    f4 dbbtot = splat(0.f), dactot = splat(0.f);
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        dbbtot += ld4(sMisc + (w * 2 + 0) * CW + 4 * fl);
        dactot += ld4(sMisc + (w * 2 + 1) * CW + 4 * fl);
    }
    const f4 colsum = ld4(sMisc + kWaves * 2 * CW + 4 * fl);
    const float *const T[3] = {sSab, sSbc, sDac};
    for (int d = grp; d < N; d += NGRP) {
        f4 m[3];
        small_matvec<3, CW>(L, N, d, fl, T, m);
        const float rd = L.r[d];
        if (fok) {
            float *o = Out + (rowbase + (size_t)b * N + d) * (size_t)(kK * C) + f;
            st4_nt(o + 3 * C, colsum * rd);   // k3   (sum_{a,c} P[a,b,c]) r[d]
            st4_nt(o + 10 * C, dactot * rd);  // k10  (sum_a P[a,b,a]) r[d]
            st4_nt(o + 11 * C, m[0]);         // k11  sum_e A[d,e] S_ab[e,b]
            st4_nt(o + 12 * C, m[1]);         // k12  sum_e A[d,e] S_bc[b,e]
            st4_nt(o + 16 * C, m[2]);         // k16  sum_e A[d,e] P[e,b,e]
        }
    }
    if (grp == 0 && fok) {
        float *s = wsScal + (pairbase + b) * 4 * (size_t)C + f;
        st4(s + 0 * C, colsum);                       // -> total = sum_b colsum_b
        st4(s + 1 * C, ld4(sSab + b * CW + 4 * fl));  // -> s14   = sum_a S_ab[a,a]
        st4(s + 2 * C, dbbtot);                       // -> s15   = sum_{a,b} P[a,b,b]
        st4(s + 3 * C, ld4(sDac + b * CW + 4 * fl));  // -> s18   = sum_a P[a,a,a]
    }
}

template <int LPC, int NI>
static size_t fwd_slab_lds_bytes(int N) {
    constexpr int CW = 4 * LPC;
    constexpr int NCP = NI * (64 / LPC);
    return sizeof(float) * ((size_t)adj_lds_floats(N) + 2 * (size_t)N * CW + 3 * (size_t)NCP * CW + (kWaves * 2 + 1) * CW);
}

// ------------------------------------------------------------------------------------------------------------
// F2: forward row kernel.  Workgroup (g, a) finishes the (a,d)- and (d=a,e)-indexed slices.
// ------------------------------------------------------------------------------------------------------------
template <int LPC>
__global__ __launch_bounds__(kThreads) void r18_fwd_rows(const float *__restrict__ A, float *__restrict__ Out,
                                                         const float *__restrict__ wsSab,
                                                         const float *__restrict__ wsDbb,
                                                         const float *__restrict__ wsScal, Ragged R, int C, int nwin) {
    constexpr int CW = 4 * LPC;
    constexpr int NGRP = kThreads / LPC;
    const int tid = threadIdx.x;
    const int grp = tid / LPC, fl = tid % LPC;
    const Where W = locate(R, nwin);
    const int N = W.N, a = W.i;
    const size_t rowbase = W.rowbase, pbase = W.pbase, pairbase = W.pairbase;
    (void)pbase;
    (void)pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency<false>(smem, A + rowbase, N);
    float *sT0 = smem + adj_lds_floats(N);  // [N][CW] S_ab[a, e]
    float *sT1 = sT0 + N * CW;              // [N][CW] P[a, e, e]
    float *sS = sT1 + N * CW;               // [(NPART+1)*4][CW] pieces, then total, s14, s15, s18

    for (int e = grp; e < N; e += NGRP) {
        const size_t w = (rowbase + (size_t)a * N + e) * (size_t)C + f;
        st4(sT0 + e * CW + 4 * fl, fok ? ld4(wsSab + w) : splat(0.f));
        st4(sT1 + e * CW + 4 * fl, fok ? ld4(wsDbb + w) : splat(0.f));
    }
    {   // four length-N sums of the partial scalars, each split in NPART pieces over the thread groups
        constexpr int NPART = (NGRP >= 16) ? 4 : 1;
        const int j = grp % 4, part = grp / 4;
        if (part < NPART) {
            f4 sum = splat(0.f);
            if (fok) {
                const int per = (N + NPART - 1) / NPART;
                const int lo = part * per, hi = (lo + per < N) ? lo + per : N;
                if (lo < hi)
                    sum = batched_sum(wsScal + (pairbase * 4 + j) * (size_t)C + f, 4 * (size_t)C, lo, hi,
                                      [](int) { return 1.f; });
            }
            st4(sS + (part * 4 + j) * CW + 4 * fl, sum);
        }
    }
    __syncthreads();
    if (tid < 4 * LPC) {  // fold the pieces
        constexpr int NPART = (NGRP >= 16) ? 4 : 1;
        f4 sum = splat(0.f);
#pragma unroll
        for (int p = 0; p < NPART; ++p) sum += ld4(sS + (p * 4 + grp) * CW + 4 * fl);
        st4(sS + (NPART * 4 + grp) * CW + 4 * fl, sum);
    }
    __syncthreads();
    {
        constexpr int NPART = (NGRP >= 16) ? 4 : 1;
        sS += NPART * 4 * CW;  // folded totals live behind the pieces
    }

    f4 rowsum = splat(0.f), d8 = splat(0.f);
    for (int e = 0; e < N; ++e) {
        rowsum += ld4(sT0 + e * CW + 4 * fl);
        d8 += ld4(sT1 + e * CW + 4 * fl);
    }
    const f4 total = ld4(sS + 0 * CW + 4 * fl), s14 = ld4(sS + 1 * CW + 4 * fl);
    const f4 s15 = ld4(sS + 2 * CW + 4 * fl), s18 = ld4(sS + 3 * CW + 4 * fl);
    const float *const T[2] = {sT0, sT1};
    for (int y = grp; y < N; y += NGRP) {
        f4 m[2];
        small_matvec<2, CW>(L, N, y, fl, T, m);
        const float ry = L.r[y], aay = L.at(a, y, N);
        if (fok) {
            float *o = Out + (rowbase + (size_t)a * N + y) * (size_t)(kK * C) + f;
            st4_nt(o + 1 * C, rowsum * ry);  // k1   (sum_{b,c} P[a,b,c]) r[d]
            st4_nt(o + 7 * C, d8 * ry);      // k7   (sum_b P[a,b,b]) r[d]
            st4_nt(o + 8 * C, m[0]);         // k8   sum_e A[d,e] S_ab[a,e]
            st4_nt(o + 15 * C, m[1]);        // k15  sum_e A[d,e] P[a,e,e]
            st4_nt(o + 4 * C, total * aay);  // k4   A[d,e] sum_{abc} P          (d = a, e = y)
            st4_nt(o + 13 * C, s14 * aay);   // k13  A[d,e] sum_{a,c} P[a,a,c]
            st4_nt(o + 14 * C, s15 * aay);   // k14  A[d,e] sum_{a,b} P[a,b,b]
            st4_nt(o + 17 * C, s18 * aay);   // k17  A[d,e] sum_a P[a,a,a]
        }
    }
}

template <int LPC>
static size_t fwd_rows_lds_bytes(int N) {
    constexpr int CW = 4 * LPC;
    return sizeof(float) * ((size_t)adj_lds_floats(N) + 2 * (size_t)N * CW + 20 * CW);
}

// ------------------------------------------------------------------------------------------------------------
// B1: backward row kernel.  Workgroup (g, a): from G[g][a][d][k] (k = 1,7,8,15) and G[g][a][e][k] (k = 4,13,14,17)
//   WX[a,b] = sum_d G1[a,d] r[d] + sum_d G8[a,d] A[d,b]        (cases 2, 9)
//   WZ[a,b] = sum_d G7[a,d] r[d] + sum_d G15[a,d] A[d,b]       (cases 8, 16; applied where b == c)
//   part[a][j] = sum_e G_{4,13,14,17}[a,e] A[a,e]              (row a of the four (d,e)-indexed cases)
// ------------------------------------------------------------------------------------------------------------
template <int LPC>
__global__ __launch_bounds__(kThreads) void r18_bwd_rows(const float *__restrict__ G, const float *__restrict__ A,
                                                         float *__restrict__ wsWX, float *__restrict__ wsWZ,
                                                         float *__restrict__ wsPart, Ragged R, int C, int nwin) {
    constexpr int CW = 4 * LPC;
    constexpr int NGRP = kThreads / LPC;
    static_assert(NGRP >= 12, "six reductions in two halves are spread over thread groups");
    const int tid = threadIdx.x;
    const int grp = tid / LPC, fl = tid % LPC;
    const Where W = locate(R, nwin);
    const int N = W.N, a = W.i;
    const size_t rowbase = W.rowbase, pbase = W.pbase, pairbase = W.pairbase;
    (void)pbase;
    (void)pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency<true>(smem, A + rowbase, N);  // L.A[b][d] = A+[d][b]
    float *sT8 = smem + adj_lds_floats(N);  // [N][CW] G8[a, d]
    float *sT15 = sT8 + N * CW;             // [N][CW] G15[a, d]
    float *sU = sT15 + N * CW;              // [6][CW]

    const float *Grow = G + (rowbase + (size_t)a * N) * (size_t)(kK * C) + f;  // + (y*18 + k)*C
    for (int d = grp; d < N; d += NGRP) {
        st4(sT8 + d * CW + 4 * fl, fok ? ld4_nt(Grow + ((size_t)d * kK + 8) * C) : splat(0.f));
        st4(sT15 + d * CW + 4 * fl, fok ? ld4_nt(Grow + ((size_t)d * kK + 15) * C) : splat(0.f));
    }
    {   // six length-N reductions, each split in two halves over 12 thread groups
        const int j = grp % 6, half = grp / 6;
        if (half < 2) {
            const int kk = (j == 0) ? 1 : (j == 1) ? 7 : (j == 2) ? 4 : (j == 3) ? 13 : (j == 4) ? 14 : 17;
            f4 sum = splat(0.f);
            if (fok) {
                const int lo = half ? (N + 1) / 2 : 0, hi = half ? N : (N + 1) / 2;
                const float *w = (j < 2) ? L.r : (L.A + a);          // r[d]  or  A+[a][e] (stored transposed)
                const int ws = (j < 2) ? 1 : (N + 1);
                if (lo < hi)
                    sum = batched_sum(Grow + (size_t)kk * C, (size_t)kK * C, lo, hi, [=](int y) { return w[y * ws]; });
            }
            st4(sU + (half * 6 + j) * CW + 4 * fl, sum);
        }
    }
    __syncthreads();

    const f4 u2 = ld4(sU + 0 * CW + 4 * fl) + ld4(sU + 6 * CW + 4 * fl);
    const f4 u8 = ld4(sU + 1 * CW + 4 * fl) + ld4(sU + 7 * CW + 4 * fl);
    const float *const T[2] = {sT8, sT15};
    for (int bb = grp; bb < N; bb += NGRP) {
        f4 m[2];
        small_matvec<2, CW>(L, N, bb, fl, T, m);  // sum_d A+[d][bb] * G[a,d]
        if (fok) {
            const size_t w = (rowbase + (size_t)a * N + bb) * (size_t)C + f;
            st4(wsWX + w, u2 + m[0]);
            st4(wsWZ + w, u8 + m[1]);
        }
    }
    if (grp < 4 && fok)
        st4(wsPart + ((pairbase + a) * 4 + grp) * (size_t)C + f,
            ld4(sU + (2 + grp) * CW + 4 * fl) + ld4(sU + (8 + grp) * CW + 4 * fl));
}

template <int LPC>
static size_t bwd_rows_lds_bytes(int N) {
    constexpr int CW = 4 * LPC;
    return sizeof(float) * ((size_t)adj_lds_floats(N) + 2 * (size_t)N * CW + 12 * CW);
}

// ------------------------------------------------------------------------------------------------------------
// B2: backward slab kernel.  Workgroup (g, b) writes dP[g][:, b, :, :]:
//   dP[a,b,c] = X[a] + Y[c] + G5[a,b] r[c] + G9[b,c] r[a] + [c==b] Z1[a] + [c==a] Z2[a]
//   X[a]  = tot G0[a,b] + tr G6[a,b] + WX[a,b] + U4 + u5 + V12[a] + [a==b] u14
//   Y[c]  = tot G2[b,c] + V13[c]
//   Z1[a] = WZ[a,b] + u15 + [a==b] u18          Z2[a] = U11 + V17[a]
//   U4 = sum_d G3[b,d] r[d], U11 = sum_d G10[b,d] r[d], V12[a] = sum_d G11[b,d] A[d,a], V13[c] = sum_d G12[b,d] A[d,c],
//   V17[a] = sum_d G16[b,d] A[d,a];  u5,u14,u15,u18 = sum_a part[a][.]
// ------------------------------------------------------------------------------------------------------------
template <int LPC, int NI, bool FULL, bool ACC>
__global__ __launch_bounds__(kThreads, 3) void r18_bwd_slab(const float *__restrict__ G, const float *__restrict__ A,
                                                         float *__restrict__ dP, const float *__restrict__ wsWX,
                                                         const float *__restrict__ wsWZ,
                                                         const float *__restrict__ wsPart, Ragged R, int C, int nwin) {
    constexpr int PPW = 64 / LPC;
    constexpr int CW = 4 * LPC;
    constexpr int NGRP = kThreads / LPC;
    static_assert(NGRP >= 12, "six reductions in two halves are spread over thread groups");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: row base addresses stay in SGPRs
    const int cg = lane / LPC, fl = lane % LPC;  // streaming role
    const int grp = tid / LPC;                   // table-building role (same fl: LPC divides 64)
    const Where W = locate(R, nwin);
    const int N = W.N, b = W.i;
    const size_t rowbase = W.rowbase, pbase = W.pbase, pairbase = W.pairbase;
    (void)pbase;
    (void)pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency<true>(smem, A + rowbase, N);  // L.A[y][d] = A+[d][y]
    // LDS: phase (i) holds the three G tables feeding the A^T products; phase (ii) overwrites them with the five
    // tables the streaming loop reads (the products are held in registers across the barrier in between).
    float *sT11 = smem + adj_lds_floats(N);  // [N][CW] G11[b, d]
    float *sT12 = sT11 + N * CW;             //         G12[b, d]
    float *sT16 = sT12 + N * CW;             //         G16[b, d]
    float *sX = sT11;                        // [N][CW] (aliases)
    float *sY = sT12;
    float *sG5 = sT16;
    float *sZ1 = sT16 + N * CW;
    float *sZ2 = sZ1 + N * CW;
    float *sU = sZ2 + N * CW;  // [2][6][CW]: two half-sums of U4, U11, u5, u14, u15, u18
    const float tot = L.st[0], tr = L.st[1];

    const float *Grow = G + (rowbase + (size_t)b * N) * (size_t)(kK * C) + f;  // G[g][b][y][k][f]
    for (int d = grp; d < N; d += NGRP) {
        st4(sT11 + d * CW + 4 * fl, fok ? BS_LD(Grow + ((size_t)d * kK + 11) * C) : splat(0.f));
        st4(sT12 + d * CW + 4 * fl, fok ? BS_LD(Grow + ((size_t)d * kK + 12) * C) : splat(0.f));
        st4(sT16 + d * CW + 4 * fl, fok ? BS_LD(Grow + ((size_t)d * kK + 16) * C) : splat(0.f));
    }
    {   // six length-N reductions, each split in two halves over 12 thread groups
        const int j = grp % 6, half = grp / 6;
        if (half < 2) {
            f4 s = splat(0.f);
            if (fok) {
                const int lo = half ? (N + 1) / 2 : 0, hi = half ? N : (N + 1) / 2;
                if (lo < hi) {
                    if (j < 2) {
                        const float *rr = L.r;
                        s = batched_sum(Grow + (size_t)((j == 0) ? 3 : 10) * C, (size_t)kK * C, lo, hi,
                                        [=](int d) { return rr[d]; });
                    } else {
                        s = batched_sum(wsPart + (pairbase * 4 + (j - 2)) * (size_t)C + f, 4 * (size_t)C, lo, hi,
                                        [](int) { return 1.f; });
                    }
                }
            }
            st4(sU + (half * 6 + j) * CW + 4 * fl, s);
        }
    }
    __syncthreads();

    {
        f4 u[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) u[j] = ld4(sU + j * CW + 4 * fl) + ld4(sU + (6 + j) * CW + 4 * fl);
        const f4 u4 = u[0], u11 = u[1], u5 = u[2], u14 = u[3], u15 = u[4], u18 = u[5];
        const float *const T[3] = {sT11, sT12, sT16};
        constexpr int YB = (8 * PPW + NGRP - 1) / NGRP;  // rows per thread group: N <= 8*PPW  =>  YB <= 2
        f4 tx[YB], ty[YB], tg5[YB], tz1[YB], tz2[YB];
#pragma unroll
        for (int q = 0; q < YB; ++q) {
            const int y = grp + q * NGRP;
            const int yy = y < N ? y : 0;
            f4 m[3];
            small_matvec<3, CW>(L, N, yy, fl, T, m);  // V12[y], V13[y], V17[y]
            const int fc = fok ? f : 0;
            const float *gab = G + (rowbase + (size_t)yy * N + b) * (size_t)(kK * C) + fc;  // G[g][y][b][k][f]
            const f4 g0 = BS_LD(gab + 0 * C), g6 = BS_LD(gab + 6 * C), g5 = BS_LD(gab + 5 * C);
            const size_t w = (rowbase + (size_t)yy * N + b) * (size_t)C + fc;
            const f4 wx = ld4(wsWX + w), wz = ld4(wsWZ + w);
            const f4 g2 = BS_LD(G + (rowbase + (size_t)b * N + yy) * (size_t)(kK * C) + 2 * C + fc);
            f4 x = tot * g0 + tr * g6 + wx + u4 + u5 + m[0];
            f4 z1 = wz + u15;
            if (yy == b) {
                x += u14;
                z1 += u18;
            }
            tx[q] = x;
            ty[q] = tot * g2 + m[1];
            tg5[q] = g5;
            tz1[q] = z1;
            tz2[q] = u11 + m[2];
        }
        __syncthreads();  // every thread is done reading the G tables: overwrite them
#pragma unroll
        for (int q = 0; q < YB; ++q) {
            const int y = grp + q * NGRP;
            if (y < N) {
                st4(sX + y * CW + 4 * fl, tx[q]);
                st4(sY + y * CW + 4 * fl, ty[q]);
                st4(sG5 + y * CW + 4 * fl, tg5[q]);
                st4(sZ1 + y * CW + 4 * fl, tz1[q]);
                st4(sZ2 + y * CW + 4 * fl, tz2[q]);
            }
        }
    }
    __syncthreads();

    // streaming phase: lane-resident Y[c], G9[b,c], r[c]; one row of dP per iteration.  Branch-free like F1: ragged
    // lanes compute on clamped addresses and only their stores are predicated.
    f4 yv[NI], g9[NI];
    float rc[NI];
    int coff[NI];
    bool live[NI];
    const int fld = fok ? f : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * PPW + cg;
        const bool ok = FULL || (c < N);
        const int cc = ok ? c : 0;
        live[i] = ok && fok;
        coff[i] = cc * C + fld;
        rc[i] = L.r[cc];
        yv[i] = ld4(sY + cc * CW + 4 * fl);
        g9[i] = BS_LD(G + (rowbase + (size_t)b * N + cc) * (size_t)(kK * C) + 9 * C + fld);
    }
    const int ib = b / PPW, cgb = b % PPW;
    float *dPg = dP + pbase * C + (size_t)b * N * C;
    const size_t rowStride = (size_t)N * N * C;
    for (int a = wave; a < N; a += kWaves) {
        float *row = dPg + a * rowStride;
        f4 old[NI];
        if (ACC) {
#pragma unroll
            for (int i = 0; i < NI; ++i) old[i] = ld4(row + coff[i]);
        }
        const f4 xa = ld4(sX + a * CW + 4 * fl), g5a = ld4(sG5 + a * CW + 4 * fl);
        const float ra = L.r[a];
        const int ia = a / PPW, cga = a % PPW;
        const f4 z1 = ld4(sZ1 + a * CW + 4 * fl) * ((cg == cgb) ? 1.f : 0.f);
        const f4 z2 = ld4(sZ2 + a * CW + 4 * fl) * ((cg == cga) ? 1.f : 0.f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f4 o = xa + yv[i] + g5a * rc[i] + g9[i] * ra;
            if (i == ib) o += z1;  // wave-uniform conditions: scalar branches
            if (i == ia) o += z2;
            if (ACC) o += old[i];
            if (FULL || live[i]) BS_ST(row + coff[i], o);
        }
    }
}

template <int LPC>
static size_t bwd_slab_lds_bytes(int N) {
    constexpr int CW = 4 * LPC;
    return sizeof(float) * ((size_t)adj_lds_floats(N) + 5 * (size_t)N * CW + 12 * CW);
}

// ------------------------------------------------------------------------------------------------------------
// Generic kernels: any N, any C, any alignment.  One thread per table / output element; used for shapes the slab
// kernels do not cover and as an independent second implementation in the tests.
// ------------------------------------------------------------------------------------------------------------
// workspace per graph: Aplus[N*N], r[N], st[2] (tot, tr)
__global__ void r18_gen_adj(const float *__restrict__ A, float *__restrict__ Ap, float *__restrict__ r,
                            float *__restrict__ st, int N) {
    const int g = blockIdx.x;
    const float *Ag = A + (size_t)g * N * N;
    float *Apg = Ap + (size_t)g * N * N;
    __shared__ float sh[2];
    if (threadIdx.x == 0) sh[0] = sh[1] = 0.f;
    __syncthreads();
    for (int d = threadIdx.x; d < N; d += blockDim.x) {
        float s = 0.f;
        for (int e = 0; e < N; ++e) {
            float a = Ag[d * N + e];
            a = (a > 0.f) ? a : 0.f;
            Apg[d * N + e] = a;
            s += a;
        }
        r[(size_t)g * N + d] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, dg = 0.f;
        for (int d = 0; d < N; ++d) {
            t += r[(size_t)g * N + d];
            dg += Apg[d * N + d];
        }
        st[2 * g] = t;
        st[2 * g + 1] = dg;
    }
}

// tables[g][4][N][N][C]: S_ab, T6 (r-weighted over c), S_bc, T10 (r-weighted over a)
__global__ void r18_gen_tables(const float *__restrict__ P, const float *__restrict__ r, float *__restrict__ tab, int N,
                               int C, size_t total) {
    const size_t NNC = (size_t)N * N * C;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = idx % C;
        size_t t = idx / C;
        const int j = t % N;
        t /= N;
        const int i = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *rg = r + g * N;
        float sab = 0.f, t6 = 0.f, sbc = 0.f, t10 = 0.f;
        for (int s = 0; s < N; ++s) {
            const float pij = Pg[(((size_t)i * N + j) * N + s) * C + f];  // P[i][j][s]
            sab += pij;
            t6 += pij * rg[s];
            const float psij = Pg[(((size_t)s * N + i) * N + j) * C + f];  // P[s][i][j]
            sbc += psij;
            t10 += psij * rg[s];
        }
        float *tg = tab + g * 4 * NNC + ((size_t)i * N + j) * C + f;
        tg[0 * NNC] = sab;
        tg[1 * NNC] = t6;
        tg[2 * NNC] = sbc;
        tg[3 * NNC] = t10;
    }
}

__global__ void r18_gen_forward(const float *__restrict__ P, const float *__restrict__ Ap, const float *__restrict__ r,
                                const float *__restrict__ st, const float *__restrict__ tab, float *__restrict__ Out,
                                int N, int C, size_t total) {
    const size_t NNC = (size_t)N * N * C;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = idx % C;
        size_t t = idx / C;
        const int y = t % N;
        t /= N;
        const int x = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *Ag = Ap + g * N * N;
        const float *rg = r + g * N;
        const float tot = st[2 * g], tr = st[2 * g + 1];
        const float *Sab = tab + g * 4 * NNC, *T6 = Sab + NNC, *Sbc = T6 + NNC, *T10 = Sbc + NNC;
#define TAB(T, i, j) T[((size_t)(i) * N + (j)) * C + f]
#define PP(a, b, c) Pg[(((size_t)(a) * N + (b)) * N + (c)) * C + f]
        float rowsum = 0.f, colsum = 0.f, d8 = 0.f, d11 = 0.f, m8 = 0.f, m11 = 0.f, m12 = 0.f, m15 = 0.f, m16 = 0.f;
        for (int e = 0; e < N; ++e) {
            rowsum += TAB(Sab, x, e);
            colsum += TAB(Sab, e, x);
            d8 += PP(x, e, e);
            d11 += PP(e, x, e);
            const float w = Ag[y * N + e];
            m8 += w * TAB(Sab, x, e);
            m11 += w * TAB(Sab, e, x);
            m12 += w * TAB(Sbc, x, e);
            m15 += w * PP(x, e, e);
            m16 += w * PP(e, x, e);
        }
        float total_ = 0.f, s14 = 0.f, s15 = 0.f, s18 = 0.f;
        for (int i = 0; i < N; ++i) {
            s14 += TAB(Sab, i, i);
            s18 += PP(i, i, i);
            for (int j = 0; j < N; ++j) {
                total_ += TAB(Sab, i, j);
                s15 += PP(i, j, j);
            }
        }
        const float axy = Ag[x * N + y], ry = rg[y];
        float *o = Out + (((size_t)g * N + x) * N + y) * (size_t)(kK * C) + f;
        o[0 * C] = TAB(Sab, x, y) * tot;
        o[1 * C] = rowsum * ry;
        o[2 * C] = TAB(Sbc, x, y) * tot;
        o[3 * C] = colsum * ry;
        o[4 * C] = axy * total_;
        o[5 * C] = TAB(T6, x, y);
        o[6 * C] = TAB(Sab, x, y) * tr;
        o[7 * C] = d8 * ry;
        o[8 * C] = m8;
        o[9 * C] = TAB(T10, x, y);
        o[10 * C] = d11 * ry;
        o[11 * C] = m11;
        o[12 * C] = m12;
        o[13 * C] = axy * s14;
        o[14 * C] = axy * s15;
        o[15 * C] = m15;
        o[16 * C] = m16;
        o[17 * C] = axy * s18;
#undef TAB
#undef PP
    }
}

// scal[g][4][C]: u5, u14, u15, u18 = sum_{d,e} G_{4,13,14,17}[d,e] A+[d,e]
__global__ void r18_gen_bwd_scalars(const float *__restrict__ G, const float *__restrict__ Ap, float *__restrict__ scal,
                                    int N, int C, size_t total) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = idx % C;
        const int j = (idx / C) % 4;
        const size_t g = idx / (4 * (size_t)C);
        const int kk = (j == 0) ? 4 : (j == 1) ? 13 : (j == 2) ? 14 : 17;
        const float *Gg = G + g * (size_t)N * N * kK * C;
        const float *Ag = Ap + g * N * N;
        float s = 0.f;
        for (int d = 0; d < N; ++d)
            for (int e = 0; e < N; ++e) s += Gg[(((size_t)d * N + e) * kK + kk) * C + f] * Ag[d * N + e];
        scal[idx] = s;
    }
}

__global__ void r18_gen_backward(const float *__restrict__ G, const float *__restrict__ Ap, const float *__restrict__ r,
                                 const float *__restrict__ st, const float *__restrict__ scal, float *__restrict__ dP,
                                 int N, int C, size_t total, int accumulate) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = idx % C;
        size_t t = idx / C;
        const int c = t % N;
        t /= N;
        const int b = t % N;
        t /= N;
        const int a = t % N;
        const size_t g = t / N;
        const float *Gg = G + g * (size_t)N * N * kK * C;
        const float *Ag = Ap + g * N * N;
        const float *rg = r + g * N;
        const float tot = st[2 * g], tr = st[2 * g + 1];
        const float *sc = scal + g * 4 * (size_t)C;
#define GG(x, y, k) Gg[(((size_t)(x) * N + (y)) * kK + (k)) * C + f]
        float v = tot * GG(a, b, 0) + tr * GG(a, b, 6) + GG(a, b, 5) * rg[c] + tot * GG(b, c, 2) + GG(b, c, 9) * rg[a] +
                  sc[0 * C + f];
        float u2 = 0.f, u4 = 0.f, u8 = 0.f, u11 = 0.f, v9 = 0.f, v12 = 0.f, v13 = 0.f, v16 = 0.f, v17 = 0.f;
        for (int d = 0; d < N; ++d) {
            const float rd = rg[d];
            u2 += GG(a, d, 1) * rd;
            u4 += GG(b, d, 3) * rd;
            u8 += GG(a, d, 7) * rd;
            u11 += GG(b, d, 10) * rd;
            v9 += GG(a, d, 8) * Ag[d * N + b];
            v12 += GG(b, d, 11) * Ag[d * N + a];
            v13 += GG(b, d, 12) * Ag[d * N + c];
            v16 += GG(a, d, 15) * Ag[d * N + b];
            v17 += GG(b, d, 16) * Ag[d * N + a];
        }
        v += u2 + u4 + v9 + v12 + v13;
        if (b == c) v += u8 + v16 + sc[2 * C + f];
        if (a == c) v += u11 + v17;
        if (a == b) v += sc[1 * C + f];
        if (a == b && b == c) v += sc[3 * C + f];
#undef GG
        if (accumulate)
            dP[idx] += v;
        else
            dP[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct FastShape {
    int lpc, ni, nwin;
};

bool fast_shape(int N, int C, const void *p0, const void *p1, const void *p2, FastShape *fs) {
    if (C % 4 != 0) return false;
    if (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15u) return false;
    // lanes per position = the channel window a workgroup works on (4 lpc channels).  Beyond eight wave loads per slab row the window is
    // halved instead of leaving the slab kernels (round 4, second session): N <= 64 runs on 32-channel windows (128-byte pieces of a
    // position's record: still whole cache lines), N <= 128 on 16-channel windows -- the thread-per-element kernels that used to take
    // over at N > 32 (C = 64) moved 77 GB/s against 3.6 TB/s on the slabs (cfg2 shape at N = 40: 28 ms against 0.42 ms at N = 32).
    int lpc = (C <= 16) ? 4 : (C <= 32) ? 8 : 16;
    while (lpc > 4 && (N + 64 / lpc - 1) / (64 / lpc) > 8) lpc >>= 1;
    const int ppw = 64 / lpc;
    const int ni = (N + ppw - 1) / ppw;
    if (ni > 8) return false;
    fs->lpc = lpc;
    fs->ni = (ni <= 1) ? 1 : (ni <= 2) ? 2 : (ni <= 4) ? 4 : 8;
    fs->nwin = (C + 4 * lpc - 1) / (4 * lpc);
    return true;
}


// One launch set = `blocks` (node/graph, index) pairs x nwin channel windows; R says where each pair lives.
// smax = largest N among them (sizes the LDS); `uniform_full` enables the unmasked fast variant.
template <int LPC, int NI, bool FULL>
gf_status launch_fwd_slab(gf_ctx *ctx, const float *P, const float *A, float *Out, float *wsSab, float *wsDbb,
                          float *wsScal, const Ragged &R, int smax, int C, unsigned grid, int nwin) {
    const size_t l1 = fwd_slab_lds_bytes<LPC, NI>(smax);
    gf_status st = opt_in_lds(ctx, r18_fwd_slab<LPC, NI, FULL>, l1);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "r18_fwd_slab", (r18_fwd_slab<LPC, NI, FULL>), dim3(grid), dim3(kThreads), l1, P, A, Out, wsSab,
                       wsDbb, wsScal, R, C, nwin);
    return GF_OK;
}

template <int LPC, int NI>
gf_status launch_fwd(gf_ctx *ctx, const float *P, const float *A, float *Out, float *wsSab, float *wsDbb, float *wsScal,
                     Ragged R, long long blocks, int smax, int C, int nwin) {
    const unsigned grid = (unsigned)((size_t)blocks * nwin);
    const size_t l2 = fwd_rows_lds_bytes<LPC>(smax);
    const bool full = !R.pair_node && (smax == NI * (64 / LPC)) && (C % (4 * LPC) == 0);
    gf_status st = full ? launch_fwd_slab<LPC, NI, true>(ctx, P, A, Out, wsSab, wsDbb, wsScal, R, smax, C, grid, nwin)
                        : launch_fwd_slab<LPC, NI, false>(ctx, P, A, Out, wsSab, wsDbb, wsScal, R, smax, C, grid, nwin);
    if (st != GF_OK) return st;
    st = opt_in_lds(ctx, r18_fwd_rows<LPC>, l2);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "r18_fwd_rows", (r18_fwd_rows<LPC>), dim3(grid), dim3(kThreads), l2, A, Out, wsSab, wsDbb, wsScal, R,
                       C, nwin);
    return GF_OK;
}

template <int LPC, int NI, bool FULL, bool ACC>
gf_status launch_bwd_slab(gf_ctx *ctx, const float *G, const float *A, float *dP, float *wsWX, float *wsWZ,
                          float *wsPart, const Ragged &R, int smax, int C, unsigned grid, int nwin) {
    const size_t l2 = bwd_slab_lds_bytes<LPC>(smax);
    gf_status st = opt_in_lds(ctx, r18_bwd_slab<LPC, NI, FULL, ACC>, l2);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "r18_bwd_slab", (r18_bwd_slab<LPC, NI, FULL, ACC>), dim3(grid), dim3(kThreads), l2, G, A, dP, wsWX,
                       wsWZ, wsPart, R, C, nwin);
    return GF_OK;
}

template <int LPC, int NI>
gf_status launch_bwd(gf_ctx *ctx, const float *G, const float *A, float *dP, float *wsWX, float *wsWZ, float *wsPart,
                     Ragged R, long long blocks, int smax, int C, int nwin, int accumulate) {
    const unsigned grid = (unsigned)((size_t)blocks * nwin);
    const size_t l1 = bwd_rows_lds_bytes<LPC>(smax);
    gf_status st = opt_in_lds(ctx, r18_bwd_rows<LPC>, l1);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "r18_bwd_rows", (r18_bwd_rows<LPC>), dim3(grid), dim3(kThreads), l1, G, A, wsWX, wsWZ, wsPart, R, C,
                       nwin);
    const bool full = !R.pair_node && (smax == NI * (64 / LPC)) && (C % (4 * LPC) == 0);
    if (full)
        return accumulate ? launch_bwd_slab<LPC, NI, true, true>(ctx, G, A, dP, wsWX, wsWZ, wsPart, R, smax, C, grid, nwin)
                          : launch_bwd_slab<LPC, NI, true, false>(ctx, G, A, dP, wsWX, wsWZ, wsPart, R, smax, C, grid, nwin);
    return accumulate ? launch_bwd_slab<LPC, NI, false, true>(ctx, G, A, dP, wsWX, wsWZ, wsPart, R, smax, C, grid, nwin)
                      : launch_bwd_slab<LPC, NI, false, false>(ctx, G, A, dP, wsWX, wsWZ, wsPart, R, smax, C, grid, nwin);
}

#define GF_DISPATCH_NI(FN, LPC, ...)                     \
    switch (fs.ni) {                                     \
        case 1: return FN<LPC, 1>(__VA_ARGS__);          \
        case 2: return FN<LPC, 2>(__VA_ARGS__);          \
        case 4: return FN<LPC, 4>(__VA_ARGS__);          \
        default: return FN<LPC, 8>(__VA_ARGS__);         \
    }
#define GF_DISPATCH(FN, ...)                                           \
    switch (fs.lpc) {                                                  \
        case 4: GF_DISPATCH_NI(FN, 4, __VA_ARGS__)                     \
        case 8: GF_DISPATCH_NI(FN, 8, __VA_ARGS__)                     \
        default: GF_DISPATCH_NI(FN, 16, __VA_ARGS__)                   \
    }

Ragged uniform_batch(int N) {
    Ragged R = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, N};
    return R;
}

unsigned gen_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 65536 ? 65536 : (blocks == 0 ? 1 : blocks));
}

size_t fast_ws_floats(int N, int C, int batch) { return (size_t)batch * (2 * (size_t)N * N * C + 4 * (size_t)N * C); }
size_t gen_ws_floats(int N, int C, int batch) {
    return (size_t)batch * ((size_t)N * N + N + 2 + 4 * (size_t)N * N * C + 4 * (size_t)C) + 64;
}

}  // namespace

size_t r18_workspace_bytes(int N, int C, int batch) {
    const size_t a = fast_ws_floats(N, C, batch), b = gen_ws_floats(N, C, batch);
    return sizeof(float) * (a > b ? a : b) + 256;
}


gf_status r18_forward(gf_ctx *ctx, const float *P, const float *A, float *Out, int N, int C, int batch) {
    gf_status st = ensure_ws(ctx, r18_workspace_bytes(N, C, batch));
    if (st != GF_OK) return st;
    float *ws = static_cast<float *>(ctx->ws);
    FastShape fs;
    if (!ctx->r18_generic && fast_shape(N, C, P, Out, nullptr, &fs)) {
        const size_t nnc = (size_t)batch * N * N * C;
        float *wsSab = ws, *wsDbb = ws + nnc, *wsScal = ws + 2 * nnc;
        GF_DISPATCH(launch_fwd, ctx, P, A, Out, wsSab, wsDbb, wsScal, uniform_batch(N), (long long)batch * N, N, C, fs.nwin)
    }
    float *Ap = ws;
    float *r = Ap + (size_t)batch * N * N;
    float *stv = r + (size_t)batch * N;
    float *tab = stv + align_up((size_t)batch * 2, 4);
    GF_LAUNCH(ctx, "r18_gen_adj", r18_gen_adj, dim3(batch), dim3(64), 0, A, Ap, r, stv, N);
    const size_t total = (size_t)batch * N * N * C;
    GF_LAUNCH(ctx, "r18_gen_tables", r18_gen_tables, dim3(gen_grid(total)), dim3(256), 0, P, r, tab, N, C, total);
    GF_LAUNCH(ctx, "r18_gen_forward", r18_gen_forward, dim3(gen_grid(total)), dim3(256), 0, P, Ap, r, stv, tab, Out, N, C,
                       total);
    return GF_OK;
}

gf_status r18_backward(gf_ctx *ctx, const float *G, const float *A, float *dP, int N, int C, int batch, int accumulate) {
    gf_status st = ensure_ws(ctx, r18_workspace_bytes(N, C, batch));
    if (st != GF_OK) return st;
    float *ws = static_cast<float *>(ctx->ws);
    FastShape fs;
    if (!ctx->r18_generic && fast_shape(N, C, G, dP, nullptr, &fs)) {
        const size_t nnc = (size_t)batch * N * N * C;
        float *wsWX = ws, *wsWZ = ws + nnc, *wsPart = ws + 2 * nnc;
        GF_DISPATCH(launch_bwd, ctx, G, A, dP, wsWX, wsWZ, wsPart, uniform_batch(N), (long long)batch * N, N, C, fs.nwin, accumulate)
    }
    float *Ap = ws;
    float *r = Ap + (size_t)batch * N * N;
    float *stv = r + (size_t)batch * N;
    float *scal = stv + align_up((size_t)batch * 2, 4);
    GF_LAUNCH(ctx, "r18_gen_adj", r18_gen_adj, dim3(batch), dim3(64), 0, A, Ap, r, stv, N);
    const size_t nsc = (size_t)batch * 4 * C;
    GF_LAUNCH(ctx, "r18_gen_bwd_scalars", r18_gen_bwd_scalars, dim3(gen_grid(nsc)), dim3(256), 0, G, Ap, scal, N, C, nsc);
    const size_t total = (size_t)batch * N * N * N * C;
    GF_LAUNCH(ctx, "r18_gen_backward", r18_gen_backward, dim3(gen_grid(total)), dim3(256), 0, G, Ap, r, stv, scal, dP, N, C,
                       total, accumulate);
    return GF_OK;
}


// ---- ragged batches (SMP driver): nodes of different sizes in one launch ------------------------------------------
// Pairs [pair_lo, pair_hi) are (node, index) workgroups; all their nodes have size <= smax <= 8 * 64/LPC.
// Workspace: two [total_rows][C] tables + [total_pairs][4][C] partial scalars, addressed with the node offsets.
bool r18_ragged_supported(int smax, int C, const void *p0, const void *p1) {
    FastShape fs;
    return fast_shape(smax, C, p0, p1, nullptr, &fs);
}

size_t r18_ragged_workspace_bytes(long long total_rows, long long total_pairs, int C) {
    return sizeof(float) * (2 * (size_t)total_rows * C + 4 * (size_t)total_pairs * C) + 256;
}

static Ragged ragged_of(const gf_ragged_nodes &t, long long pair_lo, int smax) {
    Ragged R = {t.pair_node, t.node_s, t.node_p, t.node_row, t.node_pair, pair_lo, smax};
    return R;
}

gf_status r18_forward_ragged(gf_ctx *ctx, const float *P, const float *A, float *Out, const gf_ragged_nodes &t,
                             long long pair_lo, long long pair_hi, int smax, int C) {
    if (pair_hi <= pair_lo) return GF_OK;
    gf_status st = ensure_ws(ctx, r18_ragged_workspace_bytes(t.total_rows, t.total_pairs, C));
    if (st != GF_OK) return st;
    FastShape fs;
    if (!fast_shape(smax, C, P, Out, nullptr, &fs)) return fail(ctx, GF_ERR_UNSUPPORTED, "ragged r18: shape outside the slab kernels (smax=%d C=%d)", smax, C);
    float *ws = static_cast<float *>(ctx->ws);
    const size_t nnc = (size_t)t.total_rows * C;
    float *wsSab = ws, *wsDbb = ws + nnc, *wsScal = ws + 2 * nnc;
    GF_DISPATCH(launch_fwd, ctx, P, A, Out, wsSab, wsDbb, wsScal, ragged_of(t, pair_lo, smax), pair_hi - pair_lo, smax, C, fs.nwin)
}

gf_status r18_backward_ragged(gf_ctx *ctx, const float *G, const float *A, float *dP, const gf_ragged_nodes &t,
                              long long pair_lo, long long pair_hi, int smax, int C, int accumulate) {
    if (pair_hi <= pair_lo) return GF_OK;
    gf_status st = ensure_ws(ctx, r18_ragged_workspace_bytes(t.total_rows, t.total_pairs, C));
    if (st != GF_OK) return st;
    FastShape fs;
    if (!fast_shape(smax, C, G, dP, nullptr, &fs)) return fail(ctx, GF_ERR_UNSUPPORTED, "ragged r18: shape outside the slab kernels (smax=%d C=%d)", smax, C);
    float *ws = static_cast<float *>(ctx->ws);
    const size_t nnc = (size_t)t.total_rows * C;
    float *wsWX = ws, *wsWZ = ws + nnc, *wsPart = ws + 2 * nnc;
    GF_DISPATCH(launch_bwd, ctx, G, A, dP, wsWX, wsWZ, wsPart, ragged_of(t, pair_lo, smax), pair_hi - pair_lo, smax, C, fs.nwin, accumulate)
}

}  // namespace gf
