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
#include "gf_internal.h"

namespace gf {
namespace {

#define GRID_STRIDE(idx, total) \
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

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

template <int K>
__global__ void fam_tables(const float *__restrict__ P, const float *__restrict__ adjs, float *__restrict__ tab, int N,
                           int C, size_t total) {
    const size_t NNC = (size_t)N * N * C;
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int j = t % N;
        t /= N;
        const int i = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *dg = q + N;
        float acc[kNTab];
#pragma unroll
        for (int k = 0; k < kNTab; ++k) acc[k] = 0.f;
        for (int s = 0; s < N; ++s) {
            const float rs = r[s], qs = q[s], ds = dg[s];
            const float pab = Pg[(((size_t)i * N + j) * N + s) * C + f];  // P[i][j][s]
            const float pac = Pg[(((size_t)i * N + s) * N + j) * C + f];  // P[i][s][j]
            const float pbc = Pg[(((size_t)s * N + i) * N + j) * C + f];  // P[s][i][j]
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
        for (int k = 0; k < (K == 50 ? kNTab : 3); ++k) tg[k * NNC] = acc[k];
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

#define OUTC(c, expr)                                \
    do {                                             \
        if (slot<K>(c) >= 0) o[slot<K>(c) * C] = (expr); \
    } while (0)

template <int K>
__global__ void fam_forward(const float *__restrict__ P, const float *__restrict__ A, const float *__restrict__ adjs,
                            const float *__restrict__ tab, const float *__restrict__ vec, const float *__restrict__ sc,
                            float *__restrict__ Out, int N, int C, size_t total) {
    const size_t NNC = (size_t)N * N * C, NC = (size_t)N * C;
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int y = t % N;
        t /= N;
        const int x = t % N;
        const size_t g = t / N;
        const float *Pg = P + g * NNC * N;
        const float *Ag = A + g * N * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
        const float tot = st[0], tr = st[1];
        const float *T = tab + g * kNTab * NNC;
        const float *v = vec + g * kNVec * NC + f;
        const float *s = sc + g * kNSc * (size_t)C + f;
#define TB(k, i, j) T[(k)*NNC + ((size_t)(i) * N + (j)) * C + f]
#define PP(a, b, c) Pg[(((size_t)(a) * N + (b)) * N + (c)) * C + f]
        float *o = Out + (((size_t)g * N + x) * N + y) * (size_t)(K * C) + f;
        const float ry = r[y], qy = q[y], axy = Ag[x * N + y];
        // "1+1+1"
        OUTC(1, TB(0, x, y) * tot);
        OUTC(2, TB(1, x, y) * tot);
        OUTC(3, v[0 * NC + (size_t)x * C] * ry);
        OUTC(4, v[0 * NC + (size_t)x * C] * qy);
        OUTC(5, TB(2, x, y) * tot);
        OUTC(6, v[1 * NC + (size_t)x * C] * ry);
        OUTC(7, v[1 * NC + (size_t)x * C] * qy);
        OUTC(8, v[2 * NC + (size_t)x * C] * ry);
        OUTC(9, v[2 * NC + (size_t)x * C] * qy);
        OUTC(10, s[0 * C] * axy);
        if (K == 50) {
            // "1+2" that are plain table reads or outer products
            OUTC(11, TB(3, x, y));
            OUTC(12, TB(4, x, y));
            OUTC(13, TB(0, x, y) * tr);
            OUTC(14, TB(6, x, y));
            OUTC(15, TB(7, x, y));
            OUTC(16, TB(1, x, y) * tr);
            OUTC(17, v[3 * NC + (size_t)x * C] * ry);
            OUTC(20, v[3 * NC + (size_t)x * C] * qy);
            OUTC(23, TB(9, x, y));
            OUTC(24, TB(10, x, y));
            OUTC(25, TB(2, x, y) * tr);
            OUTC(26, v[4 * NC + (size_t)x * C] * ry);
            OUTC(29, v[4 * NC + (size_t)x * C] * qy);
            OUTC(32, v[5 * NC + (size_t)x * C] * ry);
            OUTC(35, v[5 * NC + (size_t)x * C] * qy);
            OUTC(38, s[1 * C] * axy);
            OUTC(39, s[2 * C] * axy);
            OUTC(40, s[3 * C] * axy);
            OUTC(41, TB(5, x, y));
            OUTC(42, TB(8, x, y));
            OUTC(45, TB(11, x, y));
            OUTC(50, s[4 * C] * axy);
            // N x N products with A: one pass over the contracted index
            float m18 = 0, m19 = 0, m21 = 0, m22 = 0, m27 = 0, m28 = 0, m30 = 0, m31 = 0, m33 = 0, m34 = 0, m36 = 0,
                  m37 = 0, m43 = 0, m44 = 0, m46 = 0, m47 = 0, m48 = 0, m49 = 0;
            for (int z = 0; z < N; ++z) {
                const float ayz = Ag[y * N + z], azy = Ag[z * N + y];
                const float sab_xz = TB(0, x, z), sab_zx = TB(0, z, x);
                const float sac_xz = TB(1, x, z), sac_zx = TB(1, z, x);
                const float sbc_xz = TB(2, x, z), sbc_zx = TB(2, z, x);
                const float pxzz = PP(x, z, z), pzxz = PP(z, x, z), pzzx = PP(z, z, x);
                m18 += sab_xz * ayz;  // (a,d) tie(b,e): sum_b S_ab[a,b] A[d,b]
                m19 += sac_xz * ayz;  // (a,d) tie(c,e)
                m21 += sab_xz * azy;  // (a,e) tie(b,d): sum_b S_ab[a,b] A[b,e]
                m22 += sac_xz * azy;  // (a,e) tie(c,d)
                m27 += sab_zx * ayz;  // (b,d) tie(a,e): sum_a S_ab[a,b] A[d,a]
                m28 += sbc_xz * ayz;  // (b,d) tie(c,e)
                m30 += sab_zx * azy;  // (b,e) tie(a,d)
                m31 += sbc_xz * azy;  // (b,e) tie(c,d)
                m33 += sac_zx * ayz;  // (c,d) tie(a,e): sum_a S_ac[a,c] A[d,a]
                m34 += sbc_zx * ayz;  // (c,d) tie(b,e)
                m36 += sac_zx * azy;  // (c,e) tie(a,d)
                m37 += sbc_zx * azy;  // (c,e) tie(b,d)
                m43 += pxzz * ayz;    // (a,d) b=c=e
                m44 += pxzz * azy;    // (a,e) b=c=d
                m46 += pzxz * ayz;    // (b,d) a=c=e
                m47 += pzxz * azy;    // (b,e) a=c=d
                m48 += pzzx * ayz;    // (c,d) a=b=e
                m49 += pzzx * azy;    // (c,e) a=b=d
            }
            OUTC(18, m18);
            OUTC(19, m19);
            OUTC(21, m21);
            OUTC(22, m22);
            OUTC(27, m27);
            OUTC(28, m28);
            OUTC(30, m30);
            OUTC(31, m31);
            OUTC(33, m33);
            OUTC(34, m34);
            OUTC(36, m36);
            OUTC(37, m37);
            OUTC(43, m43);
            OUTC(44, m44);
            OUTC(46, m46);
            OUTC(47, m47);
            OUTC(48, m48);
            OUTC(49, m49);
        }
#undef TB
#undef PP
    }
}
#undef OUTC

// backward scalars, bsc[g][5][C]: u10, u38, u39, u40, u50 = sum_{d,e} G_c[d,e] A[d,e]
template <int K>
__global__ void fam_bwd_scalars(const float *__restrict__ G, const float *__restrict__ A, float *__restrict__ bsc, int N,
                                int C, size_t total) {
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        const int j = (idx / C) % 5;
        const size_t g = idx / (5 * (size_t)C);
        const int cs = (j == 0) ? 10 : (j == 1) ? 38 : (j == 2) ? 39 : (j == 3) ? 40 : 50;
        float sum = 0.f;
        if (slot<K>(cs) >= 0) {
            const float *Gg = G + g * (size_t)N * N * K * C + (size_t)slot<K>(cs) * C + f;
            const float *Ag = A + g * N * N;
            for (int de = 0; de < N * N; ++de) sum += Gg[(size_t)de * K * C] * Ag[de];
        }
        bsc[idx] = sum;
    }
}

// backward pair tables, btab[g][6][N][N][C]:
//   0 X_ab[a,b]  1 X_ac[a,c]  2 X_bc[b,c]  3 Z_bc[a,b] (applies at b==c)  4 Z_ac[b,a] (a==c)  5 Z_ab[c,a] (a==b)
constexpr int kNBTab = 6;

template <int K>
__global__ void fam_bwd_tables(const float *__restrict__ G, const float *__restrict__ A, const float *__restrict__ adjs,
                               const float *__restrict__ bsc, float *__restrict__ btab, int N, int C, size_t total) {
    const size_t NNC = (size_t)N * N * C;
    GRID_STRIDE(idx, total) {
        const int f = idx % C;
        size_t t = idx / C;
        const int j = t % N;
        t /= N;
        const int i = t % N;
        const size_t g = t / N;
        const float *Gg = G + g * (size_t)N * N * K * C + f;
        const float *Ag = A + g * N * N;
        const float *r = adjs + g * adjs_stride(N), *q = r + N, *st = q + 2 * N;
        const float tot = st[0], tr = st[1];
        const float *u = bsc + g * 5 * (size_t)C + f;
#define GC(c, x, y) (slot<K>(c) >= 0 ? Gg[(((size_t)(x) * N + (y)) * K + slot<K>(c)) * C] : 0.f)
        float xab = tot * GC(1, i, j) + u[0 * C];
        float xac = tot * GC(2, i, j);
        float xbc = tot * GC(5, i, j);
        float zbc = 0.f, zac = 0.f, zab = 0.f;
        if (K == 50) {
            xab += tr * GC(13, i, j);
            xac += tr * GC(16, i, j);
            xbc += tr * GC(25, i, j);
            zbc = u[3 * C];  // u40
            zac = u[2 * C];  // u39
            zab = u[1 * C];  // u38
        }
        for (int z = 0; z < N; ++z) {
            const float rz = r[z], qz = q[z];
            // outer-product cases: X_ab[a=i,b=j] takes U3[a]+U4[a]+U6[b]+U7[b]; X_ac[a=i,c=j] takes U8[c]+U9[c]
            xab += GC(3, i, z) * rz + GC(4, i, z) * qz + GC(6, j, z) * rz + GC(7, j, z) * qz;
            xac += GC(8, j, z) * rz + GC(9, j, z) * qz;
            if (K == 50) {
                const float azj = Ag[z * N + j], ajz = Ag[j * N + z], azi = Ag[z * N + i], aiz = Ag[i * N + z];
                // X_ab[a=i, b=j]
                xab += GC(18, i, z) * azj + GC(21, i, z) * ajz + GC(27, j, z) * azi + GC(30, j, z) * aiz;
                // X_ac[a=i, c=j]
                xac += GC(19, i, z) * azj + GC(22, i, z) * ajz + GC(33, j, z) * azi + GC(36, j, z) * aiz;
                // X_bc[b=i, c=j]
                xbc += GC(28, i, z) * azj + GC(31, i, z) * ajz + GC(34, j, z) * azi + GC(37, j, z) * aiz;
                // Z_bc[a=i, b=j]  (b == c)
                zbc += GC(17, i, z) * rz + GC(20, i, z) * qz + GC(43, i, z) * azj + GC(44, i, z) * ajz;
                // Z_ac[b=i, a=j]  (a == c)
                zac += GC(26, i, z) * rz + GC(29, i, z) * qz + GC(46, i, z) * azj + GC(47, i, z) * ajz;
                // Z_ab[c=i, a=j]  (a == b)
                zab += GC(32, i, z) * rz + GC(35, i, z) * qz + GC(48, i, z) * azj + GC(49, i, z) * ajz;
            }
        }
#undef GC
        float *bt = btab + g * kNBTab * NNC + ((size_t)i * N + j) * C + f;
        bt[0 * NNC] = xab;
        bt[1 * NNC] = xac;
        bt[2 * NNC] = xbc;
        if (K == 50) {
            bt[3 * NNC] = zbc;
            bt[4 * NNC] = zac;
            bt[5 * NNC] = zab;
        }
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
    GF_LAUNCH(ctx, "fam_tables", fam_tables<K>, dim3(grid_for(nn)), dim3(256), 0, P, w.adjs, w.tab, N, C, nn);
    GF_LAUNCH(ctx, "fam_vectors", fam_vectors, dim3(grid_for(nv)), dim3(256), 0, P, w.tab, w.vec, w.sc, N, C, nv);
    GF_LAUNCH(ctx, "fam_scalars", fam_scalars, dim3(grid_for(ns)), dim3(256), 0, P, w.vec, w.sc, N, C, ns);
    GF_LAUNCH(ctx, "fam_forward", fam_forward<K>, dim3(grid_for(nn)), dim3(256), 0, P, A, w.adjs, w.tab, w.vec, w.sc, Out,
              N, C, nn);
    return GF_OK;
}

template <int K>
gf_status fam_backward_launch(gf_ctx *ctx, const float *G, const float *A, float *dP, int N, int C, int batch,
                              int accumulate) {
    gf_status st = ensure_ws(ctx, sizeof(float) * fam_ws_floats(N, C, batch) + 256);
    if (st != GF_OK) return st;
    const FamWs w = carve(static_cast<float *>(ctx->ws), N, C, batch);  // tab doubles as btab, sc as bsc
    const size_t nn = (size_t)batch * N * N * C, ns = (size_t)batch * 5 * C, np = nn * N;
    GF_LAUNCH(ctx, "fam_adj", fam_adj, dim3(batch), dim3(64), 0, A, w.adjs, N);
    GF_LAUNCH(ctx, "fam_bwd_scalars", fam_bwd_scalars<K>, dim3(grid_for(ns)), dim3(256), 0, G, A, w.sc, N, C, ns);
    GF_LAUNCH(ctx, "fam_bwd_tables", fam_bwd_tables<K>, dim3(grid_for(nn)), dim3(256), 0, G, A, w.adjs, w.sc, w.tab, N, C,
              nn);
    GF_LAUNCH(ctx, "fam_backward", fam_backward<K>, dim3(grid_for(np)), dim3(256), 0, G, A, w.adjs, w.sc, w.tab, dP, N, C,
              np, accumulate);
    return GF_OK;
}

}  // namespace

size_t family_workspace_bytes(int K, int N, int C, int batch) {
    if (K == 4) return 0;
    return sizeof(float) * fam_ws_floats(N, C, batch) + 256;
}

gf_status family_forward(gf_ctx *ctx, int K, const float *P, const float *A, float *Out, int N, int C, int batch) {
    if (K == 4) {
        const size_t total = (size_t)batch * N * N * C;
        GF_LAUNCH(ctx, "r4_forward", r4_forward, dim3(grid_for(total)), dim3(256), 0, P, Out, N, C, total);
        return GF_OK;
    }
    if (K == 10) return fam_forward_launch<10>(ctx, P, A, Out, N, C, batch);
    if (K == 50) return fam_forward_launch<50>(ctx, P, A, Out, N, C, batch);
    return fail(ctx, GF_ERR_INVALID, "family_forward: K=%d", K);
}

gf_status family_backward(gf_ctx *ctx, int K, const float *G, const float *A, float *dP, int N, int C, int batch,
                          int accumulate) {
    if (K == 4) {
        const size_t total = (size_t)batch * N * N * N * C;
        GF_LAUNCH(ctx, "r4_backward", r4_backward, dim3(grid_for(total)), dim3(256), 0, G, dP, N, C, total, accumulate);
        return GF_OK;
    }
    if (K == 10) return fam_backward_launch<10>(ctx, G, A, dP, N, C, batch, accumulate);
    if (K == 50) return fam_backward_launch<50>(ctx, G, A, dP, N, C, batch, accumulate);
    return fail(ctx, GF_ERR_INVALID, "family_backward: K=%d", K);
}

}  // namespace gf
