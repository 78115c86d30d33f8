// smp.hip -- batched SMP_omega (second-order CCN) forward/backward on the device.
//
// Reproduces the op DAG that SMP_omega::complete_computation_graph builds per molecule
// (GraphFlow/SMP_omega.h:607-692) for a whole BATCH of molecules at once:
//   level 0   f_0[v] = LeakyReLU(H x_v)                                         (:617-626)
//   level l   T_w = X_vw f_{l-1}[w] X_vw^T  for w in phi_l(v)   -> index gather (MatTensorMul + TensorMatMul, :641-645)
//             P = stack_w T_w ; Q = RisiContraction_18(P, A_v)                  (:647-651)
//             f_l[v] = LeakyReLU(reshape(Q)[s^2,18C] K_l + b_l)                 (:654-669)
//   readout   g = sum_v LeakyReLU(sum_ij f_L[v]) ; y = <g, W> ; loss = (y - t)^2 / 2   (:676-692)
// and the reverse sweep (GraphFlow.h:729: reverse insertion order; every op `+=` into its inputs).
// The X matrices are 0/1 selections, so X F X^T is F[pi(i), pi(j)] or 0 (verified exact in the tests): the promotion
// is an index gather forward and a deterministic consumer-list gather backward (no atomics).
// Nodes of a level are bucketed by receptive-field size; each bucket is one uniform-N contraction launch and all
// buckets share one tall K-projection GEMM per level.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "smp_internal.h"
#include "r18_device.h"

namespace gf {
namespace {
using namespace dev;

constexpr float kAlpha = 0.01f;  // LeakyReLU3D.h:41, LeakyReLU.h default

#define GRID_STRIDE(idx, total) \
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

unsigned grid_for(size_t total, int per_block = 256) {
    size_t blocks = (total + per_block - 1) / per_block;
    return (unsigned)(blocks > 1048576 ? 1048576 : (blocks == 0 ? 1 : blocks));
}

__device__ __forceinline__ float lrelu(float z) { return z > 0.f ? z : kAlpha * z; }

// ---- promotion: P[n][a][b][c][:] = f_prev[src(n,a)][pi(b)][pi(c)][:] or 0 -------------------------------------------
// one workgroup per (node, neighbour) pair; threads run over (b, c, channel) with the channel fastest (coalesced)
__global__ void promote_forward(const float *__restrict__ fprev, float *__restrict__ P, const int *__restrict__ node_s,
                                const long long *__restrict__ node_row, const long long *__restrict__ node_p,
                                const long long *__restrict__ node_pair, const int *__restrict__ pair_node,
                                const long long *__restrict__ pair_src_row, const int *__restrict__ pair_src_s,
                                const short *__restrict__ pi, int C) {
    const long long e = blockIdx.x;
    const int n = pair_node[e];
    const int s = node_s[n], a = (int)(e - node_pair[n]), sw = pair_src_s[e];
    const short *map = pi + node_row[n] + (long long)a * s;
    const float *src = fprev + pair_src_row[e] * C;
    float *dst = P + (node_p[n] + (long long)a * s * s) * C;
    if ((C & 3) == 0) {  // 16 B per lane over the channel axis; (c, quad) flattened per row b
        const int Q = C >> 2, per_row = s * Q;
        for (int b = 0; b < s; ++b) {
            const int pb = map[b];
            float4 *drow = reinterpret_cast<float4 *>(dst + (size_t)b * s * C);
            for (int i = threadIdx.x; i < per_row; i += blockDim.x) {
                const int c = i / Q, q = i - c * Q;
                const int pc = map[c];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pb >= 0 && pc >= 0) v = reinterpret_cast<const float4 *>(src + ((size_t)pb * sw + pc) * C)[q];
                drow[i] = v;
            }
        }
        return;
    }
    const int total = s * s * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int f = i % C, bc = i / C;
        const int b = bc / s, c = bc - b * s;
        const int pb = map[b], pc = map[c];
        dst[i] = (pb >= 0 && pc >= 0) ? src[((size_t)pb * sw + pc) * C + f] : 0.f;
    }
}

// backward: df_prev[w][p][q][:] = sum over consumers (n,a) of dP[n][a][inv(p)][inv(q)][:]
constexpr int kPromoteChunk = 64, kPromoteMaxS = 32;

__global__ void promote_backward(const float *__restrict__ dP, float *__restrict__ dfprev,
                                 const int *__restrict__ prev_s, const long long *__restrict__ prev_row,
                                 const long long *__restrict__ cons_ptr, const long long *__restrict__ cons_slab,
                                 const int *__restrict__ cons_s, const long long *__restrict__ cons_inv_off,
                                 const short *__restrict__ inv, int C,
                                 // compact-diagonal variant of the fused level (smp_fused.hip), else null: the gradients of
                                 // f[w][p,p] and f[w][p,c_w] arrive as dFdc[node_pair[w] + p] = [ .. | .. ]
                                 const float *__restrict__ dFdc, const long long *__restrict__ prev_pair,
                                 const int *__restrict__ prev_center) {
    const int w = blockIdx.x;
    const int sw = prev_s[w];
    const float *dfd = dFdc ? dFdc + (size_t)prev_pair[w] * 2 * C : nullptr;
    const int cw = dFdc ? prev_center[w] : -1;
    auto diag_terms = [&](int p, int q, int q4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dfd) {
            if (p == q) {
                const float4 v = reinterpret_cast<const float4 *>(dfd + (size_t)p * 2 * C)[q4];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            if (q == cw) {
                const float4 v = reinterpret_cast<const float4 *>(dfd + (size_t)p * 2 * C + C)[q4];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
        }
        return t;
    };
    float *dst = dfprev + prev_row[w] * C;
    const long long c0 = cons_ptr[w], c1 = cons_ptr[w + 1];
    if ((C & 3) == 0 && sw <= kPromoteMaxS) {
        // consumer tables of this source node in LDS (chunks of kPromoteChunk consumers), then four consumers' loads in
        // flight per item: clamped address + 0/1 weight instead of a branch around every load.  Same summation order.
        __shared__ long long sSlab[kPromoteChunk];
        __shared__ int sS[kPromoteChunk];
        __shared__ short sInv[kPromoteChunk][kPromoteMaxS];
        const int Q = C >> 2, total4 = sw * sw * Q;
        const int npass = (total4 + (int)blockDim.x - 1) / (int)blockDim.x;
        for (long long cb = c0; cb < c1 || cb == c0; cb += kPromoteChunk) {
            const int nc = (int)((c1 - cb < kPromoteChunk) ? c1 - cb : kPromoteChunk);
            __syncthreads();
            for (int i = threadIdx.x; i < nc; i += blockDim.x) {
                sSlab[i] = cons_slab[cb + i];
                sS[i] = cons_s[cb + i];
            }
            for (int i = threadIdx.x; i < nc * sw; i += blockDim.x) sInv[i / sw][i % sw] = inv[cons_inv_off[cb + i / sw] + i % sw];
            __syncthreads();
            for (int pass = 0; pass < npass; ++pass) {
                const int i = pass * (int)blockDim.x + threadIdx.x;
                if (i >= total4) break;
                const int q4 = i % Q, pq = i / Q;
                const int p = pq / sw, q = pq - p * sw;
                float4 acc = (cb == c0) ? diag_terms(p, q, q4) : reinterpret_cast<float4 *>(dst)[i];
                for (int e0 = 0; e0 < nc; e0 += 4) {
                    float4 v[4];
                    float m[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = (e0 + j < nc) ? e0 + j : e0;
                        const int ib = sInv[e][p], ic = sInv[e][q];
                        const bool ok = e0 + j < nc && ib >= 0 && ic >= 0;
                        const long long pos = sSlab[e] + (ok ? (long long)ib * sS[e] + ic : 0);
                        v[j] = reinterpret_cast<const float4 *>(dP + pos * C)[q4];
                        m[j] = ok ? 1.f : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (m[j] != 0.f) {  // (select, not a multiply: a structurally-zero position holds stale data)
                            acc.x += v[j].x;
                            acc.y += v[j].y;
                            acc.z += v[j].z;
                            acc.w += v[j].w;
                        }
                    }
                }
                reinterpret_cast<float4 *>(dst)[i] = acc;
            }
            if (c1 == c0) break;
        }
        return;
    }
    if ((C & 3) == 0) {  // larger receptive fields than the LDS tables hold
        const int Q = C >> 2, total4 = sw * sw * Q;
        for (int i = threadIdx.x; i < total4; i += blockDim.x) {
            const int q4 = i % Q, pq = i / Q;
            const int p = pq / sw, q = pq - p * sw;
            float4 acc = diag_terms(p, q, q4);
            for (long long e = c0; e < c1; ++e) {
                const short *iv = inv + cons_inv_off[e];
                const int ib = iv[p], ic = iv[q];
                if (ib >= 0 && ic >= 0) {
                    const float4 v = reinterpret_cast<const float4 *>(dP + (cons_slab[e] + (long long)ib * cons_s[e] + ic) * C)[q4];
                    acc.x += v.x;
                    acc.y += v.y;
                    acc.z += v.z;
                    acc.w += v.w;
                }
            }
            reinterpret_cast<float4 *>(dst)[i] = acc;
        }
        return;
    }
    const int total = sw * sw * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int f = i % C, pq = i / C;
        const int p = pq / sw, q = pq - p * sw;
        float acc = 0.f;
        for (long long e = c0; e < c1; ++e) {
            const short *iv = inv + cons_inv_off[e];
            const int ib = iv[p], ic = iv[q];
            if (ib >= 0 && ic >= 0) acc += dP[(cons_slab[e] + (long long)ib * cons_s[e] + ic) * C + f];
        }
        dst[i] = acc;
    }
}

// ---- bias + LeakyReLU ---------------------------------------------------------------------------------------------
__global__ void bias_lrelu_forward(float *__restrict__ Y, const float *__restrict__ b, int C, size_t total) {
    GRID_STRIDE(i, total) Y[i] = lrelu(Y[i] + (b ? b[i % C] : 0.f));
}

// dZ = dF * lrelu'(z), decided from the sign of the stored activation (LeakyReLU is monotone: f > 0 <=> z > 0);
// also leaves per-block partial column sums of dZ for the bias gradient (VectorAddTensor.h:61-72)
__global__ void lrelu_backward_colsum(const float *__restrict__ F, float *__restrict__ dF, float *__restrict__ part, int C,
                                      long long rows, int rows_per_block) {
    __shared__ float red[256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    const int lanes = (C < 256) ? C : 256;          // channel lanes per row
    const int rl = 256 / lanes;                     // rows handled concurrently
    const int f0 = threadIdx.x % lanes, rr = threadIdx.x / lanes;
    for (int fb = 0; fb < C; fb += lanes) {
        const int f = fb + f0;
        float s = 0.f;
        if (f < C && rr < rl)
            for (long long r = r0 + rr; r < r1; r += rl) {
                const size_t i = (size_t)r * C + f;
                const float d = dF[i] * (F[i] > 0.f ? 1.f : kAlpha);
                dF[i] = d;
                s += d;
            }
        red[threadIdx.x] = s;
        __syncthreads();
        if (rr == 0 && f < C) {
            float t = 0.f;
            for (int k = 0; k < rl; ++k) t += red[k * lanes + f0];
            part[(size_t)blockIdx.x * C + f] = t;
        }
        __syncthreads();
    }
}

// out[f] += sum_b part[b][f]; one workgroup, row lanes in parallel then a fixed-order fold
__global__ void colsum_finish(const float *__restrict__ part, float *__restrict__ out, int C, int nblocks) {
    __shared__ float red[256];
    const int lanes = (C < 256) ? C : 256, rl = 256 / lanes;
    const int f0 = threadIdx.x % lanes, rr = threadIdx.x / lanes;
    for (int fb = 0; fb < C; fb += lanes) {
        const int f = fb + f0;
        float s = 0.f;
        if (f < C && rr < rl)
            for (int b = rr; b < nblocks; b += rl) s += part[(size_t)b * C + f];
        red[threadIdx.x] = s;
        __syncthreads();
        if (rr == 0 && f < C) {
            float t = 0.f;
            for (int k = 0; k < rl; ++k) t += red[k * lanes + f0];
            out[f] += t;
        }
        __syncthreads();
    }
}

// ---- readout ----------------------------------------------------------------------------------------------------------
// sh[n][:] = sum_{ij} f_L[n][i][j][:]  (ShrinkTensor.h:37-50), vf = LeakyReLU(sh)
__global__ void readout_nodes(const float *__restrict__ fL, const int *__restrict__ node_s,
                              const long long *__restrict__ node_row, float *__restrict__ sh, float *__restrict__ vf,
                              int C, size_t total) {
    GRID_STRIDE(i, total) {
        const int f = i % C;
        const size_t n = i / C;
        const int s = node_s[n];
        const float *src = fL + node_row[n] * C + f;
        float acc = 0.f;
        for (int r = 0; r < s * s; ++r) acc += src[(size_t)r * C];
        sh[i] = acc;
        vf[i] = lrelu(acc);
    }
}

// C % 4 == 0 and C <= 1024: workgroup per node, 256 threads = (256 / (C/4)) row groups x C/4 float4 lanes; every group
// sums rows g, g+ng, ... with batched loads, the groups are folded through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void readout_nodes_v(const float *__restrict__ fL, const int *__restrict__ node_s,
                                                       const long long *__restrict__ node_row, float *__restrict__ sh,
                                                       float *__restrict__ vf, int C) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    const int n = blockIdx.x, nl = C / 4, ng = 256 / nl;
    const int g = threadIdx.x / nl, fl = threadIdx.x % nl;
    const int rows = node_s[n] * node_s[n];
    if (g < ng) {
        const int cnt = (rows - g + ng - 1) / ng;
        const f4 acc = batched_sum(fL + ((size_t)node_row[n] + g) * C + 4 * fl, (size_t)ng * C, 0, cnt > 0 ? cnt : 0,
                                   [](int) { return 1.f; });
        st4(red + g * C + 4 * fl, acc);
    }
    __syncthreads();
    if (g == 0) {
        f4 t = ld4(red + 4 * fl);
        for (int k = 1; k < ng; ++k) t += ld4(red + k * C + 4 * fl);
        f4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = lrelu(t[j]);
        st4(sh + (size_t)n * C + 4 * fl, t);
        st4(vf + (size_t)n * C + 4 * fl, v);
    }
}

// the same sums from the row-panel partials the top level's combine-forward left behind (C = 64): thread per (node, channel),
// the node's panels in order
template <int CB>   // channels: 64 or 32
__global__ __launch_bounds__(256) void readout_nodes_panels(const float *__restrict__ psum, const int *__restrict__ node_panel, int nodes,
                                                            int npanels, float *__restrict__ sh, float *__restrict__ vf) {
    const int n = blockIdx.x * (256 / CB) + threadIdx.x / CB, c = threadIdx.x % CB;
    if (n >= nodes) return;
    const int p0 = node_panel[n], p1 = (n + 1 < nodes) ? node_panel[n + 1] : npanels;
    float t = 0.f;
    for (int p = p0; p < p1; ++p) t += psum[(size_t)p * CB + c];
    sh[(size_t)n * CB + c] = t;
    vf[(size_t)n * CB + c] = lrelu(t);
}

// one workgroup per molecule: g = sum_v vf (SumVectors), y = <g, W> (InnerProduct.h:39-46), loss (SquaredLoss.h:45-53)
__global__ void readout_molecules(const float *__restrict__ vf, const int *__restrict__ mol_ptr,
                                  const int *__restrict__ mol_nodes, const float *__restrict__ W,
                                  const float *__restrict__ target, float *__restrict__ g, float *__restrict__ yhat,
                                  float *__restrict__ loss, float *__restrict__ dy, int C) {
    __shared__ float red[256];
    const int m = blockIdx.x;
    float part = 0.f;
    for (int f = threadIdx.x; f < C; f += blockDim.x) {
        float acc = 0.f;
        for (int k = mol_ptr[m]; k < mol_ptr[m + 1]; ++k) acc += vf[(size_t)mol_nodes[k] * C + f];
        g[(size_t)m * C + f] = acc;
        part += acc * W[f];
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float y = red[0], t = target ? target[m] : 0.f;
        yhat[m] = y;
        if (loss) loss[m] = 0.5f * (y - t) * (y - t);
        dy[m] = y - t;  // SquaredLoss::backward: predict->gradient += predict - target
    }
}

// dW[f] += sum_m dy[m] g[m][f]   (InnerProduct.h:48-53, second operand)
__global__ __launch_bounds__(1024) void readout_dW(const float *__restrict__ dy, const float *__restrict__ g, float *__restrict__ dW,
                                                   int C, int nMol) {
    __shared__ float red[1024];
    const int nt = (int)blockDim.x;
    const int lanes = (C < nt) ? C : nt, rl = nt / lanes;
    const int f0 = threadIdx.x % lanes, rr = threadIdx.x / lanes;
    for (int fb = 0; fb < C; fb += lanes) {
        const int f = fb + f0;
        float acc = 0.f;
        if (f < C && rr < rl)
            for (int m = rr; m < nMol; m += rl) acc += dy[m] * g[(size_t)m * C + f];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rr == 0 && f < C) {
            float t = 0.f;
            for (int k = 0; k < rl; ++k) t += red[k * lanes + f0];
            dW[f] += t;
        }
        __syncthreads();
    }
}

// df_L[n][i][j][:] = dy[mol(n)] * W[:] * lrelu'(sh[n][:])   (InnerProduct first operand -> SumVectors -> LeakyReLU
// -> ShrinkTensor::backward broadcast, ShrinkTensor.h:52-61)
__global__ void readout_backward_nodes(const float *__restrict__ dy, const float *__restrict__ W, const float *__restrict__ sh,
                                       const int *__restrict__ node_mol, const int *__restrict__ node_s,
                                       const long long *__restrict__ node_row, float *__restrict__ dfL, int C) {
    const int n = blockIdx.x;
    const int s = node_s[n];
    float *dst = dfL + node_row[n] * C;
    const float d = dy[node_mol[n]];
    const int total = s * s * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int f = i % C;
        dst[i] = d * W[f] * (sh[(size_t)n * C + f] > 0.f ? 1.f : kAlpha);
    }
}

// Adam::Learn(alpha, nBatch) (GraphFlow/Adam.h:106-133) on the flat parameter buffer.  The reference advances its
// bias-correction powers INSIDE the element loop (beta1_t *= beta1 per element, :121,:125), so element i of the call that
// starts after n0 processed elements uses beta^(n0 + i + 1); restated in closed form, in double like the reference.
__global__ void adam_step(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                          size_t n, double alpha, double inv_batch, unsigned long long n0, double beta1, double beta2,
                          double eps) {
    const double l1 = log(beta1), l2 = log(beta2);
    GRID_STRIDE(i, n) {
        const double g = (double)grad[i] * inv_batch;
        const double mi = beta1 * (double)m[i] + (1.0 - beta1) * g;
        const double vi = beta2 * (double)v[i] + (1.0 - beta2) * g * g;
        const double t = (double)(n0 + i + 1);
        const double mh = mi / (1.0 - exp(t * l1)), vh = vi / (1.0 - exp(t * l2));
        m[i] = (float)mi;
        v[i] = (float)vi;
        p[i] = (float)((double)p[i] - alpha * mh / (sqrt(vh) + eps));
    }
}

// Momentum::Learn(learning_rate, nBatch) (GraphFlow/Momentum.h:64-71): m = gamma m + lr g / nBatch;  p -= m
__global__ void momentum_step(float *__restrict__ p, const float *__restrict__ grad, float *__restrict__ m, size_t n, double lr,
                              double inv_batch, double gamma) {
    GRID_STRIDE(i, n) {
        const double mi = gamma * (double)m[i] + lr * (double)grad[i] * inv_batch;
        m[i] = (float)mi;
        p[i] = (float)((double)p[i] - mi);
    }
}

// the same per node only: dsh[n][:] = dy[mol(n)] * W[:] * lrelu'(sh[n][:]) -- the fused top level reads this vector instead
// of a broadcast copy of it at every (i,j)
__global__ void readout_backward_nodevec(const float *__restrict__ dy, const float *__restrict__ W, const float *__restrict__ sh,
                                         const int *__restrict__ node_mol, float *__restrict__ dsh, int C, size_t total) {
    GRID_STRIDE(i, total) {
        const int f = i % C;
        const size_t n = i / C;
        dsh[i] = dy[node_mol[n]] * W[f] * (sh[i] > 0.f ? 1.f : kAlpha);
    }
}

__global__ void zero_f32(float *p, size_t n) { GRID_STRIDE(i, n) p[i] = 0.f; }

// rowscale[row] = (tot, tr) of the row's node (the per-row factors of the level's block products), from the per-node pairs
__global__ void expand_rowscale(float2 *__restrict__ rowscale, const float2 *__restrict__ node_scale, const int *__restrict__ node_s,
                                const long long *__restrict__ node_row) {
    const int n = blockIdx.x, s = node_s[n];
    const long long r0 = node_row[n];
    const float2 v = node_scale[n];
    for (int i = threadIdx.x; i < s * s; i += blockDim.x) rowscale[r0 + i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Level tables on the device (round 3; BatchLayout::device_tables).  What gfsmp::build_batch writes per node in its phase B --
// the reduced adjacency (SMP_omega.h:556-581: 1 on the diagonal and adj[v1][v2] elsewhere, or the Coulomb entries), its gated row
// sums and (tot, tr), the selection maps pi (:461-474) -- and per consumer entry in phase D (the inverse maps) are rows-sized:
// 2.2 of the 6.2 ms of host graph preparation on 32 threads (8 of 14 ms on 8) and 17 MB of the upload per 1024 molecules.  The
// kernels below build them from the receptive fields (sum-s ints per level), the pair tables and the molecules' adjacency
// matrices, with the host's summation orders (bit-identical tables: tests/test_smp_gpu.py::test_device_level_tables...).
// Workgroup per node; wave w builds the maps of the neighbours a = w, w + 4, ...
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_level_rows(const int *__restrict__ node_s, const int *__restrict__ node_mol,
                                                        const long long *__restrict__ node_row, const long long *__restrict__ node_pair,
                                                        const int *__restrict__ field, const int *__restrict__ prev_field,
                                                        const long long *__restrict__ pair_src_pair, const int *__restrict__ pair_src_s,
                                                        const int *__restrict__ mol_nv, const long long *__restrict__ mol_adj_off,
                                                        const int *__restrict__ mol_adj, const double *__restrict__ mol_coul,
                                                        float *__restrict__ adj, float *__restrict__ rsum, float *__restrict__ node_scale,
                                                        short *__restrict__ pi, int *__restrict__ node_present, int vmax) {
    extern __shared__ int lr_smem[];
    const int n = blockIdx.x, s = node_s[n], m = node_mol[n], V = mol_nv[m];
    const long long r0 = node_row[n], p0 = node_pair[n];
    int *f = lr_smem;                                             // [s] the node's field
    float *rs = reinterpret_cast<float *>(lr_smem + s);           // [s] gated row sums
    float *dg = rs + s;                                           // [s] gated diagonal
    short *pos = reinterpret_cast<short *>(dg + s);               // [4][vmax] position inside the source's field, -1 outside
    const int *madj = mol_adj + mol_adj_off[m];
    const double *mc = mol_coul ? mol_coul + mol_adj_off[m] : nullptr;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < s; i += blockDim.x) f[i] = field[p0 + i];
    for (int i = tid; i < 4 * vmax; i += blockDim.x) pos[i] = -1;
    __syncthreads();
    auto entry = [&](int i, int j) {
        return mc ? (float)mc[(size_t)f[i] * V + f[j]] : ((f[i] == f[j]) ? 1.f : (float)madj[(size_t)f[i] * V + f[j]]);
    };
    for (int idx = tid; idx < s * s; idx += blockDim.x) adj[r0 + idx] = entry(idx / s, idx % s);
    for (int i = tid; i < s; i += blockDim.x) {  // (entries with A <= 0 are skipped: RisiContraction_18.h:90; j in order, as the host sums)
        float acc = 0.f;
        for (int j = 0; j < s; ++j) {
            const float av = entry(i, j);
            if (av > 0.f) acc += av;
        }
        rs[i] = acc;
        rsum[p0 + i] = acc;
        const float d = entry(i, i);
        dg[i] = d > 0.f ? d : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f, tr = 0.f;
        for (int i = 0; i < s; ++i) {
            tot += rs[i];
            tr += dg[i];
        }
        node_scale[2 * (size_t)n] = tot;
        node_scale[2 * (size_t)n + 1] = tr;
    }
    short *mypos = pos + wave * vmax;
    unsigned cnt = 0;
    for (int a0 = 0; a0 < s; a0 += 4) {
        const int a = a0 + wave;
        const int *wf = nullptr;
        int sw = 0;
        if (a < s) {
            wf = prev_field + pair_src_pair[p0 + a];
            sw = pair_src_s[p0 + a];
            for (int k = lane; k < sw; k += 64) mypos[wf[k]] = (short)k;
        }
        __syncthreads();
        if (a < s)
            for (int p = lane; p < s; p += 64) {
                const short k = mypos[f[p]];
                pi[r0 + (long long)a * s + p] = k;
                cnt += k >= 0;
            }
        __syncthreads();
        if (a < s)
            for (int k = lane; k < sw; k += 64) mypos[wf[k]] = -1;
    }
    // rows with data of the node (level_table_stats sums them: one hot word for 17,000 workgroups' atomics cost 1 ms per level)
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    int *wcnt = reinterpret_cast<int *>(pos);
    if (lane == 0) wcnt[wave] = (int)cnt;
    __syncthreads();
    if (tid == 0) node_present[n] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// Round 4: ONE workgroup per node builds every rows-sized table of the node (receptive fields of at most 32 vertices): what
// build_level_rows, build_level_inv, expand_rowscale, build_trow and build_fwd_goff built in five launches that each re-read the
// selection maps -- 1.0 of the 1.45 ms of table-building kernels per prepared 1024-molecule batch, which run beside the step of
// another handle in the loop with a new batch every step.  The source fields of the node's neighbours are staged in LDS once and the
// map of a (neighbour, position) pair is a scan of at most 32 entries by its own thread (no per-neighbour barriers); the maps stay
// in LDS for the presence masks, the transposed-row table and the gather offsets of combine-forward.
// cons_of_pair[e] = index of pair e in its source's consumer list (the inverse of cons_pair: invert_cons_pair).
// The per-consumer entries of a level's consumer lists from the list itself (round 4, second session; the host wrote them in a pass of
// its own -- phase D of gfsmp::build_batch, 2 ms of a 1024-molecule prepare -- and uploaded 24 bytes per pair): consumer c is the pair
// e = cons_pair[c] = (node n, index a); its slab of the promoted tensor, its size, its first row and a.
__global__ void build_consumer_entries(const long long *__restrict__ cons_pair, const int *__restrict__ pair_node, const int *__restrict__ node_s,
                                       const long long *__restrict__ node_pair, const long long *__restrict__ node_row,
                                       const long long *__restrict__ node_p, long long *__restrict__ cons_slab, int *__restrict__ cons_s,
                                       long long *__restrict__ cons_row, int *__restrict__ cons_a, long long pairs) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= pairs) return;
    const long long e = cons_pair[c];
    const int n = pair_node[e], sz = node_s[n], a = (int)(e - node_pair[n]);
    cons_slab[c] = node_p[n] + (long long)a * sz * sz;
    cons_s[c] = sz;
    cons_row[c] = node_row[n];
    cons_a[c] = a;
}
__global__ void invert_cons_pair(const long long *__restrict__ cons_pair, int *__restrict__ cons_of_pair, long long pairs) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < pairs) cons_of_pair[cons_pair[c]] = (int)c;
}
__global__ __launch_bounds__(128) void build_node_tables(
    const int *__restrict__ node_s, const int *__restrict__ node_mol, const long long *__restrict__ node_row,
    const long long *__restrict__ node_pair, const int *__restrict__ field, const int *__restrict__ prev_field,
    const long long *__restrict__ pair_src_pair, const int *__restrict__ pair_src_s, const int *__restrict__ mol_nv,
    const long long *__restrict__ mol_adj_off, const int *__restrict__ mol_adj, const double *__restrict__ mol_coul,
    float *__restrict__ adj, float *__restrict__ rsum, float *__restrict__ node_scale, short *__restrict__ pi,
    int *__restrict__ node_present, int swp,                                     // swp: largest field of the level below
    const int *__restrict__ cons_of_pair, const long long *__restrict__ cons_inv_off, short *__restrict__ inv,   // or null (no consumers' maps)
    float2 *__restrict__ rowscale,                                               // or null
    int *__restrict__ trow, unsigned char *__restrict__ rowflag, int *__restrict__ trowf,   // trow null: none of the three
    int2 *__restrict__ goff) {                                                   // or null
    extern __shared__ int nt_smem[];
    const int n = blockIdx.x, s = node_s[n], m = node_mol[n], V = mol_nv[m];
    const long long r0 = node_row[n], p0 = node_pair[n];
    int *f = nt_smem;                                             // [s] the node's field
    float *rs = reinterpret_cast<float *>(f + s);                 // [s] gated row sums
    float *dg = rs + s;                                           // [s] gated diagonal
    unsigned *mask = reinterpret_cast<unsigned *>(dg + s);        // [s] bit p: neighbour a's source holds the vertex of position p
    int *sf = reinterpret_cast<int *>(mask + s);                  // [s][swp] the neighbours' source fields, -1 padded
    short *spi = reinterpret_cast<short *>(sf + s * swp);         // [s][s] the maps
    const int *madj = mol_adj + mol_adj_off[m];
    const double *mc = mol_coul ? mol_coul + mol_adj_off[m] : nullptr;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < s; i += nt) f[i] = field[p0 + i];
    for (int i = tid; i < s * swp; i += nt) {
        const int a = i / swp, k = i - a * swp;
        sf[i] = k < pair_src_s[p0 + a] ? prev_field[pair_src_pair[p0 + a] + k] : -1;
    }
    __syncthreads();
    auto entry = [&](int i, int j) {
        return mc ? (float)mc[(size_t)f[i] * V + f[j]] : ((f[i] == f[j]) ? 1.f : (float)madj[(size_t)f[i] * V + f[j]]);
    };
    for (int idx = tid; idx < s * s; idx += nt) adj[r0 + idx] = entry(idx / s, idx % s);
    for (int i = tid; i < s; i += nt) {  // (entries with A <= 0 are skipped: RisiContraction_18.h:90; j in order, as the host sums)
        float acc = 0.f;
        for (int j = 0; j < s; ++j) {
            const float av = entry(i, j);
            if (av > 0.f) acc += av;
        }
        rs[i] = acc;
        rsum[p0 + i] = acc;
        const float d = entry(i, i);
        dg[i] = d > 0.f ? d : 0.f;
    }
    // the maps: pi[a][p] = position of the vertex of position p inside the field of neighbour a's source, -1 outside (:461-474)
    for (int i = tid; i < s * s; i += nt) {
        const int a = i / s, p = i - a * s, v = f[p];
        const int *row = sf + a * swp;
        int k = -1;
        for (int kk = 0; kk < swp; ++kk) k = row[kk] == v ? kk : k;   // (a field holds a vertex once)
        pi[r0 + i] = (short)k;
        spi[i] = (short)k;
        if (inv && k >= 0) inv[cons_inv_off[cons_of_pair[p0 + a]] + k] = (short)p;
    }
    __syncthreads();
    float tot = 0.f, tr = 0.f;   // (every thread forms them, in the host's order: the row factors below need them)
    for (int i = 0; i < s; ++i) {
        tot += rs[i];
        tr += dg[i];
    }
    if (tid == 0) {
        node_scale[2 * (size_t)n] = tot;
        node_scale[2 * (size_t)n + 1] = tr;
    }
    for (int a = tid; a < s; a += nt) {
        unsigned mk = 0u;
        for (int p = 0; p < s; ++p) mk |= (spi[a * s + p] >= 0 ? 1u : 0u) << p;
        mask[a] = mk;
    }
    __syncthreads();
    if (tid == 0) {
        int cnt = 0;
        for (int a = 0; a < s; ++a) cnt += __popc(mask[a]);
        node_present[n] = cnt;
    }
    for (int i = tid; i < s * s; i += nt) {
        const int x = i / s, e = i - x * s, it = e * s + x;
        if (rowscale) rowscale[r0 + i] = make_float2(tot, tr);
        const short pxe = spi[i], pex = spi[it];
        if (goff) goff[r0 + i] = make_int2(pxe >= 0 ? (int)(pair_src_pair[p0 + x] + pxe) : -1, pex >= 0 ? (int)(pair_src_pair[p0 + e] + pex) : -1);
        if (trow) {
            const long long t = r0 + it;
            trow[r0 + i] = (int)t;
            const bool own = pxe >= 0, trp = pex >= 0;
            // row (b, c) = (x, e) of the S_bc / T10 blocks has data when SOME neighbour's source holds both b and c
            unsigned both = 0u;
            for (int a = 0; a < s; ++a) both |= (mask[a] >> x) & (mask[a] >> e);
            const bool bc = (both & 1u) != 0;
            rowflag[r0 + i] = (own ? 1 : 0) | (bc ? 2 : 0);
            if (trowf)
                trowf[r0 + i] = (t < (1ll << 29)) ? (int)((unsigned)t | (own ? 0x80000000u : 0u) | (trp ? 0x40000000u : 0u) | (bc ? 0x20000000u : 0u)) : -1;
        }
    }
}

// stats = {max |tot| (float bits), max |tr|, rows with data (two words)} of a level; one workgroup, fixed order
__global__ __launch_bounds__(1024) void level_table_stats(const float *__restrict__ node_scale, const int *__restrict__ node_present,
                                                          int nodes, unsigned *__restrict__ stats) {
    __shared__ float mt[1024], mr[1024];
    __shared__ unsigned long long sc[1024];
    float a = 0.f, b = 0.f;
    unsigned long long c = 0;
    for (int n = threadIdx.x; n < nodes; n += 1024) {
        a = fmaxf(a, fabsf(node_scale[2 * (size_t)n]));
        b = fmaxf(b, fabsf(node_scale[2 * (size_t)n + 1]));
        c += (unsigned long long)node_present[n];
    }
    mt[threadIdx.x] = a, mr[threadIdx.x] = b, sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            mt[threadIdx.x] = fmaxf(mt[threadIdx.x], mt[threadIdx.x + o]);
            mr[threadIdx.x] = fmaxf(mr[threadIdx.x], mr[threadIdx.x + o]);
            sc[threadIdx.x] += sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] = __float_as_uint(mt[0]);
        stats[1] = __float_as_uint(mr[0]);
        stats[2] = (unsigned)(sc[0] & 0xffffffffull);
        stats[3] = (unsigned)(sc[0] >> 32);
    }
}

// inv[cons_inv_off[c] + k] = p  where source position k is the image of the consumer's position p (inv prefilled with -1)
__global__ __launch_bounds__(256) void build_level_inv(const long long *__restrict__ cons_pair, const int *__restrict__ pair_node,
                                                       const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                                       const long long *__restrict__ node_pair, const long long *__restrict__ cons_inv_off,
                                                       const short *__restrict__ pi, short *__restrict__ inv, long long pairs) {
    const long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= pairs) return;
    const int lane = threadIdx.x & 63;
    const long long e = cons_pair[c];
    const int n = pair_node[e], s = node_s[n], a = (int)(e - node_pair[n]);
    const short *row = pi + node_row[n] + (long long)a * s;
    short *iv = inv + cons_inv_off[c];
    for (int p = lane; p < s; p += 64) {
        const short k = row[p];
        if (k >= 0) iv[k] = (short)p;
    }
}

// trow[row of (x, e)] = row of (e, x) inside the same node (compact O layout of the fused C = 64 level, smp_level_c64.hip)
__global__ void build_trow(int *__restrict__ trow, const int *__restrict__ node_s, const long long *__restrict__ node_row,
                           const short *__restrict__ pi, unsigned char *__restrict__ rowflag, int *__restrict__ trowf) {
    const int n = blockIdx.x, s = node_s[n];
    const long long r0 = node_row[n];
    for (int i = threadIdx.x; i < s * s; i += blockDim.x) {
        const int it = (i % s) * s + i / s;
        const long long t = r0 + it;
        trow[r0 + i] = (int)t;
        const bool own = pi[r0 + i] >= 0, tr = pi[r0 + it] >= 0;
        // row (b, c) of the S_bc / T10 blocks (sums over a of P[a,b,c]) has data when SOME source a holds both b and c: 92 % of the
        // rows at level 3 of QM9-size molecules, 71 % at level 2, 29 % at level 1 (only b == c: a level-0 field is one vertex)
        bool bc = false;
        {
            const int b = i / s, c = i % s;
            for (int a = 0; a < s && !bc; ++a) bc = pi[r0 + a * s + b] >= 0 && pi[r0 + a * s + c] >= 0;
        }
        rowflag[r0 + i] = (own ? 1 : 0) | (bc ? 2 : 0);  // bit 0: the S_ab / T6 blocks of the row are written by tables-forward
                                                          // (DevLevel::t_zeros), bit 1: its S_bc / T10 blocks are
        if (trowf)
            trowf[r0 + i] = (t < (1ll << 29)) ? (int)((unsigned)t | (own ? 0x80000000u : 0u) | (tr ? 0x40000000u : 0u) | (bc ? 0x20000000u : 0u)) : -1;
    }
}

// RisiContraction_18_dropout over the nodes of a level: slice k of node n is multiplied by scale if bit k of keep[n] is set,
// else zeroed (RisiContraction_18_dropout.h:106-132 forward, :479-510 backward: dropped slices neither produce nor receive)
// (round 4: a dropped slice is a store of zeros, a kept one at scale 1 -- train mode -- is not touched at all, and the channels move as
//  float4 where C % 4 == 0: the kernel read and wrote all of Q element by element, 5.9 of the 34 ms of an SMP_sigma_pairgraphs step)
template <int VW>
__global__ void node_slice_scale(float *__restrict__ Q, const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                 const unsigned *__restrict__ keep, float scale, int C) {
    const int n = blockIdx.x;
    const unsigned m = keep[n];
    const int cv = C / VW;
    const size_t cnt = (size_t)node_s[n] * node_s[n] * 18 * cv;
    float *q = Q + (size_t)node_row[n] * 18 * C;
    for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int k = (int)((i / cv) % 18);
        const bool kept = (m >> k) & 1u;
        if (kept && scale == 1.f) continue;
        if constexpr (VW == 4) {
            float4 *p4 = reinterpret_cast<float4 *>(q) + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kept) {
                v = *p4;
                v.x *= scale, v.y *= scale, v.z *= scale, v.w *= scale;
            }
            *p4 = v;
        } else {
            q[i] = kept ? q[i] * scale : 0.f;
        }
    }
}
static void launch_node_slice_scale(gf_ctx *ctx, float *Q, const int *node_s, const long long *node_row, const unsigned *keep, float scale, int C,
                                    int nNodes) {
    if (C % 4 == 0 && (((uintptr_t)Q) & 15) == 0)
        hipLaunchKernelGGL(node_slice_scale<4>, dim3(nNodes), dim3(256), 0, ctx->stream, Q, node_s, node_row, keep, scale, C);
    else
        hipLaunchKernelGGL(node_slice_scale<1>, dim3(nNodes), dim3(256), 0, ctx->stream, Q, node_s, node_row, keep, scale, C);
}

// physics towers: level_feature[l] = sum over the molecule's vertices of LeakyReLU(sum_ij f_l[v]) (SMP_omega_physics.h:572-588),
// written into columns [off, off + C) of the molecule's feature row (ConcatVectors, :590)
__global__ void level_feature_sum(const float *__restrict__ vf, const int *__restrict__ mol_ptr, const int *__restrict__ node_of_vertex,
                                  float *__restrict__ feat, int C, int width, int off) {
    const int m = blockIdx.x;
    for (int f = threadIdx.x; f < C; f += blockDim.x) {
        float acc = 0.f;
        for (int k = mol_ptr[m]; k < mol_ptr[m + 1]; ++k) acc += vf[(size_t)node_of_vertex[k] * C + f];
        feat[(size_t)m * width + off + f] = acc;
    }
}

// the same gradient as ONE vector per node (a fused level's combine-backward adds it to every row of the node itself)
__global__ void level_feature_nodevec(const float *__restrict__ dfeat, const float *__restrict__ sh, const int *__restrict__ node_mol,
                                      float *__restrict__ out, int C, int width, int off, size_t total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const size_t n = i / C;
        out[i] = dfeat[(size_t)node_mol[n] * width + off + c] * (sh[i] > 0.f ? 1.f : kAlpha);
    }
}

// reverse: df_l[n][i][j][:] (+)= dfeat[mol(n)][off + :] * lrelu'(sh_l[n][:])   (SumVectors -> LeakyReLU -> ShrinkTensor::backward)
__global__ void level_feature_backward(const float *__restrict__ dfeat, const float *__restrict__ sh, const int *__restrict__ node_mol,
                                       const int *__restrict__ node_s, const long long *__restrict__ node_row, float *__restrict__ df,
                                       int C, int width, int off, int accumulate) {
    const int n = blockIdx.x;
    const int s = node_s[n];
    float *dst = df + node_row[n] * C;
    const float *g = dfeat + (size_t)node_mol[n] * width + off;
    const int total = s * s * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int f = i % C;
        const float v = g[f] * (sh[(size_t)n * C + f] > 0.f ? 1.f : kAlpha);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

size_t param_count(const gfsmp::Config &c);
static size_t param_count_of(const gf_smp *s) { return param_count(s->ucfg); }   // (the caller's layout)

// page-locked host memory for the per-batch tables (smp_prep.h: table_alloc): their uploads then run on the DMA engines beside
// the step that is executing, instead of blit kernels queued behind its compute kernels.  16-byte header: how it was obtained.
void *pinned_table_alloc(size_t bytes) {
    void *p = nullptr;
    const size_t total = bytes + 16;
    unsigned kind = 1;
    // (portable: the preparation's worker threads never call hipSetDevice, and a table pinned against device 0 only would be
    //  pageable memory to the uploads of every other rank's device)
    if (hipHostMalloc(&p, total, hipHostMallocPortable) != hipSuccess) p = nullptr;
    if (!p) {
        (void)hipGetLastError();
        p = std::malloc(total);
        kind = 0;
        if (!p) return nullptr;
    }
    *static_cast<unsigned *>(p) = kind;
    return static_cast<char *>(p) + 16;
}
void pinned_table_free(void *q) {
    if (!q) return;
    void *p = static_cast<char *>(q) - 16;
    if (*static_cast<unsigned *>(p) == 1)
        (void)hipHostFree(p);
    else
        std::free(p);
}

template <typename T>
gf_status upload(gf_smp *s, T **dst, const void *src, size_t count) {
    *dst = nullptr;
    const size_t bytes = sizeof(T) * (count ? count : 1);
    // best fit among the idle blocks of the pool (no block more than twice the request: keeps big blocks for big buffers)
    int best = -1;
    for (size_t i = 0; i < s->pool.size(); ++i) {
        const gf_smp::Block &b = s->pool[i];
        if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (best < 0 || b.bytes < s->pool[best].bytes)) best = (int)i;
    }
    void *p = nullptr;
    if (best >= 0) {
        s->pool[best].used = true;
        s->pool[best].idle = 0;
        p = s->pool[best].p;
    } else {
        const size_t cap = bytes + bytes / 8 + 256;  // slack: the next batch is about, not exactly, this size
        hipError_t e = hipMalloc(&p, cap);
        if (e != hipSuccess) return fail(s->ctx, GF_ERR_NOMEM, "smp: hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        gf_smp::Block b = {p, cap, true, 0};
        s->pool.push_back(b);
    }
    if (src && count)
        GF_HIP_TRY(s->ctx, hipMemcpyAsync(p, src, sizeof(T) * count, hipMemcpyHostToDevice, s->upload ? s->upload : s->ctx->stream));
    else if (gf::poison_buffers()) {  // GF_POISON=1 (debug): a buffer handed out without contents is filled with NaN bit patterns, so a
        hipStream_t st = s->upload ? s->upload : s->ctx->stream;                    // read-before-write shows
        GF_HIP_TRY(s->ctx, hipMemsetAsync(p, 0xff, bytes, st));
        // (finished before anything else is launched: buffers are also taken from the pool in the middle of a pass -- the promoted
        //  stack of the op-by-op levels -- where the upload stream is not ordered against the pass)
        GF_HIP_TRY(s->ctx, hipStreamSynchronize(st));
    }
    *dst = static_cast<T *>(p);
    return GF_OK;
}

gf_status ensure_P_impl(gf_smp *s) {
    if (s->P) return GF_OK;
    return upload(s, &s->P, nullptr, s->P_count);
}

// End of a batch: its buffers go back to the pool (blocks idle for three batches in a row are returned to the device).
// the handle's buffers were last touched by the launches before this mark
void mark_used(gf_smp *s) {
    if (s->ev_last && hipEventRecord(s->ev_last, s->ctx->stream) == hipSuccess) s->used = true;
}

void release(gf_smp *s) {
    if (s->ctx) {
        // wait for this handle's own work only: another handle of the context may be in the middle of its step
        if (s->ev_last) {
            if (s->used) (void)hipEventSynchronize(s->ev_last);
        } else {
            (void)hipStreamSynchronize(s->ctx->stream);
        }
        s->used = false;
    }
    std::vector<gf_smp::Block> keep;
    for (gf_smp::Block &b : s->pool) {
        if (!b.used && ++b.idle >= 3) {
            (void)hipFree(b.p);
            continue;
        }
        b.used = false;
        keep.push_back(b);
    }
    s->pool.swap(keep);
    s->lv.clear();
    s->own_t = s->own_y = s->own_loss = s->own_feat = nullptr;
    s->prepared = s->forwarded = false;
}

void release_pool(gf_smp *s) {
    for (gf_smp::Block &b : s->pool) (void)hipFree(b.p);
    s->pool.clear();
}

// the handle-owned parameter / gradient buffers (host-pointer mode), created on first use
gf_status own_model(gf_smp *s) {
    if (s->own_p) return GF_OK;
    GF_HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    const size_t n = param_count_of(s);
    GF_HIP_TRY(s->ctx, hipMalloc(reinterpret_cast<void **>(&s->own_p), n * sizeof(float)));
    GF_HIP_TRY(s->ctx, hipMalloc(reinterpret_cast<void **>(&s->own_g), n * sizeof(float)));
    GF_HIP_TRY(s->ctx, hipMemsetAsync(s->own_p, 0, n * sizeof(float), s->ctx->stream));
    GF_HIP_TRY(s->ctx, hipMemsetAsync(s->own_g, 0, n * sizeof(float), s->ctx->stream));
    return GF_OK;
}

struct ParamView {
    const float *H, *W;
    std::vector<const float *> K, b;
};
size_t param_count(const gfsmp::Config &c) {
    size_t n = (size_t)c.nChanels * c.fdim();
    for (int l = 1; l <= c.nLevels; ++l)
        n += (size_t)c.nContractions * c.level_channels(l - 1) * c.level_channels(l) + c.level_channels(l);
    return n + (c.physics ? 0 : c.nChanels);  // a physics tower ends in its level features: the head's weights are the caller's
}
// order H, (K_1, b_1), ..., (K_L, b_L), W -- the registration order of SMP_omega.h:289-295 (= save_model order)
template <typename P>
void view_params(const gfsmp::Config &c, P *base, P **H, std::vector<P *> *K, std::vector<P *> *b, P **W) {
    P *p = base;
    *H = p;
    p += (size_t)c.nChanels * c.fdim();
    K->assign(c.nLevels + 1, nullptr);
    b->assign(c.nLevels + 1, nullptr);
    for (int l = 1; l <= c.nLevels; ++l) {
        (*K)[l] = p;
        p += (size_t)c.nContractions * c.level_channels(l - 1) * c.level_channels(l);
        (*b)[l] = p;
        p += c.level_channels(l);
    }
    *W = p;
}

// RisiContraction_18 over every node of level l.  Nodes are sorted by receptive-field size, so consecutive buckets are
// merged into at most four launches (size classes s <= PPW, 2 PPW, 4 PPW, 8 PPW of the slab kernels) through the ragged
// entry points; anything larger falls back to one uniform launch per bucket.
gf_status smp_contract(gf_smp *s, int l, bool backward) {
    gf_ctx *ctx = s->ctx;
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.level_channels(l - 1), nK = s->cfg.nContractions;  // the contraction runs on the level below's channels
    const int ppw = (C <= 16) ? 16 : (C <= 32) ? 8 : 4;
    const gf_ragged_nodes t = {d.pair_node, d.node_s, d.node_p, d.node_row, d.node_pair, (long long)h.rows, (long long)h.pairs};
    gf_status st = ensure_P_impl(s);
    if (st != GF_OK) return st;
    // (_10 / _50 of the SMP_2D_ver6 / ver7 wirings: one uniform launch per size bucket)
    const bool ragged_ok = nK == 18 && r18_ragged_supported(ppw, C, s->P, d.Q);
    size_t k = 0;
    for (int cls = 1; cls <= 8 && ragged_ok && k < h.buckets.size(); cls *= 2) {
        const int smax_cls = cls * ppw;
        const size_t k0 = k;
        int smax = 0;
        while (k < h.buckets.size() && h.buckets[k].s <= smax_cls) smax = h.buckets[k++].s;
        if (k == k0) continue;
        const long long lo = h.node_pair[h.buckets[k0].first_node];
        const long long hi = (k < h.buckets.size()) ? h.node_pair[h.buckets[k].first_node] : (long long)h.pairs;
        st = backward ? r18_backward_ragged(ctx, d.Q, d.adj, s->P, t, lo, hi, smax, C, 0)
                      : r18_forward_ragged(ctx, s->P, d.adj, d.Q, t, lo, hi, smax, C);
        if (st != GF_OK) return st;
    }
    for (; k < h.buckets.size(); ++k) {  // sizes beyond the slab kernels (or unaligned C): uniform launches
        const gfsmp::Bucket &bk = h.buckets[k];
        float *Pb = s->P + bk.first_p * C, *Qb = d.Q + bk.first_row * (long long)(nK * C);
        st = backward ? gf_contract_backward_f32(ctx, nK, Qb, d.adj + bk.first_row, Pb, bk.s, C, bk.count, 0)
                      : gf_contract_forward_f32(ctx, nK, Pb, d.adj + bk.first_row, Qb, bk.s, C, bk.count);
        if (st != GF_OK) return st;
    }
    return GF_OK;
}

}  // namespace

gf_status ensure_P(gf_smp *s) { return ensure_P_impl(s); }
size_t feature_width(const gfsmp::Config &c) {
    size_t w = 0;
    for (int l = 0; l <= c.nLevels; ++l) w += (size_t)c.level_channels(l);
    return w;
}

// Data-parallel reverse sweep.  The flat gradient buffer is H | K_1 b_1 | ... | K_L b_L | W; the segment of level l is
// [K_l | b_l] (plus W for l = L: the readout gradient is the first thing the sweep computes) and H for l = 0.  The segment is
// final on the context's current stream when this is called: the communicator's stream waits for that point and runs the
// all-reduce there, while the sweep continues with the table gradients and the levels below.
gf_status smp_dp_level_done(gf_smp *s, int l) {
    if (!s->dp_grads) return GF_OK;
    gf_ctx *ctx = s->ctx;
    const gfsmp::Config &c = s->cfg;
    const size_t C = (size_t)c.nChanels, nH = C * c.fdim(), per = (size_t)c.nContractions * C * C + C;
    float *seg = s->dp_grads;
    size_t n = nH;
    if (l >= 1) {
        seg += nH + (size_t)(l - 1) * per;
        n = per + (l == c.nLevels ? C : 0);
    }
    hipStream_t comm = dist_stream(ctx);
    GF_HIP_TRY(ctx, hipEventRecord(s->ev_grad, ctx->stream));
    GF_HIP_TRY(ctx, hipStreamWaitEvent(comm, s->ev_grad, 0));
    char what[64];
    if (l >= 1) std::snprintf(what, sizeof what, "gf_smp_backward: gradient segment [K_%d | b_%d%s]", l, l, l == c.nLevels ? " | W" : "");
    else std::snprintf(what, sizeof what, "gf_smp_backward: gradient segment [H]");
    gf_status st = dist_allreduce_on(ctx, seg, n, comm, what);
    if (st == GF_OK && l == 0 && s->n_extra)   // (SMP_2D_ver7 on the 18-slice level: the extra products' blocks sit behind W; every level has written its own by now)
        st = dist_allreduce_on(ctx, s->dp_grads + param_count(c), (size_t)c.nLevels * s->n_extra * C * C, comm, "gf_smp_backward: gradient segment [X_1 .. X_L]");
    return st;
}
}  // namespace gf

using gf::fail;

extern "C" {

gf_status gf_smp_create(gf_ctx *ctx, const gf_smp_config *cfg, gf_smp **out) { return gf::smp_create(ctx, cfg, /*pad_channels=*/true, out); }

}  // extern "C"

// What the device computes with (gf_smp::cfg, dup_channels, n_extra), derived from the caller's configuration (gf_smp::ucfg).
// allow_embed = false: the `_10` / `_50` families on their own op-by-op levels whatever the environment says -- gf_smp_prepare switches a
// handle to that plan for a batch the embedding in the 18-slice level cannot take (an asymmetric adjacency, Coulomb entries <= 0, a `_50`
// Coulomb batch) and back for the next batch it can (round-5 advice: the refusal used to surface mid-epoch, with an environment variable
// read at create time as the only way out).  The caller's parameter layout does not depend on the plan.
void gf::smp_derive_plan(gf_smp *s, bool allow_embed) {
    // Round 4: the channel count the DEVICE computes with.  The dedicated kernels of the fused level exist at 32 and 64 channels, the
    // generic fused level needs C % 4 == 0, and anything else ran the op-by-op level on the one-thread-per-element contraction kernels
    // (the reference's own tests use nChanels = 10: 35.8 ms per 1024-molecule step, against 8.0 ms at 12 channels and 4.1 ms at 32).
    // A model is therefore computed with its channels PADDED to 32 / 64 (above 64: to a multiple of 4): padded weights, biases and
    // features are zero, LeakyReLU(0) = 0 keeps them zero through every level, so the real channels see exactly the sums they saw
    // before (plus zero terms).  ucfg keeps the caller's layout: parameters, gradients, features and activations cross the C ABI in
    // it and are padded / cropped at the boundary (gf_smp_forward / gf_smp_backward).  GF_SMP_PAD_CHANNELS=0: compute at nChanels.
    const bool pad_channels = s->req_pad_channels;
    const int min_pad = s->req_min_pad;
    s->cfg = s->ucfg;
    s->dup_channels = 0;
    s->n_extra = 0;
    {
        const char *e = std::getenv("GF_SMP_PAD_CHANNELS");
        const int C = s->cfg.nChanels;
        int Cc = C;
        if ((pad_channels || (e && e[0] == '2')) && !(e && e[0] == '0') && s->cfg.nContractions == 18 && s->cfg.nLevels < gf::kPadMaxLevels) {   // (2: tests)
            // (round 5: a 16-channel build of the row-panel family -- the reference's own models have nChanels = 10; min_pad = 32 keeps a
            //  tower that will run under slice dropout on the 32-channel kernels, the only ones with per-product row factors)
            if (C <= 16 && min_pad <= 16 && !(e && e[0] == '3')) Cc = 16;   // GF_SMP_PAD_CHANNELS=3: pad to 32 as rounds 1-4 did (tests)
            else if (C <= 32) Cc = 32;
            else if (C <= 64) Cc = 64;
            else Cc = (C + 3) & ~3;
            // a physics tower (channels halve per level, SMP_omega_physics.h:141-151) is computed at ONE width: K_l [18 C_{l-1}][C_l]
            // sits in the corner of a square [18 Cc][Cc] block, the level features are cropped level by level
            if (s->cfg.physics) s->cfg.uniform = 1;
        }
        // SMP_2D_ver6 (RisiContraction_10) embedded in the 18-slice fused level (see gf_smp::dup_channels): 2 C channels padded to 16 / 32 / 64.
        // GF_SMP_VER6_FUSED=0: the `_10` contraction op by op.  gf_smp_prepare switches to that plan by itself for a batch with an asymmetric adjacency.
        // (only where every field fits the fused level -- max_receptive_field <= 64 (32 until round 6): on an op-by-op level the 18-slice model at 2 C padded
        //  channels moves more than the `_10` / `_50` contraction at C; GF_SMP_VER6_FUSED=2 / GF_SMP_VER7_FUSED=2 embed at any cap)
        auto embed = [&](const char *name) {
            const char *v = std::getenv(name);
            if (!allow_embed || (v && v[0] == '0')) return false;
            return s->cfg.max_receptive_field <= gf::kFusedMaxField || (v && v[0] == '2');   // (64 since round 6: fields of 33 .. 64 positions stay fused)
        };
        if (pad_channels && !(e && e[0] == '0') && s->cfg.nContractions == 10 && 2 * C <= 64 && s->cfg.nLevels < gf::kPadMaxLevels && !s->cfg.physics &&
            embed("GF_SMP_VER6_FUSED")) {
            s->dup_channels = C;
            s->cfg.nContractions = 18;
            s->cfg.custom_matmul = 0;   // (the device's own copy of the weights is in the [18 Cc][Cc] layout whatever the caller's is)
            Cc = 2 * C <= 16 ? 16 : 2 * C <= 32 ? 32 : 64;
        }
        // SMP_2D_ver7 (RisiContraction_50) the same way, with three extra products per level (gf_smp::n_extra).  GF_SMP_VER7_FUSED=0: op by op.
        if (pad_channels && !(e && e[0] == '0') && s->cfg.nContractions == 50 && 2 * C <= 64 && s->cfg.nLevels < gf::kPadMaxLevels && !s->cfg.physics &&
            embed("GF_SMP_VER7_FUSED")) {
            s->dup_channels = C;
            s->n_extra = 3;
            s->cfg.nContractions = 18;
            s->cfg.custom_matmul = 0;
            Cc = 2 * C <= 16 ? 16 : 2 * C <= 32 ? 32 : 64;
        }
        // the `_10` / `_50` wirings (SMP_2D_ver6 / ver7: op-by-op levels): a channel count that is not a multiple of 4 runs the contraction
        // kernels at one channel per lane; padded to the next multiple (10 -> 12) they take the float4 / one-stream-per-graph kernels
        if (pad_channels && !(e && e[0] == '0') && s->cfg.nContractions != 18 && !s->dup_channels && s->cfg.nLevels < gf::kPadMaxLevels && !s->cfg.physics) {
            const char *m = std::getenv("GF_SMP_PAD_FAMILY");   // experiment: 0 = off, 4 / 8 / 16 = pad to that multiple
            const int mult = m ? std::atoi(m) : 4;
            if (mult > 0) Cc = (C + mult - 1) / mult * mult;
        }
        s->cfg.nChanels = Cc;
    }
}

// pad_channels = false: compute at the configuration's own channel counts (the towers of a model with RisiContraction_18_dropout --
// SMP_sigma_pairgraphs -- whose levels run op by op, where a padded width only costs)
gf_status gf::smp_create(gf_ctx *ctx, const gf_smp_config *cfg, bool pad_channels, gf_smp **out, int min_pad) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!cfg || !out) return fail(ctx, GF_ERR_INVALID, "gf_smp_create: null argument");
    if (cfg->nLevels < 1 || cfg->nChanels < 1 || cfg->nFeatures < 1 || cfg->nDepth < 0 || cfg->max_receptive_field < 1)
        return fail(ctx, GF_ERR_INVALID, "gf_smp_create: bad configuration");
    gfsmp::table_alloc = gf::pinned_table_alloc;   // (before the first table is built; idempotent)
    gfsmp::table_free = gf::pinned_table_free;
    gf_smp *s = new gf_smp();
    s->ctx = ctx;
    s->cfg.nLevels = cfg->nLevels;
    s->cfg.nChanels = cfg->nChanels;
    s->cfg.nFeatures = cfg->nFeatures;
    s->cfg.nDepth = cfg->nDepth;
    s->cfg.max_receptive_field = cfg->max_receptive_field;
    s->cfg.has_WL_ordering = cfg->has_WL_ordering;
    s->cfg.nContractions = cfg->nContractions ? cfg->nContractions : 18;
    s->cfg.custom_matmul = cfg->custom_matmul ? 1 : 0;
    s->cfg.physics = cfg->physics ? 1 : 0;
    s->ucfg = s->cfg;
    s->req_pad_channels = pad_channels;
    s->req_min_pad = min_pad;
    gf::smp_derive_plan(s, /*allow_embed=*/true);
    if (s->cfg.physics && (s->cfg.nDepth != 0 || s->cfg.nContractions != 18 || s->cfg.custom_matmul)) {
        delete s;
        return fail(ctx, GF_ERR_INVALID, "gf_smp_create: a physics tower has nDepth 0 (raw features), RisiContraction_18 and [18 C', C] weights");
    }
    if (s->cfg.nContractions != 10 && s->cfg.nContractions != 18 && s->cfg.nContractions != 50) {
        const int bad = cfg->nContractions;
        delete s;
        return fail(ctx, GF_ERR_INVALID, "gf_smp_create: nContractions = %d (expected 10, 18 or 50)", bad);
    }
    {
        const char *e = std::getenv("GF_SMP_BWD_GATHER");  // 0: keep the two-kernel tables-backward + consumer gather
        s->bwd_gather = (e && e[0] == '0') ? 0 : 1;
    }
    *out = s;
    return GF_OK;
}

// Moves a handle between its two plans (see smp_derive_plan).  Everything sized by the device-side configuration goes: the batch's
// buffers (release), the padded parameter / gradient / feature copies.  What is in the caller's layout stays: the handle-owned model,
// the optimiser's moments, the pool of device blocks.
gf_status gf::smp_switch_plan(gf_smp *s, bool embed) {
    gf_ctx *ctx = s->ctx;
    if (s->ev_last && s->used) GF_HIP_TRY(ctx, hipEventSynchronize(s->ev_last));   // (the handle's last launch: its padded buffers are about to go)
    GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    gf::release(s);
    if (s->pad_p) (void)hipFree(s->pad_p);
    if (s->pad_g) (void)hipFree(s->pad_g);
    if (s->pad_feat) (void)hipFree(s->pad_feat);
    s->pad_p = s->pad_g = s->pad_feat = nullptr;
    s->pad_feat_n = 0;
    s->extra_w = nullptr;
    s->extra_g = nullptr;
    gf::smp_derive_plan(s, embed);
    s->embed_auto_off = !embed;
    return GF_OK;
}

extern "C" {

gf_status gf_smp_destroy(gf_smp *s) {
    if (!s) return GF_OK;
    if (s->rs_inv) (void)hipFree(s->rs_inv);
    gf::release(s);
    gf::release_pool(s);
    if (s->upload) (void)hipStreamDestroy(s->upload);
    if (s->ev_last) (void)hipEventDestroy(s->ev_last);
    if (s->ev_grad) (void)hipEventDestroy(s->ev_grad);
    if (s->ev_comm) (void)hipEventDestroy(s->ev_comm);
    if (s->ev_mask) {
        (void)hipEventSynchronize(s->ev_mask);
        (void)hipEventDestroy(s->ev_mask);
    }
    if (s->mask_stage) (void)hipHostFree(s->mask_stage);
    if (s->adam_m) (void)hipFree(s->adam_m);
    if (s->adam_v) (void)hipFree(s->adam_v);
    if (s->own_p) (void)hipFree(s->own_p);
    if (s->own_g) (void)hipFree(s->own_g);
    if (s->pad_p) (void)hipFree(s->pad_p);
    if (s->pad_g) (void)hipFree(s->pad_g);
    if (s->pad_feat) (void)hipFree(s->pad_feat);
    delete s;
    return GF_OK;
}

size_t gf_smp_param_count(const gf_smp *s) { return s ? gf::param_count(s->ucfg) : 0; }

// ---- host-pointer mode of the driver: the handle owns the model, batches and results cross as host arrays -------------
gf_status gf_smp_parameters_upload(gf_smp *s, const float *host) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!host) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_parameters_upload: null argument");
    gf_status st = gf::own_model(s);
    if (st != GF_OK) return st;
    GF_HIP_TRY(s->ctx, hipMemcpyAsync(s->own_p, host, gf::param_count(s->ucfg) * sizeof(float), hipMemcpyHostToDevice, s->ctx->stream));
    GF_HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return GF_OK;
}

gf_status gf_smp_parameters_download(gf_smp *s, float *host_params, float *host_grads) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!s->own_p) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_parameters_download: no handle-owned model");
    const size_t bytes = gf::param_count(s->ucfg) * sizeof(float);
    if (host_params) GF_HIP_TRY(s->ctx, hipMemcpyAsync(host_params, s->own_p, bytes, hipMemcpyDeviceToHost, s->ctx->stream));
    if (host_grads) GF_HIP_TRY(s->ctx, hipMemcpyAsync(host_grads, s->own_g, bytes, hipMemcpyDeviceToHost, s->ctx->stream));
    GF_HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return GF_OK;
}

gf_status gf_smp_forward_host(gf_smp *s, const double *targets, double *predict, double *loss, double *graph_feature) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!s->prepared) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward_host before gf_smp_prepare");
    if (!s->own_p) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward_host: no handle-owned model (gf_smp_parameters_upload)");
    const int nMol = s->lay.nMol, C = s->ucfg.nChanels;
    gf_status st;
    if (!s->own_y) {
        st = gf::upload(s, &s->own_t, nullptr, (size_t)nMol);
        if (st != GF_OK) return st;
        st = gf::upload(s, &s->own_y, nullptr, (size_t)nMol);
        if (st != GF_OK) return st;
        st = gf::upload(s, &s->own_loss, nullptr, (size_t)nMol);
        if (st != GF_OK) return st;
        st = gf::upload(s, &s->own_feat, nullptr, (size_t)nMol * C);
        if (st != GF_OK) return st;
    }
    std::vector<float> tmp((size_t)nMol * (C > 1 ? C : 1));
    if (targets) {
        for (int m = 0; m < nMol; ++m) tmp[m] = (float)targets[m];
        GF_HIP_TRY(ctx, hipMemcpyAsync(s->own_t, tmp.data(), sizeof(float) * nMol, hipMemcpyHostToDevice, ctx->stream));
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // tmp is reused below
    }
    st = gf_smp_forward(s, s->own_p, targets ? s->own_t : nullptr, s->own_y, s->own_loss, s->own_feat);
    if (st != GF_OK) return st;
    struct Out { double *dst; const float *src; size_t n; } outs[3] = {
        {predict, s->own_y, (size_t)nMol}, {targets ? loss : nullptr, s->own_loss, (size_t)nMol}, {graph_feature, s->own_feat, (size_t)nMol * C}};
    for (const Out &o : outs) {
        if (!o.dst) continue;
        GF_HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), o.src, sizeof(float) * o.n, hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < o.n; ++i) o.dst[i] = (double)tmp[i];
    }
    return GF_OK;
}

// One optimiser step of SMP_omega::BatchLearn (SMP_omega.h:820-821): grads hold the SUM over the batch (gf_smp_backward),
// Adam::Learn(learning_rate, nBatch) divides by nBatch.  Defaults of Adam.h:26-29: beta1 0.9, beta2 0.999, epsilon 1e-8.
gf_status gf_smp_adam_step(gf_smp *s, float *params, const float *grads, double learning_rate, int nBatch) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!params && !grads && s->own_p) {
        params = s->own_p;
        grads = s->own_g;
    }
    if (!params || !grads || nBatch <= 0) return fail(ctx, GF_ERR_INVALID, "gf_smp_adam_step: bad argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = gf::param_count(s->ucfg);
    if (!s->adam_m) {
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->adam_m), n * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->adam_v), n * sizeof(float)));
        GF_HIP_TRY(ctx, hipMemsetAsync(s->adam_m, 0, n * sizeof(float), ctx->stream));
        GF_HIP_TRY(ctx, hipMemsetAsync(s->adam_v, 0, n * sizeof(float), ctx->stream));
        s->adam_n = 0;
    }
    GF_LAUNCH(ctx, "smp_adam", gf::adam_step, dim3(gf::grid_for(n)), dim3(256), 0, params, grads, s->adam_m, s->adam_v, n,
              learning_rate, 1.0 / (double)nBatch, s->adam_n, 0.9, 0.999, 1e-8);
    s->adam_n += n;   // (touches the moment buffers only, not the batch's: the handle's next gf_smp_prepare need not wait for it)
    return GF_OK;
}

// The optimiser of the SMP_2D_ver6-8 models (sgd = new Momentum(momentum_param), SMP_2D_ver6.h:204).  Shares the handle's
// first moment buffer with Adam: a model uses one optimiser or the other.
gf_status gf_smp_momentum_step(gf_smp *s, float *params, const float *grads, double learning_rate, int nBatch, double gamma) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!params && !grads && s->own_p) {
        params = s->own_p;
        grads = s->own_g;
    }
    if (!params || !grads || nBatch <= 0) return fail(ctx, GF_ERR_INVALID, "gf_smp_momentum_step: bad argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = gf::param_count(s->ucfg);
    if (!s->adam_m) {
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->adam_m), n * sizeof(float)));
        GF_HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&s->adam_v), n * sizeof(float)));
        GF_HIP_TRY(ctx, hipMemsetAsync(s->adam_m, 0, n * sizeof(float), ctx->stream));
        GF_HIP_TRY(ctx, hipMemsetAsync(s->adam_v, 0, n * sizeof(float), ctx->stream));
        s->adam_n = 0;
    }
    GF_LAUNCH(ctx, "smp_momentum", gf::momentum_step, dim3(gf::grid_for(n)), dim3(256), 0, params, grads, s->adam_m, n,
              learning_rate, 1.0 / (double)nBatch, gamma);
    return GF_OK;
}

// Adam::Learn(learning_rate, nBatch) (GraphFlow/Adam.h:106-133) on any flat parameter buffer with caller-owned moments:
// element i uses the bias-correction powers beta^(elements_before + i + 1) (the reference advances them per element).
gf_status gf_adam_step_f32(gf_ctx *ctx, float *params, const float *grads, float *m, float *v, size_t n, double learning_rate,
                           int nBatch, unsigned long long elements_before) {
    if (!ctx) return fail(nullptr, GF_ERR_INVALID, "null context");
    if (!params || !grads || !m || !v || nBatch <= 0) return fail(ctx, GF_ERR_INVALID, "gf_adam_step_f32: bad argument");
    if (n == 0) return GF_OK;
    GF_LAUNCH(ctx, "smp_adam", gf::adam_step, dim3(gf::grid_for(n)), dim3(256), 0, params, grads, m, v, n, learning_rate,
              1.0 / (double)nBatch, elements_before, 0.9, 0.999, 1e-8);
    return GF_OK;
}

gf_status gf_smp_adam_reset(gf_smp *s) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (s->adam_m) {
        const size_t n = gf::param_count(s->ucfg);
        GF_HIP_TRY(s->ctx, hipMemsetAsync(s->adam_m, 0, n * sizeof(float), s->ctx->stream));
        GF_HIP_TRY(s->ctx, hipMemsetAsync(s->adam_v, 0, n * sizeof(float), s->ctx->stream));
    }
    s->adam_n = 0;
    return GF_OK;
}

// SMP_omega::weights_initialization (SMP_omega.h:334-338) = GraphFlow::uniform_init (GraphFlow.h:1297-1306) over the
// parameters in registration order, drawn from the C library's rand() exactly as the reference draws them: after the
// same srand() a model built here starts from the same weights as one built by the reference.  Host buffer.
gf_status gf_smp_uniform_init_host(const gf_smp_config *cfg, float *params) {
    if (!cfg || !params) return GF_ERR_INVALID;
    gfsmp::Config c = {cfg->nLevels, cfg->nChanels, cfg->nFeatures, cfg->nDepth, cfg->max_receptive_field, cfg->has_WL_ordering};
    c.nContractions = cfg->nContractions ? cfg->nContractions : 18;
    c.physics = cfg->physics ? 1 : 0;
    const size_t C = (size_t)c.nChanels;
    std::vector<size_t> sizes;
    sizes.push_back(C * c.fdim());
    for (int l = 1; l <= c.nLevels; ++l) {
        sizes.push_back((size_t)c.nContractions * c.level_channels(l - 1) * c.level_channels(l));
        sizes.push_back((size_t)c.level_channels(l));
    }
    if (!c.physics) sizes.push_back(C);
    size_t off = 0;
    for (size_t v = 0; v < sizes.size(); ++v)
        for (size_t i = 0; i < sizes[v]; ++i) {
            double x = (double)(rand() % 10) / (10.0 * (double)sizes[v]);
            if (rand() % 2 == 1) x = -x;
            params[off++] = (float)x;
        }
    return GF_OK;
}

// Text checkpoints in the reference's format (SMP_omega.h:1033-1042 / :1044-1055): every parameter value in
// registration order, printed with the default ostream format (= "%g", 6 significant digits) followed by one blank.
gf_status gf_smp_save_model(const gf_smp *s, const float *params, const char *path) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp");
    if (!params) params = s->own_p;
    if (!params || !path) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_save_model: null argument");
    const size_t n = gf::param_count(s->ucfg);
    std::vector<float> host(n);
    GF_HIP_TRY(s->ctx, hipMemcpyAsync(host.data(), params, n * sizeof(float), hipMemcpyDeviceToHost, s->ctx->stream));
    GF_HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_save_model: cannot open %s", path);
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) ok = std::fprintf(f, "%g ", (double)host[i]) > 0;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? GF_OK : fail(s->ctx, GF_ERR_INVALID, "gf_smp_save_model: write to %s failed", path);
}

gf_status gf_smp_load_model(gf_smp *s, float *params, const char *path) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp");
    if (!params) {
        gf_status st0 = gf::own_model(s);
        if (st0 != GF_OK) return st0;
        params = s->own_p;
    }
    if (!params || !path) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_load_model: null argument");
    const size_t n = gf::param_count(s->ucfg);
    FILE *f = std::fopen(path, "r");
    if (!f) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_load_model: cannot open %s", path);
    std::vector<float> host(n);
    size_t got = 0;
    double v;
    while (got < n && std::fscanf(f, "%lf", &v) == 1) host[got++] = (float)v;
    std::fclose(f);
    // the reference would silently keep reading garbage; a short file is an error here
    if (got != n) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_load_model: %s holds %zu values, the model has %zu", path, got, n);
    GF_HIP_TRY(s->ctx, hipMemcpyAsync(params, host.data(), n * sizeof(float), hipMemcpyHostToDevice, s->ctx->stream));
    GF_HIP_TRY(s->ctx, hipStreamSynchronize(s->ctx->stream));
    return GF_OK;
}

gf_status gf_smp_prepare(gf_smp *s, int nMol, const int *nVertices, const int *adj, const double *feature) {
    return gf_smp_prepare_coulomb(s, nMol, nVertices, adj, feature, nullptr);
}

gf_status gf_smp_prepare_coulomb(gf_smp *s, int nMol, const int *nVertices, const int *adj, const double *feature,
                                 const double *coulomb) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (nMol <= 0 || !nVertices || !adj || !feature) return fail(ctx, GF_ERR_INVALID, "gf_smp_prepare: bad argument");
    for (int m = 0; m < nMol; ++m)
        if (nVertices[m] <= 0 || nVertices[m] > 4096) return fail(ctx, GF_ERR_INVALID, "molecule %d has %d vertices", m, nVertices[m]);
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (s->dup_channels || s->embed_auto_off) {
        // SMP_2D_ver6 / ver7 on the 18-slice level: the identities behind the embedding need a symmetric, non-negative adjacency (row sums =
        // column sums), for `_50` the unit diagonal of a reduced adjacency (cases 25, 41, 42, 45: no Coulomb mode), and in Coulomb mode
        // positive entries (RisiContraction_18 drops A <= 0 -- its `if (adj_value > 0)` -- and RisiContraction_10 does not).  A batch that
        // does not qualify runs on the op-by-op `_10` / `_50` levels, which take anything the reference takes; the plan is per batch.
        const bool is50 = s->ucfg.nContractions == 50;
        bool embeddable = !(is50 && coulomb);
        const int *a = adj;
        const double *cm = coulomb;
        for (int m = 0; m < nMol && embeddable; ++m) {
            const int V = nVertices[m];
            if (cm)
                for (int i = 0; i < V * V && embeddable; ++i) embeddable = cm[i] > 0.0;
            for (int i = 0; i < V && embeddable; ++i)
                for (int j = i + 1; j < V; ++j)
                    if (a[i * V + j] < 0 || a[i * V + j] != a[j * V + i] || (cm && cm[i * V + j] != cm[j * V + i])) {
                        embeddable = false;
                        break;
                    }
            a += (size_t)V * V;
            if (cm) cm += (size_t)V * V;
        }
        if (embeddable != (s->dup_channels != 0)) {
            gf_status stp = gf::smp_switch_plan(s, embeddable);
            if (stp != GF_OK) return stp;
        }
    }
    const bool prep_timing = std::getenv("GF_PREP_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    if (!s->upload) {
        // the batch's uploads and table-building kernels run at the LOWEST stream priority: in the loop with a new batch every step they
        // share the device with the running step of another handle, which is what the loop waits for (GF_PREP_PRIORITY=0: default priority)
        int least = 0, greatest = 0;
        const char *pe = std::getenv("GF_PREP_PRIORITY");
        if (!(pe && pe[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            if (hipStreamCreateWithPriority(&s->upload, hipStreamNonBlocking, least) != hipSuccess) s->upload = nullptr;
        } else if (hipStreamCreateWithFlags(&s->upload, hipStreamNonBlocking) != hipSuccess) {
            s->upload = nullptr;
        }
        if (s->upload && hipEventCreateWithFlags(&s->ev_last, hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(s->upload);
            s->upload = nullptr;
            s->ev_last = nullptr;
        }
    }
    gf::release(s);
    const auto tp1 = std::chrono::steady_clock::now();
    // rows-sized level tables on the device (GF_PREP_DEVICE_TABLES=0: on the host, as rounds 1-2 built them; the parity tests hold
    // the two against each other bit for bit)
    {
        const char *e = std::getenv("GF_PREP_DEVICE_TABLES");
        s->lay.device_tables = !(e && e[0] == '0');
        // build_level_rows keeps three ints per field position and four shorts per vertex of its molecule in LDS: beyond the
        // default 32 KiB window (a molecule of ~4,000 vertices) the tables are built on the host, as rounds 1-2 built all of them
        // (decided BEFORE build_batch lays the batch out for one builder or the other; round-3 advice)
        int vmax = 0;
        for (int m = 0; m < nMol; ++m) vmax = nVertices[m] > vmax ? nVertices[m] : vmax;
        const int smax = s->cfg.max_receptive_field > 0 && s->cfg.max_receptive_field < vmax ? s->cfg.max_receptive_field : vmax;
        if (sizeof(int) * (size_t)smax * 3 + sizeof(short) * 4 * (size_t)vmax + 16 > 32 * 1024) s->lay.device_tables = false;
    }
    s->tab_stats = nullptr;
    s->h_tab_stats.clear();
    s->h_covered.clear();
    gfsmp::build_batch(s->cfg, nMol, nVertices, adj, feature, coulomb, &s->lay);
    const auto tp2 = std::chrono::steady_clock::now();
    const gfsmp::BatchLayout &B = s->lay;
    const int L = s->cfg.nLevels, C = s->cfg.nChanels;
    s->lv.assign(L + 1, gf_smp::DevLevel());
    gf_status st;
#define UP(dst, vec)                                                         \
    st = gf::upload(s, &(dst), (vec).empty() ? nullptr : &(vec)[0], (vec).size()); \
    if (st != GF_OK) return st;
    static_assert(sizeof(long long) == sizeof(int64_t), "int64 layout");
    long long maxp = 0;
    size_t contract_ws = 0;
    for (int l = 0; l <= L; ++l) {
        const gfsmp::LevelLayout &h = B.level[l];
        gf_smp::DevLevel &d = s->lv[l];
        UP(d.node_s, h.node_s);
        UP(d.node_center, h.node_center);
        UP(d.mol_order, h.mol_order);
        UP(d.gather_items, h.gather_items);
        if (B.device_tables) {
            UP(d.field, h.field);
            if (!s->cfg.physics) UP(d.node_mol, h.node_mol);   // (the towers upload it below)
        }
        if (l >= 1 && s->cfg.square()) {
            st = gf::upload(s, &d.tf_recs, nullptr, (size_t)h.nNodes * 2);
            if (st != GF_OK) return st;
        }
        st = gf::upload(s, &d.node_row, &h.node_row[0], h.node_row.size());
        if (st != GF_OK) return st;
        st = gf::upload(s, &d.node_pair, &h.node_pair[0], h.node_pair.size());
        if (st != GF_OK) return st;
        const int Cl = s->cfg.level_channels(l), Cp = l ? s->cfg.level_channels(l - 1) : Cl;
        st = gf::upload(s, &d.f, nullptr, (size_t)h.rows * Cl);
        if (st != GF_OK) return st;
        st = gf::upload(s, &d.df, nullptr, (size_t)h.rows * Cl);
        if (st != GF_OK) return st;
        if (s->cfg.physics) {  // every level is read out: per-node sums, their activation and gradient, the vertex -> node map
            st = gf::upload(s, &d.sh, nullptr, (size_t)h.nNodes * Cl);
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.vf, nullptr, (size_t)h.nNodes * Cl);
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.dshl, nullptr, (size_t)h.nNodes * Cl);
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.node_of_vertex, &B.node_of_vertex[l][0], B.node_of_vertex[l].size());
            if (st != GF_OK) return st;
            UP(d.node_mol, h.node_mol);
            st = gf::upload(s, &d.keep_mask, nullptr, (size_t)h.nNodes);
            if (st != GF_OK) return st;
        }
        if (l == 0) continue;
        st = gf::upload(s, &d.node_p, &h.node_p[0], h.node_p.size());
        if (st != GF_OK) return st;
        d.max_tot = d.max_tr = 0.f;
        if (B.device_tables) {  // built by build_level_rows below
            st = gf::upload(s, &d.adj, nullptr, (size_t)h.rows);
            if (st == GF_OK) st = gf::upload(s, &d.rsum, nullptr, (size_t)h.pairs);
            if (st == GF_OK) st = gf::upload(s, &d.node_scale, nullptr, (size_t)h.nNodes * 2);
            if (st == GF_OK) st = gf::upload(s, &d.node_present, nullptr, (size_t)h.nNodes);
            if (st != GF_OK) return st;
        } else {
        UP(d.adj, h.adj);
        UP(d.rsum, h.rsum);
        UP(d.node_scale, h.rowscale);
        for (size_t i = 0; i + 1 < h.rowscale.size(); i += 2) {
            d.max_tot = std::max(d.max_tot, std::fabs(h.rowscale[i]));
            d.max_tr = std::max(d.max_tr, std::fabs(h.rowscale[i + 1]));
        }
        }
        st = gf::upload(s, &d.rowscale, nullptr, (size_t)h.rows * 2);
        if (st != GF_OK) return st;
        UP(d.quad_node, h.quad_node);
        UP(d.quad_b0, h.quad_b0);
        UP(d.quad_order, h.quad_order);
        UP(d.pair_node, h.pair_node);
        UP(d.pair_src_s, h.pair_src_s);
        if (B.device_tables) {   // (filled by build_consumer_entries below)
            st = gf::upload(s, &d.cons_s, nullptr, (size_t)h.pairs);
            if (st == GF_OK) st = gf::upload(s, &d.cons_a, nullptr, (size_t)h.pairs);
            if (st != GF_OK) return st;
        } else {
            UP(d.cons_s, h.cons_s);
            UP(d.cons_a, h.cons_a);
        }
        st = gf::upload(s, &d.pair_src_pair, &h.pair_src_pair[0], h.pair_src_pair.size());
        if (st != GF_OK) return st;
        st = B.device_tables ? gf::upload(s, &d.cons_row, nullptr, (size_t)h.pairs)
                             : gf::upload(s, &d.cons_row, h.cons_row.empty() ? nullptr : &h.cons_row[0], h.cons_row.size());
        if (st != GF_OK) return st;
        st = gf::upload(s, &d.cons_pair, h.cons_pair.empty() ? nullptr : &h.cons_pair[0], h.cons_pair.size());
        if (st != GF_OK) return st;
        {
            float **cb[] = {&d.Fdc, &d.Gc, &d.dGc, &d.dFdc};
            for (int q = 0; q < 4; ++q) {
                st = gf::upload(s, cb[q], nullptr, (size_t)B.level[l - 1].pairs * 2 * C);
                if (st != GF_OK) return st;
            }
        }
        st = gf::upload(s, &d.pair_src_row, &h.pair_src_row[0], h.pair_src_row.size());
        if (st != GF_OK) return st;
        st = gf::upload(s, &d.cons_ptr, &h.cons_ptr[0], h.cons_ptr.size());
        if (st != GF_OK) return st;
        st = B.device_tables ? gf::upload(s, &d.cons_slab, nullptr, (size_t)h.pairs)
                             : gf::upload(s, &d.cons_slab, h.cons_slab.empty() ? nullptr : &h.cons_slab[0], h.cons_slab.size());
        if (st != GF_OK) return st;
        if (B.device_tables && h.pairs) {
            hipStream_t upst = s->upload ? s->upload : ctx->stream;
            hipLaunchKernelGGL(gf::build_consumer_entries, dim3((unsigned)((h.pairs + 255) / 256)), dim3(256), 0, upst, d.cons_pair, d.pair_node, d.node_s,
                               d.node_pair, d.node_row, d.node_p, d.cons_slab, d.cons_s, d.cons_row, d.cons_a, (long long)h.pairs);
            GF_LAUNCH_CHECK(ctx, "build_consumer_entries");
        }
        st = gf::upload(s, &d.cons_inv_off, h.cons_inv_off.empty() ? nullptr : &h.cons_inv_off[0], h.cons_inv_off.size());
        if (st != GF_OK) return st;
        if (B.device_tables) {
            st = gf::upload(s, &d.pi, nullptr, (size_t)h.rows);
            if (st == GF_OK) st = gf::upload(s, &d.inv, nullptr, (size_t)h.inv_count);
            if (st != GF_OK) return st;
        } else {
            UP(d.pi, h.pi);
            UP(d.inv, h.inv);
        }
        if (s->cfg.square() && s->cfg.nContractions == 18 && C % 4 == 0 && s->bwd_gather) {  // fused levels: tables of the gather
            st = gf::upload(s, &d.cons_hdr, nullptr, (size_t)h.pairs * 2);
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.cons_qrec, nullptr, (size_t)h.qrec_total);
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.cons_qbase, h.cons_qbase.empty() ? nullptr : &h.cons_qbase[0], h.cons_qbase.size());
            if (st != GF_OK) return st;
        }
        if (s->cfg.square() && gf::smp_panel_channels(C) && h.rows < 0x7fffffffll) {   // (C = 32: the split row-panel products, round 4; 16: round 5)
            st = gf::upload(s, &d.trow, nullptr, (size_t)h.rows);
            if (st == GF_OK) st = gf::upload(s, &d.trowf, nullptr, (size_t)h.rows);
            if (st == GF_OK) {
                unsigned char *img = nullptr;
                st = gf::upload(s, &img, nullptr, gf::smp_split_image_bytes());
                d.wimg = img;
            }
            if (st != GF_OK) return st;
            st = gf::upload(s, &d.rowflag, nullptr, (size_t)h.rows);
            if (st != GF_OK) return st;
            if (s->cfg.nContractions == 18 && h.rows * 256 < 0x3fffffffll && !h.buckets.empty() && h.buckets.back().s <= gf::kFusedMaxField) {
                // row panels of the fused forward level (smp_level_c64_fwd.hip): a node of size s has ceil(s / max(1, 32 / s)) panels
                const int np = h.npanels;   // (page-locked table of the layout: no wait for the copy)
                const size_t np1 = (size_t)(np > 0 ? np : 1);   // (a level whose nodes are ALL above 32 positions has no panel: the tables exist all the same)
                d.fwd_npanels = np;
                UP(d.node_panel, h.node_panel);
                st = gf::upload(s, &d.fwd_pan, nullptr, np1);
                if (st != GF_OK) return st;
                if (l == L || s->cfg.physics) {   // (the top level -- every level of a tower: the readout's partial sums)
                    st = gf::upload(s, &d.psum, nullptr, np1 * C);
                    if (st != GF_OK) return st;
                }
                if (l < L) {   // (below the top level: the per-panel channel maxima the level above scales its weight-gradient operands with)
                    st = gf::upload(s, &d.pmax, nullptr, np1 * C);
                    if (st != GF_OK) return st;
                }
                st = gf::upload(s, &d.dzmax, nullptr, (h.quad_node.size() + np1) * 64);   // (panels, then the workgroups of the nodes above 32 positions)
                if (st != GF_OK) return st;
                st = gf::upload(s, &d.fwd_pan_node, nullptr, np1);
                if (st != GF_OK) return st;
                st = gf::upload(s, &d.fwd_goff, nullptr, (size_t)h.rows);
                if (st != GF_OK) return st;
            }
        }
        st = gf::upload(s, &d.Q, nullptr, (size_t)h.rows * std::max(18, s->cfg.nContractions) * Cp);
        if (st != GF_OK) return st;
        if (s->cfg.square()) {
            float **bufs[] = {&d.Vt, &d.dVt, &d.St, &d.dSt, &d.scal, &d.Vout, &d.dVout, &d.Sout, &d.dSout, &d.dSpart, &d.dbpart, &d.Wst, &d.dWst};
            const size_t sizes[] = {(size_t)h.pairs * 4 * C, (size_t)h.pairs * 4 * C, (size_t)h.nNodes * 4 * C, (size_t)h.nNodes * 4 * C,
                                    (size_t)h.pairs * 4 * C, (size_t)h.pairs * C, (size_t)h.pairs * C, (size_t)h.nNodes * C,
                                    (size_t)h.nNodes * C, (size_t)h.pairs * C, (size_t)h.pairs * C, (size_t)18 * C * C, (size_t)18 * C * C};
            for (int q = 0; q < 13; ++q) {
                st = gf::upload(s, bufs[q], nullptr, sizes[q]);
                if (st != GF_OK) return st;
            }
        }
        if (h.ppos * Cp > maxp) maxp = h.ppos * Cp;
        for (size_t b = 0; b < h.buckets.size(); ++b) {
            const size_t w = gf_contract_workspace_bytes(s->cfg.nContractions, h.buckets[b].s, Cp, h.buckets[b].count);
            if (w > contract_ws) contract_ws = w;
        }
        contract_ws = std::max(contract_ws, gf::r18_ragged_workspace_bytes((long long)h.rows, (long long)h.pairs, Cp));
    }
    if (B.device_tables) {  // the molecules' adjacency matrices, then the rows-sized tables of every level (kernels above)
        hipStream_t up = s->upload ? s->upload : ctx->stream;
        const int vmax = B.max_vertices;
        UP(s->mol_nv, B.mol_nv);
        UP(s->mol_adj, B.mol_adj);
        st = gf::upload(s, &s->mol_adj_off, &B.mol_adj_off[0], B.mol_adj_off.size());
        s->mol_coul = nullptr;
        if (st == GF_OK && coulomb) st = gf::upload(s, &s->mol_coul, &B.mol_coul[0], B.mol_coul.size());
        if (st == GF_OK) st = gf::upload(s, &s->tab_stats, nullptr, (size_t)4 * (L + 1));  // per level: max |tot|, max |tr| (float bits), rows with data (64 bit)
        if (st != GF_OK) return st;
        GF_HIP_TRY(ctx, hipMemsetAsync(s->tab_stats, 0, sizeof(unsigned) * 4 * (L + 1), up));
        for (int l = 1; l <= L; ++l) {
            const gfsmp::LevelLayout &h = B.level[l];
            gf_smp::DevLevel &d = s->lv[l];
            const int smax = h.buckets.empty() ? 1 : h.buckets.back().s;
            const size_t lds = sizeof(int) * (size_t)smax * 3 + sizeof(short) * 4 * (size_t)vmax + 16;
            if (lds > 32 * 1024)   // (cannot happen: gf_smp_prepare chose the host builder for such a batch)
                return fail(ctx, GF_ERR_UNSUPPORTED, "gf_smp_prepare: a receptive field of %d vertices in a molecule of %d", smax, vmax);
            unsigned *stats = s->tab_stats + 4 * l;
            if (h.inv_count) GF_HIP_TRY(ctx, hipMemsetAsync(d.inv, 0xff, sizeof(short) * (size_t)h.inv_count, up));
            // one kernel for all the node's tables where the fields fit its LDS image and its 32-bit presence masks (build_node_tables)
            const int swp = B.level[l - 1].buckets.empty() ? 1 : B.level[l - 1].buckets.back().s;
            const size_t lds_nt = sizeof(int) * (size_t)smax * (4 + (size_t)swp) + sizeof(short) * (size_t)smax * smax + 16;
            d.node_tables_merged = smax <= 32 && lds_nt <= 32 * 1024 && h.pairs > 0 && h.pairs < 0x7fffffffll;
            if (d.node_tables_merged) {
                st = gf::upload(s, &d.cons_of_pair, nullptr, (size_t)h.pairs);
                if (st != GF_OK) return st;
                hipLaunchKernelGGL(gf::invert_cons_pair, dim3((unsigned)((h.pairs + 255) / 256)), dim3(256), 0, up, d.cons_pair, d.cons_of_pair,
                                   (long long)h.pairs);
                GF_LAUNCH_CHECK(ctx, "invert_cons_pair");
                hipLaunchKernelGGL(gf::build_node_tables, dim3(h.nNodes), dim3(128), lds_nt, up, d.node_s, d.node_mol, d.node_row, d.node_pair,
                                   d.field, s->lv[l - 1].field, d.pair_src_pair, d.pair_src_s, s->mol_nv, s->mol_adj_off, s->mol_adj, s->mol_coul,
                                   d.adj, d.rsum, d.node_scale, d.pi, d.node_present, swp, d.cons_of_pair, d.cons_inv_off, d.inv,
                                   reinterpret_cast<float2 *>(d.rowscale), d.trow, d.rowflag, d.trowf, d.fwd_goff);
                GF_LAUNCH_CHECK(ctx, "build_node_tables");
            } else {
                hipLaunchKernelGGL(gf::build_level_rows, dim3(h.nNodes), dim3(256), lds, up, d.node_s, d.node_mol, d.node_row, d.node_pair, d.field,
                                   s->lv[l - 1].field, d.pair_src_pair, d.pair_src_s, s->mol_nv, s->mol_adj_off, s->mol_adj, s->mol_coul, d.adj,
                                   d.rsum, d.node_scale, d.pi, d.node_present, vmax);
                GF_LAUNCH_CHECK(ctx, "build_level_rows");
                if (h.pairs) {
                    hipLaunchKernelGGL(gf::build_level_inv, dim3((unsigned)((h.pairs + 3) / 4)), dim3(256), 0, up, d.cons_pair, d.pair_node, d.node_s,
                                       d.node_row, d.node_pair, d.cons_inv_off, d.pi, d.inv, (long long)h.pairs);
                    GF_LAUNCH_CHECK(ctx, "build_level_inv");
                }
            }
            hipLaunchKernelGGL(gf::level_table_stats, dim3(1), dim3(1024), 0, up, d.node_scale, d.node_present, h.nNodes, stats);
            GF_LAUNCH_CHECK(ctx, "level_table_stats");
        }
        // (the split-operand weight gradients read a level's largest |tot|, |tr| from the statistics words: no read-back, the
        //  preparing thread does not wait for its uploads)
        for (int l = 1; l <= L; ++l) s->lv[l].row_max = s->tab_stats + 4 * l;
        s->h_tab_stats.clear();
        s->h_covered.clear();
    }
    // (the node tables went up on the handle's upload stream: build the transposed-row tables there too, behind them)
    for (int l = 1; l <= L; ++l) {
        hipStream_t up = s->upload ? s->upload : ctx->stream;
        const bool merged = s->lv[l].node_tables_merged;   // (row factors, transposed-row tables and gather offsets are in place)
        if (!merged)
            hipLaunchKernelGGL(gf::expand_rowscale, dim3(B.level[l].nNodes), dim3(64), 0, up, reinterpret_cast<float2 *>(s->lv[l].rowscale),
                               reinterpret_cast<const float2 *>(s->lv[l].node_scale), s->lv[l].node_s, s->lv[l].node_row);
        st = gf::smp_build_gather_records(s, l, up);
        if (st != GF_OK) return st;
        st = gf::smp_build_tf_records(s, l, up);
        if (st != GF_OK) return st;
        st = gf::smp_fwd_fused_build_tables(s, l, up, !merged);
        if (st != GF_OK) return st;
        if (s->lv[l].trow && !merged)
            hipLaunchKernelGGL(gf::build_trow, dim3(B.level[l].nNodes), dim3(64), 0, up, s->lv[l].trow, s->lv[l].node_s, s->lv[l].node_row,
                               s->lv[l].pi, s->lv[l].rowflag, s->lv[l].trowf);
    }
    UP(s->x, B.x);
    s->P = nullptr;  // [max ppos][C]: by far the largest buffer of the op-by-op path, taken from the pool only when a level needs it
    s->P_count = (size_t)maxp;  // (positions x channels of the level below, maximised over the levels)
    const gfsmp::LevelLayout &top = B.level[L];
    s->wbound = nullptr;
    if (s->cfg.square() && C == 64) {
        st = gf::upload(s, &s->wbound, nullptr, gf::smp_wgrad_bound_words() * (size_t)(L + 1));
        if (st != GF_OK) return st;
    } else if (s->cfg.square() && (C == 32 || C == 16)) {   // scratch words of the C = 32 / 16 weight-gradient kernel's column bounds
        st = gf::upload(s, &s->wbound, nullptr, gf::smp_wgrad_direct_words_c32() * (size_t)(L + 1));
        if (st != GF_OK) return st;
    }
    st = gf::upload(s, &s->sh, nullptr, (size_t)top.nNodes * C);
    if (st != GF_OK) return st;
    st = gf::upload(s, &s->vf, nullptr, (size_t)top.nNodes * C);
    if (st != GF_OK) return st;
    st = gf::upload(s, &s->dsh, nullptr, (size_t)top.nNodes * C);
    if (st != GF_OK) return st;
    st = gf::upload(s, &s->g, nullptr, (size_t)nMol * (s->cfg.physics ? gf::feature_width(s->cfg) : (size_t)C));
    if (st != GF_OK) return st;
    st = gf::upload(s, &s->yhat, nullptr, (size_t)nMol);
    if (st != GF_OK) return st;
    st = gf::upload(s, &s->dy, nullptr, (size_t)nMol);
    if (st != GF_OK) return st;
    UP(s->top_node_mol, top.node_mol);
    std::vector<int> mol_ptr(nMol + 1, 0), mol_nodes(top.nNodes);
    for (int m = 0; m < nMol; ++m) mol_ptr[m + 1] = B.mol_first_vertex[m + 1];
    for (int gv = 0; gv < top.nNodes; ++gv) mol_nodes[gv] = B.top_node_of_vertex[gv];  // vertices of a molecule are contiguous
    UP(s->mol_ptr, mol_ptr);
    UP(s->mol_nodes, mol_nodes);
    long long maxrows = 0;
    for (int l = 0; l <= L; ++l) maxrows = std::max(maxrows, (long long)B.level[l].rows);
    long long maxpairs = 0;
    for (int l = 0; l <= L; ++l) maxpairs = std::max(maxpairs, (long long)B.level[l].pairs);
    s->colpart_rows = (size_t)((maxrows + 1023) / 1024 + (maxpairs + 255) / 256 + 2);
    s->colpart_rows = std::max(s->colpart_rows, (size_t)256 * (L + 1));  // fused levels: up to 256 column partials per level
    st = gf::upload(s, &s->colpart, nullptr, s->colpart_rows * C);
    if (st != GF_OK) return st;
#undef UP
    // split-K partials of the weight gradients also live in the context workspace
    const size_t gemm_ws = sizeof(float) * 4400 * (size_t)4 * C * C + sizeof(float) * 4400 * (size_t)C * s->cfg.fdim() + (1 << 20);
    s->ws_need = std::max(contract_ws, gemm_ws);  // grown by forward / backward on the compute thread (smp_internal.h)
    // the tables are on the device when this returns (the host vectors are reused by the next batch); the context's stream
    // is NOT waited for: it may be running another handle's step
    GF_HIP_TRY(ctx, hipStreamSynchronize(s->upload ? s->upload : ctx->stream));
    if (prep_timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        std::fprintf(stderr, "gf_smp_prepare: release %.1f ms, host graph preparation %.1f ms, device allocation + upload %.1f ms\n",
                     ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3));
    }
    s->prepared = true;
    return GF_OK;
}

// ---- channel padding at the C ABI (gf_smp_create: cfg = what the device computes with, ucfg = the caller's layout) ----------------
namespace gf {
// The two parameter layouts.  Order H [C_0][FD], (K_l, b_l)..., W [C] (no W in a physics tower); K_l is [18][C_{l-1}][C_l] as (k, ci, co)
// (SMP_omega.h:289-295; C_l = C, or halving per level in a tower) or, custom_matmul, [C][18 C] as (co, k, ci) (CustomMatMulTensor,
// SMP_2D_ver8).  The padded layout has Cc channels at every level.
struct PadMap {
    int L, nK, custom, FD, Cc, hasW;
    int cu[kPadMaxLevels + 1];          // the caller's channels of level l
    long long uoff[kPadMaxLevels + 2];  // the caller's offset of H (0), K_1, ..., K_L, W
};
static PadMap pad_map(const gfsmp::Config &u, const gfsmp::Config &c) {
    PadMap m = {};
    m.L = u.nLevels, m.nK = u.nContractions, m.custom = u.custom_matmul, m.FD = u.fdim(), m.Cc = c.nChanels, m.hasW = u.physics ? 0 : 1;
    for (int l = 0; l <= m.L; ++l) m.cu[l] = u.level_channels(l);
    m.uoff[0] = 0;
    m.uoff[1] = (long long)m.cu[0] * m.FD;
    for (int l = 1; l <= m.L; ++l) m.uoff[l + 1] = m.uoff[l] + (long long)m.nK * m.cu[l - 1] * m.cu[l] + m.cu[l];
    return m;
}
// element i of the PADDED parameter vector -> its place in the caller's, or -1 (a padded weight: zero)
__device__ __forceinline__ long long padded_to_user(long long i, const PadMap &m) {
    const int Cc = m.Cc;
    const long long hpad = (long long)Cc * m.FD;
    if (i < hpad) {
        const int c = (int)(i / m.FD);
        return c < m.cu[0] ? i : -1;   // (same index: rows c < C_0 come first in both layouts)
    }
    i -= hpad;
    const long long lvl_pad = (long long)m.nK * Cc * Cc + Cc;
    const long long lq = i / lvl_pad;
    if (lq < m.L) {
        const int l = (int)lq + 1, Ci = m.cu[l - 1], Co = m.cu[l];
        const long long j = i - lq * lvl_pad, base = m.uoff[l];
        if (j >= (long long)m.nK * Cc * Cc) {   // bias
            const long long c = j - (long long)m.nK * Cc * Cc;
            return c < Co ? base + (long long)m.nK * Ci * Co + c : -1;
        }
        int k, ci, co;
        if (m.custom) {
            co = (int)(j / ((long long)m.nK * Cc));
            const long long r = j % ((long long)m.nK * Cc);
            k = (int)(r / Cc), ci = (int)(r % Cc);
            return (co < Co && ci < Ci) ? base + (long long)co * m.nK * Ci + (long long)k * Ci + ci : -1;
        }
        k = (int)(j / ((long long)Cc * Cc));
        const long long r = j % ((long long)Cc * Cc);
        ci = (int)(r / Cc), co = (int)(r % Cc);
        return (ci < Ci && co < Co) ? base + ((long long)k * Ci + ci) * Co + co : -1;
    }
    const long long c = i - (long long)m.L * lvl_pad;   // W
    return (m.hasW && c < m.cu[m.L]) ? m.uoff[m.L + 1] + c : -1;
}
__global__ void pad_parameters(const float *__restrict__ user, float *__restrict__ padded, long long n_padded, PadMap m) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    const long long u = padded_to_user(i, m);
    padded[i] = u >= 0 ? user[u] : 0.f;
}
__global__ void crop_gradients(const float *__restrict__ padded, float *__restrict__ user, long long n_padded, PadMap m, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    const long long u = padded_to_user(i, m);
    if (u >= 0) user[u] = accumulate ? user[u] + padded[i] : padded[i];
}
// ---- SMP_2D_ver6 on the 18-slice level (gf_smp::dup_channels) --------------------------------------------------------------------
// RisiContraction_10's slice k (RisiContraction_10.h:94-142 = cases 1..10 of RisiContraction_50.h) as (slot of RisiContraction_18, input
// group 0 = f, 1 = f^T), for a SYMMETRIC reduced adjacency (row sums = column sums; checked numerically against the oracle, all 18 slots
// on P and on P with b and c swapped):
//   1 (a,b) -> slot 0 on f        2 (a,c) -> slot 0 on f^T      3 (a,d), 4 (a,e) -> slot 1 on f       5 (b,c) -> slot 2 on f
//   6 (b,d), 7 (b,e) -> slot 3 on f          8 (c,d), 9 (c,e) -> slot 3 on f^T          10 (d,e) -> slot 4 on f
// Two slices that share a slot share its padded weight block: the block holds their SUM (the level is linear in K), and both receive
// the block's gradient.
__device__ __forceinline__ void v6_slot(int k, int *slot, int *grp) {
    const int sl[10] = {0, 0, 1, 1, 2, 3, 3, 3, 3, 4}, gr[10] = {0, 1, 0, 0, 0, 0, 0, 1, 1, 0};
    *slot = sl[k];
    *grp = gr[k];
}
// RisiContraction_50's cases 1..50 (RisiContraction_50.h:94-430) the same way; slots 18, 19, 20 = the extra products (S_ab, 1), (S_bc, 1),
// (S_bc, tr) of gf_smp::n_extra (cases 41 / 42, 45, 25 with the reduced adjacency's unit diagonal).  At most two cases share a block.
__device__ __forceinline__ void v7_slot(int k, int *slot, int *grp) {
    const signed char sl[50] = {0, 0, 1, 1, 2, 3, 3, 3, 3, 4,   5, 5, 6, 5, 5, 6, 7, 8, 8, 7,   8, 8, 9, 9, 20, 10, 11, 12, 10, 11,
                                12, 10, 11, 12, 10, 11, 12, 13, 13, 14,   18, 18, 15, 15, 19, 16, 16, 16, 16, 17};
    const signed char gr[50] = {0, 1, 0, 0, 0, 0, 0, 1, 1, 0,   0, 0, 0, 1, 1, 1, 0, 0, 1, 0,   0, 1, 0, 0, 0, 0, 0, 0, 0, 0,
                                0, 1, 1, 1, 1, 1, 1, 0, 1, 0,   0, 1, 0, 0, 0, 0, 0, 1, 1, 0};
    *slot = sl[k];
    *grp = gr[k];
}
// the caller's parameter u -> its (only) place in the padded [H | (K_l [18 Cc][Cc], b_l [Cc]) x L | W | X_1 .. X_L] vector
__device__ __forceinline__ long long v6_user_to_padded(long long u, const PadMap &m) {
    const int C = m.cu[0], Cc = m.Cc, nK = m.nK;
    if (u < m.uoff[1]) return u;   // H: rows c < C first in both layouts
    const long long hpad = (long long)Cc * m.FD, lvl_pad = 18ll * Cc * Cc + Cc;
    for (int l = 1; l <= m.L; ++l) {
        if (u >= m.uoff[l + 1]) continue;
        const long long j = u - m.uoff[l], base = hpad + (l - 1) * lvl_pad;
        if (j >= (long long)nK * C * C) return base + 18ll * Cc * Cc + (j - (long long)nK * C * C);   // bias
        int k, ci, co;
        if (m.custom) {   // [C][nK C]
            co = (int)(j / (nK * C));
            const int r = (int)(j % (nK * C));
            k = r / C, ci = r % C;
        } else {          // [nK C][C]
            k = (int)(j / ((long long)C * C));
            const int r = (int)(j % ((long long)C * C));
            ci = r / C, co = r % C;
        }
        int slot, grp;
        if (nK == 10) v6_slot(k, &slot, &grp);
        else v7_slot(k, &slot, &grp);
        if (slot >= 18)   // an extra product's block
            return hpad + m.L * lvl_pad + Cc + ((long long)(l - 1) * 3 + (slot - 18)) * Cc * Cc + (long long)(grp * C + ci) * Cc + co;
        return base + ((long long)slot * Cc + grp * C + ci) * Cc + co;
    }
    return hpad + m.L * lvl_pad + (u - m.uoff[m.L + 1]);   // W
}
__global__ void v6_pad_parameters(const float *__restrict__ user, float *__restrict__ padded, long long n_user, PadMap m) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_user) return;
    atomicAdd(padded + v6_user_to_padded(u, m), user[u]);   // (padded starts at zero; at most two terms per entry: the order cannot matter)
}
__global__ void v6_crop_gradients(const float *__restrict__ padded, float *__restrict__ user, long long n_user, PadMap m, int accumulate) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_user) return;
    const float g = padded[v6_user_to_padded(u, m)];
    user[u] = accumulate ? user[u] + g : g;
}
// f [rows][Cc]: channels [C, 2C) of row (x, y) <- channels [0, C) of row (y, x) of the same node (trow; null: level 0, one row per node);
// pmax [panels][Cc] (or null): the per-panel channel maxima combine-forward left, copied likewise (a level-wide maximum is all they serve)
__global__ void dup_transposed_channels(float *__restrict__ f, const int *__restrict__ trow, long long rows, int C, int Cc, float *__restrict__ pmax,
                                        long long panels) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const long long t = trow ? trow[r] : r;
        f[r * Cc + C + c] = f[t * Cc + c];
    } else if (pmax && i < rows * C + panels * C) {
        const long long j = i - rows * C;
        pmax[(j / C) * Cc + C + j % C] = pmax[(j / C) * Cc + j % C];
    }
}
// the reverse: df[(x, y)][c] += df[(y, x)][C + c], and the upper channels (read exactly once, by this thread) are cleared
__global__ void fold_transposed_channels(float *__restrict__ df, const int *__restrict__ trow, long long rows, int C, int Cc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C;
    const int c = (int)(i % C);
    const long long t = trow ? trow[r] : r;
    df[r * Cc + c] += df[t * Cc + C + c];
    df[t * Cc + C + c] = 0.f;
}
static gf_status dup_level(gf_smp *s, int l) {
    if (!s->dup_channels) return GF_OK;
    const gf_smp::DevLevel &d = s->lv[l];
    const long long rows = l == 0 ? s->lay.level[0].nNodes : s->lay.level[l].rows;
    if (l > 0 && !d.trow) return fail(s->ctx, GF_ERR_UNSUPPORTED, "SMP_2D_ver6 on the fused level: level %d has no transposed-row table", l);
    const bool pm = l > 0 && d.pmax && d.pmax_ready;
    const long long panels = pm ? d.fwd_npanels : 0, n = (rows + panels) * s->dup_channels;
    GF_LAUNCH(s->ctx, "smp_dup_transposed", dup_transposed_channels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d.f, l == 0 ? (const int *)nullptr : d.trow,
              rows, s->dup_channels, s->cfg.nChanels, pm ? d.pmax : (float *)nullptr, panels);
    return GF_OK;
}
static gf_status fold_level(gf_smp *s, int l) {
    if (!s->dup_channels) return GF_OK;
    const gf_smp::DevLevel &d = s->lv[l];
    const long long rows = l == 0 ? s->lay.level[0].nNodes : s->lay.level[l].rows, n = rows * s->dup_channels;
    GF_LAUNCH(s->ctx, "smp_fold_transposed", fold_transposed_channels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d.df,
              l == 0 ? (const int *)nullptr : d.trow, rows, s->dup_channels, s->cfg.nChanels);
    return GF_OK;
}
// ---- the extra products of SMP_2D_ver7 (gf_smp::n_extra) on an OP-BY-OP level: the tables are slices of Q -- slice 0 = tot S_ab, slice 2 =
// tot S_bc (RisiContraction_18's cases 1 and 5) -- so the products take the row factors (1 / tot, tr / tot)
__global__ void invert_rowscale(const float2 *__restrict__ rowscale, float2 *__restrict__ inv, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const float2 v = rowscale[i];
    inv[i] = make_float2(1.f / v.x, v.y / v.x);   // (tot >= the trace >= 1: a reduced adjacency has a unit diagonal)
}
static gf_status extra_rs_inv(gf_smp *s, int l) {
    const long long rows = s->lay.level[l].rows;
    if (s->rs_inv_rows < (size_t)rows) {
        if (s->rs_inv) (void)hipFree(s->rs_inv);
        s->rs_inv = nullptr;
        s->rs_inv_rows = 0;
        GF_HIP_TRY(s->ctx, hipMalloc(reinterpret_cast<void **>(&s->rs_inv), sizeof(float) * 2 * (size_t)rows));
        s->rs_inv_rows = (size_t)rows;
    }
    GF_LAUNCH(s->ctx, "smp_extra_rowscale", invert_rowscale, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
              reinterpret_cast<const float2 *>(s->lv[l].rowscale), reinterpret_cast<float2 *>(s->rs_inv), rows);
    return GF_OK;
}
// forward: f_l (before bias and LeakyReLU) += (Q_0 / tot) X_a + (Q_2 / tot) X_b + (tr Q_2 / tot) X_c
static gf_status extra_products_forward(gf_smp *s, int l) {
    if (!s->n_extra) return GF_OK;
    gf_ctx *ctx = s->ctx;
    if (!s->extra_w) return fail(ctx, GF_ERR_INVALID, "level %d: the extra products' weights are not bound", l);
    gf_status st = extra_rs_inv(s, l);
    if (st != GF_OK) return st;
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels, KC = 18 * C, rows = (int)s->lay.level[l].rows;
    const float *X = s->extra_w + (size_t)(l - 1) * 3 * C * C;
    st = gemm_rs(ctx, false, false, rows, C, C, d.Q, KC, 0, X, C, 0, d.f, C, 0, 1, 1, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, false, false, rows, C, C, d.Q + 2 * C, KC, 0, X + (size_t)C * C, C, 0, d.f, C, 0, 1, 1, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, false, false, rows, C, C, d.Q + 2 * C, KC, 0, X + 2 * (size_t)C * C, C, 0, d.f, C, 0, 1, 1, s->rs_inv, 2, 1);
    return st;
}
// backward, first half (Q still holds the forward's slices, d.df = dZ): dX
static gf_status extra_products_wgrad(gf_smp *s, int l) {
    if (!s->n_extra) return GF_OK;
    gf_ctx *ctx = s->ctx;
    if (!s->extra_w || !s->extra_g) return fail(ctx, GF_ERR_INVALID, "level %d: the extra products' weights are not bound", l);
    gf_status st = extra_rs_inv(s, l);
    if (st != GF_OK) return st;
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels, KC = 18 * C, rows = (int)s->lay.level[l].rows;
    float *dX = s->extra_g + (size_t)(l - 1) * 3 * C * C;
    st = gemm_rs(ctx, true, false, C, C, rows, d.Q, KC, 0, d.df, C, 0, dX, C, 0, 1, 0, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, true, false, C, C, rows, d.Q + 2 * C, KC, 0, d.df, C, 0, dX + (size_t)C * C, C, 0, 1, 0, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, true, false, C, C, rows, d.Q + 2 * C, KC, 0, d.df, C, 0, dX + 2 * (size_t)C * C, C, 0, 1, 0, s->rs_inv, 2, 1);
    return st;
}
// ... second half (Q now holds dQ): dQ_0 += (dZ / tot) X_a^T, dQ_2 += (dZ / tot) X_b^T + (tr dZ / tot) X_c^T
static gf_status extra_products_backward(gf_smp *s, int l) {
    if (!s->n_extra) return GF_OK;
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels, KC = 18 * C, rows = (int)s->lay.level[l].rows;
    const float *X = s->extra_w + (size_t)(l - 1) * 3 * C * C;
    gf_status st = gemm_rs(ctx, false, true, rows, C, C, d.df, C, 0, X, C, 0, d.Q, KC, 0, 1, 1, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, false, true, rows, C, C, d.df, C, 0, X + (size_t)C * C, C, 0, d.Q + 2 * C, KC, 0, 1, 1, s->rs_inv, 2, 0);
    if (st == GF_OK) st = gemm_rs(ctx, false, true, rows, C, C, d.df, C, 0, X + 2 * (size_t)C * C, C, 0, d.Q + 2 * C, KC, 0, 1, 1, s->rs_inv, 2, 1);
    return st;
}
static bool padded_channels(const gf_smp *s) { return s->cfg.nChanels != s->ucfg.nChanels || s->cfg.uniform != s->ucfg.uniform; }
// floats of the device's parameter / gradient vector (the extra products' blocks behind the padded layout: gf_smp::n_extra)
static size_t padded_param_count(const gf_smp *s) {
    return param_count(s->cfg) + (size_t)s->cfg.nLevels * s->n_extra * s->cfg.nChanels * s->cfg.nChanels;
}
// the handle's padded copies of the caller's parameters / of the gradients of the running step
static gf_status pad_buffers(gf_smp *s) {
    if (s->pad_p && s->pad_g) return GF_OK;
    const size_t n = padded_param_count(s);
    if (!s->pad_p) GF_HIP_TRY(s->ctx, hipMalloc(reinterpret_cast<void **>(&s->pad_p), n * sizeof(float)));
    if (!s->pad_g && hipMalloc(reinterpret_cast<void **>(&s->pad_g), n * sizeof(float)) != hipSuccess) {
        // (both or neither: a later call must not find pad_p set and skip the gradient buffer)
        (void)hipGetLastError();
        s->pad_g = nullptr;
        (void)hipFree(s->pad_p);
        s->pad_p = nullptr;
        return fail(s->ctx, GF_ERR_NOMEM, "padded gradient buffer: %zu bytes", n * sizeof(float));
    }
    return GF_OK;
}
static gf_status pad_params_now(gf_smp *s, const float *params) {
    gf_status st = pad_buffers(s);
    if (st != GF_OK) return st;
    const long long n = (long long)padded_param_count(s);
    if (s->dup_channels) {
        const long long nu = (long long)param_count(s->ucfg);
        GF_HIP_TRY(s->ctx, hipMemsetAsync(s->pad_p, 0, (size_t)n * sizeof(float), s->ctx->stream));
        GF_LAUNCH(s->ctx, "smp_pad_params", v6_pad_parameters, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, params, s->pad_p, nu, pad_map(s->ucfg, s->cfg));
        return GF_OK;
    }
    GF_LAUNCH(s->ctx, "smp_pad_params", pad_parameters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, params, s->pad_p, n, pad_map(s->ucfg, s->cfg));
    return GF_OK;
}
static gf_status crop_grads_now(gf_smp *s, float *grads, int accumulate) {
    const long long n = (long long)param_count(s->cfg);
    if (s->dup_channels) {
        const long long nu = (long long)param_count(s->ucfg);
        GF_LAUNCH(s->ctx, "smp_crop_grads", v6_crop_gradients, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s->pad_g, grads, nu, pad_map(s->ucfg, s->cfg),
                  accumulate ? 1 : 0);
        return GF_OK;
    }
    GF_LAUNCH(s->ctx, "smp_crop_grads", crop_gradients, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->pad_g, grads, n, pad_map(s->ucfg, s->cfg),
              accumulate ? 1 : 0);
    return GF_OK;
}
// the handle's padded feature rows ([nMol][feature width of cfg]): the device's graph features, or the caller's feature gradient padded
static gf_status pad_feature_buffer(gf_smp *s) {
    const size_t n = (size_t)s->lay.nMol * (s->cfg.physics ? feature_width(s->cfg) : (size_t)s->cfg.nChanels);
    if (s->pad_feat_n >= n) return GF_OK;
    if (s->pad_feat) (void)hipFree(s->pad_feat);
    s->pad_feat = nullptr;
    s->pad_feat_n = 0;
    GF_HIP_TRY(s->ctx, hipMalloc(reinterpret_cast<void **>(&s->pad_feat), n * sizeof(float)));
    s->pad_feat_n = n;
    return GF_OK;
}
// feature rows between the two layouts, level block by level block (one block outside the towers): to_user crops, else pads (the
// caller's columns into a zeroed padded row)
static gf_status copy_feature_blocks(gf_smp *s, float *user, float *padded, bool to_user) {
    gf_ctx *ctx = s->ctx;
    const int nMol = s->lay.nMol, Cc = s->cfg.nChanels;
    const int nblk = s->cfg.physics ? s->cfg.nLevels + 1 : 1;
    const size_t wu = s->cfg.physics ? feature_width(s->ucfg) : (size_t)s->ucfg.nChanels, wp = (size_t)nblk * Cc;
    if (!to_user) GF_HIP_TRY(ctx, hipMemsetAsync(padded, 0, (size_t)nMol * wp * sizeof(float), ctx->stream));
    size_t uo = 0;
    for (int l = 0; l < nblk; ++l) {
        const size_t cu = s->cfg.physics ? (size_t)s->ucfg.level_channels(l) : (size_t)s->ucfg.nChanels;
        float *pu = user + uo, *pp = padded + (size_t)l * Cc;
        if (to_user)
            GF_HIP_TRY(ctx, hipMemcpy2DAsync(pu, wu * sizeof(float), pp, wp * sizeof(float), cu * sizeof(float), (size_t)nMol, hipMemcpyDeviceToDevice,
                                             ctx->stream));
        else
            GF_HIP_TRY(ctx, hipMemcpy2DAsync(pp, wp * sizeof(float), pu, wu * sizeof(float), cu * sizeof(float), (size_t)nMol, hipMemcpyDeviceToDevice,
                                             ctx->stream));
        uo += cu;
    }
    return GF_OK;
}
}  // namespace gf

static gf_status smp_forward_impl(gf_smp *s, const float *params, const float *targets, float *predict, float *loss, float *graph_feature);

gf_status gf_smp_forward(gf_smp *s, const float *params, const float *targets, float *predict, float *loss,
                         float *graph_feature) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!gf::padded_channels(s)) return smp_forward_impl(s, params, targets, predict, loss, graph_feature);
    gf_ctx *ctx = s->ctx;
    if (!s->prepared) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward before gf_smp_prepare");
    if (!params) {
        if (!s->own_p) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward: null params and no handle-owned model");
        params = s->own_p;
    }
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    gf_status st = gf::pad_params_now(s, params);
    if (st != GF_OK) return st;
    float *feat = nullptr;
    if (graph_feature) {
        st = gf::pad_feature_buffer(s);
        if (st != GF_OK) return st;
        feat = s->pad_feat;
    }
    st = smp_forward_impl(s, s->pad_p, targets, predict, loss, feat);
    if (st != GF_OK) return st;
    if (graph_feature) return gf::copy_feature_blocks(s, graph_feature, feat, /*to_user=*/true);   // (the padded columns are cropped)
    return GF_OK;
}

static gf_status smp_forward_impl(gf_smp *s, const float *params, const float *targets, float *predict, float *loss,
                                  float *graph_feature) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!s->prepared) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward before gf_smp_prepare");
    if (!params) {  // the handle's own model (gf_smp_parameters_upload)
        if (!s->own_p) return fail(ctx, GF_ERR_INVALID, "gf_smp_forward: null params and no handle-owned model");
        params = s->own_p;
    }
    if (s->cfg.physics && (targets || predict || loss))
        return fail(ctx, GF_ERR_INVALID, "gf_smp_forward: a physics tower only produces graph_feature (the head owns targets, predict and loss)");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    gf_status st = gf::ensure_ws(ctx, s->ws_need);
    if (st != GF_OK) return st;
    const gfsmp::BatchLayout &B = s->lay;
    const int L = s->cfg.nLevels, C = s->cfg.nChanels, FD = s->cfg.fdim();
    const float *H, *W;
    std::vector<const float *> K, b;
    gf::view_params<const float>(s->cfg, params, &H, &K, &b, &W);
    s->extra_w = s->n_extra ? params + gf::param_count(s->cfg) : nullptr;   // (SMP_2D_ver7 on the 18-slice level: [.. W | X_1 .. X_L])
    // level 0: f_0 = LeakyReLU(X H^T)   (MatMul(H, x_v) per vertex, SMP_omega.h:618)
    const int nV = B.level[0].nNodes;
    st = gf::gemm(ctx, false, true, nV, C, FD, s->x, FD, 0, H, FD, 0, s->lv[0].f, C, 0, 1, 0);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "smp_bias_lrelu", gf::bias_lrelu_forward, dim3(gf::grid_for((size_t)nV * C)), dim3(256), 0, s->lv[0].f,
              (const float *)nullptr, C, (size_t)nV * C);
    for (int l = 0; l <= L; ++l) s->lv[l].psum_ready = s->lv[l].pmax_ready = false;
    s->bwd_consumed = false;
    st = gf::dup_level(s, 0);   // (SMP_2D_ver6 on the 18-slice level: channels [C, 2C) <- the transposed matrices; level 0: copies)
    if (st != GF_OK) return st;
    if (s->fused) {
        if (s->wbound && C == 64) GF_HIP_TRY(ctx, hipMemsetAsync(s->wbound, 0, sizeof(unsigned) * gf::smp_wgrad_bound_words() * (size_t)(L + 1), ctx->stream));
        st = gf::smp_fused_stack_all(s, K);
        if (st != GF_OK) return st;
    }
    for (int l = 1; l <= L; ++l) {
        const gfsmp::LevelLayout &h = B.level[l];
        const gf_smp::DevLevel &d = s->lv[l];
        if (s->fused && gf::smp_fused_supported(s, l)) {
            st = gf::smp_fused_forward_level(s, l, K[l], b[l]);
            if (st != GF_OK) return st;
            if (l < L) st = gf::dup_level(s, l);
            if (st != GF_OK) return st;
            continue;
        }
        s->lv[l].t_zeros = s->lv[l].t_filled = false;  // (the op-by-op level uses all of Q: the zeros kept in the fused level's T region are gone)
        st = gf::ensure_P(s);
        if (st != GF_OK) return st;
        const int Cp = s->cfg.level_channels(l - 1), Cc = s->cfg.level_channels(l);  // (equal unless a physics tower)
        GF_LAUNCH(ctx, "smp_promote_fwd", gf::promote_forward, dim3((unsigned)h.pairs), dim3(256), 0, s->lv[l - 1].f, s->P,
                  d.node_s, d.node_row, d.node_p, d.node_pair, d.pair_node, d.pair_src_row, d.pair_src_s, d.pi, Cp);
        st = gf::smp_contract(s, l, /*backward=*/false);
        if (st != GF_OK) return st;
        if (s->drop_on) {
            gf::LaunchTimer lt__(ctx, "smp_slice_dropout");
            gf::launch_node_slice_scale(ctx, d.Q, d.node_s, d.node_row, d.keep_mask, s->drop_scale, Cp, h.nNodes);
            lt__.done();
            GF_LAUNCH_CHECK(ctx, "smp_slice_dropout");
        }
        // K-projection over all buckets at once: [rows, KC] x [KC, C]  (CustomMatMulTensor layout: x K_l^T, K_l = [C, KC])
        const int KC = s->cfg.nContractions * Cp;
        st = s->cfg.custom_matmul ? gf::gemm(ctx, false, true, (int)h.rows, Cc, KC, d.Q, KC, 0, K[l], KC, 0, d.f, Cc, 0, 1, 0)
                                  : gf::gemm(ctx, false, false, (int)h.rows, Cc, KC, d.Q, KC, 0, K[l], Cc, 0, d.f, Cc, 0, 1, 0);
        if (st != GF_OK) return st;
        st = gf::extra_products_forward(s, l);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smp_bias_lrelu", gf::bias_lrelu_forward, dim3(gf::grid_for((size_t)h.rows * Cc)), dim3(256), 0, d.f,
                  b[l], Cc, (size_t)h.rows * Cc);
        if (l < L) st = gf::dup_level(s, l);
        if (st != GF_OK) return st;
    }
    if (s->cfg.physics) {  // every level read out into its block of the feature row; the head (MLP, loss) is the caller's
        const int width = (int)gf::feature_width(s->cfg);
        int off = 0;
        for (int l = 0; l <= L; ++l) {
            const gf_smp::DevLevel &d = s->lv[l];
            const int Cc = s->cfg.level_channels(l);
            if (l >= 1 && d.psum && d.psum_ready && Cc == 64)   // (a fused level left its row panels' column sums behind)
                GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<64>, dim3((unsigned)((B.level[l].nNodes + 3) / 4)), dim3(256), 0, d.psum,
                          d.node_panel, B.level[l].nNodes, d.fwd_npanels, d.sh, d.vf);
            else if (l >= 1 && d.psum && d.psum_ready && Cc == 32)
                GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<32>, dim3((unsigned)((B.level[l].nNodes + 7) / 8)), dim3(256), 0, d.psum,
                          d.node_panel, B.level[l].nNodes, d.fwd_npanels, d.sh, d.vf);
            else if (l >= 1 && d.psum && d.psum_ready && Cc == 16)
                GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<16>, dim3((unsigned)((B.level[l].nNodes + 15) / 16)), dim3(256), 0, d.psum,
                          d.node_panel, B.level[l].nNodes, d.fwd_npanels, d.sh, d.vf);
            else if (Cc % 4 == 0 && Cc <= 1024)   // (workgroup per node, float4 lanes: a padded tower's levels)
                GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_v, dim3(B.level[l].nNodes), dim3(256), 0, d.f, d.node_s, d.node_row, d.sh, d.vf, Cc);
            else
                GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes, dim3(gf::grid_for((size_t)B.level[l].nNodes * Cc)), dim3(256), 0, d.f,
                          d.node_s, d.node_row, d.sh, d.vf, Cc, (size_t)B.level[l].nNodes * Cc);
            if (l >= 1 && d.psum && d.psum_ready && gf::smp_panel_channels(Cc)) {   // (the nodes above 32 positions have no panels: from their rows)
                int n0 = B.level[l].nNodes;
                for (const gfsmp::Bucket &bk : B.level[l].buckets)
                    if (bk.s > 32) {
                        n0 = bk.first_node;
                        break;
                    }
                if (n0 < B.level[l].nNodes)
                    GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_v, dim3(B.level[l].nNodes - n0), dim3(256), 0, d.f, d.node_s + n0, d.node_row + n0,
                              d.sh + (size_t)n0 * Cc, d.vf + (size_t)n0 * Cc, Cc);
            }
            GF_LAUNCH(ctx, "smp_level_feature", gf::level_feature_sum, dim3(B.nMol), dim3(64), 0, d.vf, s->mol_ptr, d.node_of_vertex, s->g,
                      Cc, width, off);
            off += Cc;
        }
        if (graph_feature)
            GF_HIP_TRY(ctx, hipMemcpyAsync(graph_feature, s->g, sizeof(float) * (size_t)B.nMol * width, hipMemcpyDeviceToDevice, ctx->stream));
        s->forwarded = true;
        s->has_targets = false;
        gf::mark_used(s);
        return GF_OK;
    }
    const gfsmp::LevelLayout &top = B.level[L];
    if (C == 64 && s->lv[L].psum_ready) {
        GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<64>, dim3((unsigned)((top.nNodes + 3) / 4)), dim3(256), 0, s->lv[L].psum,
                  s->lv[L].node_panel, top.nNodes, s->lv[L].fwd_npanels, s->sh, s->vf);
    } else if (C == 32 && s->lv[L].psum_ready) {
        GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<32>, dim3((unsigned)((top.nNodes + 7) / 8)), dim3(256), 0, s->lv[L].psum,
                  s->lv[L].node_panel, top.nNodes, s->lv[L].fwd_npanels, s->sh, s->vf);
    } else if (C == 16 && s->lv[L].psum_ready) {
        GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_panels<16>, dim3((unsigned)((top.nNodes + 15) / 16)), dim3(256), 0, s->lv[L].psum,
                  s->lv[L].node_panel, top.nNodes, s->lv[L].fwd_npanels, s->sh, s->vf);
    } else if (C % 4 == 0 && C <= 1024) {
        GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_v, dim3(top.nNodes), dim3(256), 0, s->lv[L].f, s->lv[L].node_s,
                  s->lv[L].node_row, s->sh, s->vf, C);
    } else {
        GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes, dim3(gf::grid_for((size_t)top.nNodes * C)), dim3(256), 0,
                  s->lv[L].f, s->lv[L].node_s, s->lv[L].node_row, s->sh, s->vf, C, (size_t)top.nNodes * C);
    }
    if (s->lv[L].psum_ready && gf::smp_panel_channels(C)) {
        // (the nodes above 32 positions have no row panels -- an empty range in the panel readout: theirs from the rows of f_L; nodes are
        //  numbered by size)
        int n0 = top.nNodes;
        for (const gfsmp::Bucket &bk : top.buckets)
            if (bk.s > 32) {
                n0 = bk.first_node;
                break;
            }
        if (n0 < top.nNodes)
            GF_LAUNCH(ctx, "smp_readout_nodes", gf::readout_nodes_v, dim3(top.nNodes - n0), dim3(256), 0, s->lv[L].f, s->lv[L].node_s + n0,
                      s->lv[L].node_row + n0, s->sh + (size_t)n0 * C, s->vf + (size_t)n0 * C, C);
    }
    GF_LAUNCH(ctx, "smp_readout_mol", gf::readout_molecules, dim3(B.nMol), dim3(256), 0, s->vf, s->mol_ptr, s->mol_nodes, W,
              targets, s->g, s->yhat, loss, s->dy, C);
    if (predict) GF_HIP_TRY(ctx, hipMemcpyAsync(predict, s->yhat, sizeof(float) * B.nMol, hipMemcpyDeviceToDevice, ctx->stream));
    if (graph_feature)
        GF_HIP_TRY(ctx, hipMemcpyAsync(graph_feature, s->g, sizeof(float) * (size_t)B.nMol * C, hipMemcpyDeviceToDevice, ctx->stream));
    s->forwarded = true;
    s->has_targets = targets != nullptr;
    gf::mark_used(s);
    return GF_OK;
}

// RisiContraction_18_dropout for the next forward / backward of a physics tower (SMP_sigma_pairgraphs): masks[(l-1) * nVertices + gv]
// = kept-slice bits of the contraction of global vertex gv (molecules back to back) at level l, drawn by the caller in the
// reference's order; scale = 1 (train) or nKept / 18 with all bits set (test).  masks == NULL: plain RisiContraction_18.
gf_status gf_smp_dropout_masks(gf_smp *s, const unsigned *masks, float scale) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!masks) {
        s->drop_on = false;
        return GF_OK;
    }
    if (!s->prepared || !s->cfg.physics) return fail(ctx, GF_ERR_INVALID, "gf_smp_dropout_masks: needs a prepared physics tower");
    const gfsmp::BatchLayout &B = s->lay;
    const int totalV = B.mol_first_vertex[B.nMol];
    // The masks go up through a page-locked staging table of the handle, all levels at once, behind an event: a pageable source makes
    // every copy a blocking one and the old per-level hipStreamSynchronize drained the context's stream three times per tower and
    // step (the host then drew the next masks -- a million rand() calls per 1024-sample step -- with the device idle).
    const size_t need = (size_t)s->cfg.nLevels * totalV;
    if (s->mask_stage_n < need) {
        if (s->mask_stage) (void)hipHostFree(s->mask_stage);
        s->mask_stage = nullptr;
        s->mask_stage_n = 0;
        GF_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&s->mask_stage), need * sizeof(unsigned), hipHostMallocDefault));
        s->mask_stage_n = need;
    }
    if (!s->ev_mask) GF_HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_mask, hipEventDisableTiming));
    else GF_HIP_TRY(ctx, hipEventSynchronize(s->ev_mask));   // (the previous step's copies have left the table: long done)
    for (int l = 1; l <= s->cfg.nLevels; ++l) {
        unsigned *by_node = s->mask_stage + (size_t)(l - 1) * totalV;
        for (int gv = 0; gv < totalV; ++gv) by_node[(size_t)B.node_of_vertex[l][gv]] = masks[(size_t)(l - 1) * totalV + gv];
        GF_HIP_TRY(ctx, hipMemcpyAsync(s->lv[l].keep_mask, by_node, sizeof(unsigned) * totalV, hipMemcpyHostToDevice, ctx->stream));
    }
    GF_HIP_TRY(ctx, hipEventRecord(s->ev_mask, ctx->stream));
    // the factor tables of the fused levels (smp_fused.hip: build_dropout_factors fills them at every forward): towers computed at 32 channels
    if (s->cfg.square() && (s->cfg.nChanels == 32 || s->cfg.nChanels == 16))
        for (int l = 1; l <= s->cfg.nLevels; ++l) {
            gf_smp::DevLevel &d = s->lv[l];
            if (d.nodefac && d.rowfac8) continue;
            gf_status st = gf::upload(s, &d.nodefac, nullptr, (size_t)B.level[l].nNodes * 18);
            if (st == GF_OK) st = gf::upload(s, &d.rowfac8, nullptr, (size_t)B.level[l].rows * 8);
            if (st != GF_OK) return st;
        }
    s->drop_on = true;
    s->drop_scale = scale;
    return GF_OK;
}

gf_status gf_smp_set_grad_allreduce(gf_smp *s, int on) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    s->grad_allreduce = on ? 1 : 0;
    return GF_OK;
}

// The reverse sweep.  dfeat == nullptr: from the loss of the last forward (SMP_omega / SMP_beta / SMP_2D).  dfeat != nullptr
// (physics towers): from the gradient of the tower's feature rows, [nMol][feature_width], given by the caller's head.
// Slice dropout in TEST mode: the fused level cannot run the reference's unscaled test-mode sweep (smp_fused_backward_level).  Refused
// BEFORE a gradient is written or a collective handed to RCCL -- not in the middle of the level loop, where the readout's gradients
// were already there and the peers of a data-parallel run were left waiting for segments that never came (round-5 advice).  The
// composite model asks before its head's backward as well (gf_smp_model_backward).
extern "C++" gf_status gf::smp_backward_admissible(const gf_smp *s) {
    if (s->drop_on && s->drop_scale != 1.f && s->fused && s->prepared)
        for (int l = 1; l <= s->cfg.nLevels; ++l)
            if (gf::smp_fused_supported(s, l))
                return fail(s->ctx, GF_ERR_UNSUPPORTED, "gf_smp_backward: fused level %d under slice dropout in test mode (scale %.4f): set GF_SMP_FUSED_DROPOUT=0 "
                                                        "for the reference's unscaled test-mode sweep", l, (double)s->drop_scale);
    return GF_OK;
}
static gf_status smp_backward_impl(gf_smp *s, const float *params, float *grads, int accumulate, const float *dfeat) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    gf_ctx *ctx = s->ctx;
    if (!s->forwarded) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward before gf_smp_forward");
    if (!dfeat && !s->has_targets)  // Predict / Feature forward: dy would be y - 0, a gradient against a target nobody gave
        return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: the last gf_smp_forward had no targets");
    if ((dfeat != nullptr) != (s->cfg.physics != 0))
        return fail(ctx, GF_ERR_INVALID, s->cfg.physics ? "a physics tower is differentiated with gf_smp_backward_features"
                                                         : "gf_smp_backward_features needs a physics tower");
    if (!params && !grads && s->own_p) {
        params = s->own_p;
        grads = s->own_g;
    }
    if (!params || !grads) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: null argument");
    {
        gf_status st0 = gf::smp_backward_admissible(s);
        if (st0 != GF_OK) return st0;
    }
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    {   // (a no-op after this batch's forward; it makes the reverse sweep independent of who grew the context's workspace last)
        gf_status st = gf::ensure_ws(ctx, s->ws_need);
        if (st != GF_OK) return st;
    }
    const gfsmp::BatchLayout &B = s->lay;
    const int L = s->cfg.nLevels, C = s->cfg.nChanels, FD = s->cfg.fdim();
    const float *H, *W;
    std::vector<const float *> K, b;
    gf::view_params<const float>(s->cfg, params, &H, &K, &b, &W);
    float *dH, *dW;
    std::vector<float *> dK, db;
    gf::view_params<float>(s->cfg, grads, &dH, &dK, &db, &dW);
    const size_t np = gf::param_count(s->cfg);
    s->extra_w = s->n_extra ? params + np : nullptr;
    s->extra_g = s->n_extra ? grads + np : nullptr;   // (every level writes its own blocks: nothing to clear)
    const bool dp = gf::dist_active(ctx) && s->grad_allreduce && !s->cfg.physics;  // (towers: the composite model reduces its own flat buffer)
    if (dp && accumulate)
        return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: accumulate with a communicator would re-sum earlier global sums "
                                         "(gf_smp_set_grad_allreduce(smp, 0) and reduce once at the end instead)");
    if (s->bwd_consumed)   // (an op-by-op level's reverse sweep overwrites its Q with dQ: the forward state is gone)
        return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: a second reverse sweep needs a new gf_smp_forward (op-by-op levels keep dQ in place of Q)");
    s->dp_grads = nullptr;
    if (dp) {
        if (!s->ev_grad) GF_HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_grad, hipEventDisableTiming));
        if (!s->ev_comm) GF_HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming));
        // Watchdog: the join of the PREVIOUS sweep's all-reduces is waited for here, by polling under GF_DIST_TIMEOUT_S -- a rank whose
        // peers never joined an exchange fails with its rank, the world and the segment in gf_last_error instead of queueing work
        // behind a collective that will never finish.  (The host may still run a whole forward pass ahead of the device.)
        if (s->dp_join_pending) {
            s->dp_join_pending = false;
            gf_status stw = gf::dist_wait_event(ctx, s->ev_comm, "the join of the previous gf_smp_backward's gradient all-reduces");
            if (stw != GF_OK) return stw;
        }
        s->dp_grads = grads;
    }
    struct DpScope {  // whatever the exit path, the next call starts clean
        gf_smp *s;
        ~DpScope() { s->dp_grads = nullptr; }
    } dp_scope = {s};
    if (!accumulate) GF_LAUNCH(ctx, "smp_zero", gf::zero_f32, dim3(gf::grid_for(np)), dim3(256), 0, grads, np);
    gf_status st;
    const gfsmp::LevelLayout &top = B.level[L];
    const int fwidth = dfeat ? (int)gf::feature_width(s->cfg) : 0;
    std::vector<int> foff(L + 2, 0);
    for (int l = 0; l <= L; ++l) foff[l + 1] = foff[l] + s->cfg.level_channels(l);
    // physics: the read-out of level l adds  dfeat[mol][block l] * lrelu'(sh_l)  at every position of every node of the level
    auto feature_backward = [&](int l, int acc) -> gf_status {
        const gf_smp::DevLevel &dl = s->lv[l];
        GF_LAUNCH(ctx, "smp_level_feature_bwd", gf::level_feature_backward, dim3(B.level[l].nNodes), dim3(256), 0, dfeat, dl.sh, dl.node_mol,
                  dl.node_s, dl.node_row, dl.df, s->cfg.level_channels(l), fwidth, foff[l], acc);
        return GF_OK;
    };
    if (!dfeat) GF_LAUNCH(ctx, "smp_readout_dW", gf::readout_dW, dim3(1), dim3(1024), 0, s->dy, s->g, dW, C, B.nMol);
    const bool top_fused = !dfeat && s->fused && gf::smp_fused_supported(s, L);
    // a tower's FUSED level takes its read-out gradient as one vector per node (combine-backward adds it to the node's rows): the pass
    // that broadcast it into df_l -- a read-modify-write of every row -- only runs for op-by-op levels and level 0
    auto level_fused = [&](int l) { return l >= 1 && s->fused && gf::smp_fused_supported(s, l); };
    auto feature_nodevec = [&](int l) -> gf_status {
        const gf_smp::DevLevel &dl = s->lv[l];
        const size_t n = (size_t)B.level[l].nNodes * s->cfg.level_channels(l);
        GF_LAUNCH(ctx, "smp_level_feature_bwd", gf::level_feature_nodevec, dim3(gf::grid_for(n)), dim3(256), 0, dfeat, dl.sh, dl.node_mol, dl.dshl,
                  s->cfg.level_channels(l), fwidth, foff[l], n);
        return GF_OK;
    };
    if (dfeat) {
        if (!level_fused(L)) {
            st = feature_backward(L, 0);
            if (st != GF_OK) return st;
        }
    } else if (top_fused) {
        GF_LAUNCH(ctx, "smp_readout_bwd", gf::readout_backward_nodevec, dim3(gf::grid_for((size_t)top.nNodes * C)), dim3(256), 0, s->dy,
                  W, s->sh, s->top_node_mol, s->dsh, C, (size_t)top.nNodes * C);
    } else {
        GF_LAUNCH(ctx, "smp_readout_bwd", gf::readout_backward_nodes, dim3(top.nNodes), dim3(256), 0, s->dy, W, s->sh,
                  s->top_node_mol, s->lv[L].node_s, s->lv[L].node_row, s->lv[L].df, C);
    }
    for (int l = L; l >= 1; --l) {
        const gfsmp::LevelLayout &h = B.level[l];
        const gf_smp::DevLevel &d = s->lv[l];
        if (l < L) {   // (SMP_2D_ver6 on the 18-slice level: the gradient of the transposed copies joins the matrices')
            st = gf::fold_level(s, l);
            if (st != GF_OK) return st;
        }
        if (s->fused && gf::smp_fused_supported(s, l)) {
            if (dfeat) {
                st = feature_nodevec(l);
                if (st != GF_OK) return st;
            }
            st = gf::smp_fused_backward_level(s, l, K[l], dK[l], db[l], dfeat ? s->lv[l].dshl : (l == L && top_fused) ? s->dsh : nullptr,
                                              /*rows_too=*/dfeat && l < L);
            if (st != GF_OK) return st;
        } else {
            s->bwd_consumed = true;
        // dZ = dF * lrelu'(z) in place; db_l += column sums
            const int Cc = s->cfg.level_channels(l), Cq = s->cfg.level_channels(l - 1);  // (equal unless a physics tower)
            const int rpb = 1024;
            const int nb = (int)((h.rows + rpb - 1) / rpb);
            GF_LAUNCH(ctx, "smp_lrelu_bwd", gf::lrelu_backward_colsum, dim3(nb), dim3(256), 0, d.f, d.df, s->colpart, Cc,
                      (long long)h.rows, rpb);
            GF_LAUNCH(ctx, "smp_colsum", gf::colsum_finish, dim3(1), dim3(256), 0, s->colpart, db[l], Cc, nb);
            // dK_l += Q^T dZ   (MatMul::backward second operand), then dQ = dZ K_l^T overwrites Q (first operand)
            const int KC = s->cfg.nContractions * Cq;
            st = gf::extra_products_wgrad(s, l);
            if (st != GF_OK) return st;
            if (s->cfg.custom_matmul) {  // CustomMatMulTensor::backward (CustomMatMulTensor.h:70-85): dK_l [C, KC] += dZ^T Q, dQ = dZ K_l
                st = gf::gemm(ctx, true, false, Cc, KC, (int)h.rows, d.df, Cc, 0, d.Q, KC, 0, dK[l], KC, 0, 1, 1);
                if (st != GF_OK) return st;
                st = gf::gemm(ctx, false, false, (int)h.rows, KC, Cc, d.df, Cc, 0, K[l], KC, 0, d.Q, KC, 0, 1, 0);
            } else {
                st = gf::gemm(ctx, true, false, KC, Cc, (int)h.rows, d.Q, KC, 0, d.df, Cc, 0, dK[l], Cc, 0, 1, 1);
                if (st != GF_OK) return st;
                st = gf::gemm(ctx, false, true, (int)h.rows, KC, Cc, d.df, Cc, 0, K[l], Cc, 0, d.Q, KC, 0, 1, 0);
            }
            if (st != GF_OK) return st;
            st = gf::extra_products_backward(s, l);
            if (st != GF_OK) return st;
            st = gf::smp_dp_level_done(s, l);
            if (st != GF_OK) return st;
            if (s->drop_on) {  // the dropped slices receive no gradient
                gf::LaunchTimer lt__(ctx, "smp_slice_dropout");
                gf::launch_node_slice_scale(ctx, d.Q, d.node_s, d.node_row, d.keep_mask, 1.f, Cq, h.nNodes);
                lt__.done();
                GF_LAUNCH_CHECK(ctx, "smp_slice_dropout");
            }
            st = gf::smp_contract(s, l, /*backward=*/true);
            if (st != GF_OK) return st;
        }
        const gf_smp::DevLevel &pv = s->lv[l - 1];
        const bool diag_level = s->fused && gf::smp_fused_supported(s, l);  // its D_bb / D_ac gradients arrive through dFdc
        if (diag_level && gf::smp_fused_gather_enabled(s, l)) {
            st = gf::smp_fused_gather_backward(s, l);
            if (st != GF_OK) return st;
            if (dfeat && !level_fused(l - 1)) {  // (a tower: level l-1 is read out too)
                st = feature_backward(l - 1, 1);
                if (st != GF_OK) return st;
            }
            continue;
        }
        GF_LAUNCH(ctx, "smp_promote_bwd", gf::promote_backward, dim3(B.level[l - 1].nNodes), dim3(256), 0, s->P, pv.df,
                  pv.node_s, pv.node_row, d.cons_ptr, d.cons_slab, d.cons_s, d.cons_inv_off, d.inv, s->cfg.level_channels(l - 1),
                  diag_level ? d.dFdc : (const float *)nullptr, pv.node_pair, pv.node_center);
        if (dfeat && !level_fused(l - 1)) {  // level l-1 is read out too: its own contribution joins what its consumers sent down
            st = feature_backward(l - 1, 1);
            if (st != GF_OK) return st;
        }
    }
    // level 0: dZ0 = dF0 * lrelu'; dH += dZ0^T X
    {
        const int nV = B.level[0].nNodes;
        // (level 0 has no bias: the column sums are discarded, so small row blocks cost nothing downstream; colpart holds
        //  maxrows / 1024 + maxpairs / 256 + 2 rows and level 0 has at most maxpairs / 64 blocks... keep nb within it)
        st = gf::fold_level(s, 0);
        if (st != GF_OK) return st;
        int rpb = 64;
        while ((nV + rpb - 1) / rpb > (int)s->colpart_rows && rpb < 1024) rpb *= 2;
        const int nb = (nV + rpb - 1) / rpb;
        GF_LAUNCH(ctx, "smp_lrelu_bwd", gf::lrelu_backward_colsum, dim3(nb), dim3(256), 0, s->lv[0].f, s->lv[0].df, s->colpart,
                  C, (long long)nV, rpb);
        st = gf::gemm(ctx, true, false, C, FD, nV, s->lv[0].df, C, 0, s->x, FD, 0, dH, FD, 0, 1, 1);
        if (st != GF_OK) return st;
    }
    if (dp) {  // dH, then join: everything after this call on the context's stream sees the global sums
        st = gf::smp_dp_level_done(s, 0);
        if (st != GF_OK) return st;
        GF_HIP_TRY(ctx, hipEventRecord(s->ev_comm, gf::dist_stream(ctx)));
        GF_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->ev_comm, 0));
        s->dp_join_pending = true;
    }
    gf::mark_used(s);
    return GF_OK;
}

gf_status gf_smp_backward(gf_smp *s, const float *params, float *grads, int accumulate) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!gf::padded_channels(s)) return smp_backward_impl(s, params, grads, accumulate, nullptr);
    gf_ctx *ctx = s->ctx;
    if (!params && !grads && s->own_p) {
        params = s->own_p;
        grads = s->own_g;
    }
    if (!params || !grads) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: null argument");
    if (accumulate && gf::dist_active(ctx) && s->grad_allreduce)
        return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: accumulate with a communicator would re-sum earlier global sums "
                                         "(gf_smp_set_grad_allreduce(smp, 0) and reduce once at the end instead)");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    gf_status st = gf::pad_params_now(s, params);   // (the caller may have stepped the parameters since the forward pass: same values then)
    if (st != GF_OK) return st;
    st = smp_backward_impl(s, s->pad_p, s->pad_g, 0, nullptr);   // (with a communicator: the padded segments are all-reduced)
    if (st != GF_OK) return st;
    return gf::crop_grads_now(s, grads, accumulate);
}

gf_status gf_smp_backward_features(gf_smp *s, const float *params, float *grads, const float *d_feature, int accumulate) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!d_feature) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_backward_features: null feature gradient");
    if (!gf::padded_channels(s)) return smp_backward_impl(s, params, grads, accumulate, d_feature);
    gf_ctx *ctx = s->ctx;
    if (!s->cfg.physics) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward_features needs a physics tower");
    if (!s->forwarded) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward before gf_smp_forward");
    if (!params && !grads && s->own_p) {   // (the handle-owned model, as the unpadded path and gf_smp_backward take it)
        params = s->own_p;
        grads = s->own_g;
    }
    if (!params || !grads) return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: null argument");
    GF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    gf_status st = gf::pad_params_now(s, params);
    if (st == GF_OK) st = gf::pad_feature_buffer(s);
    if (st == GF_OK) st = gf::copy_feature_blocks(s, const_cast<float *>(d_feature), s->pad_feat, /*to_user=*/false);
    if (st != GF_OK) return st;
    st = smp_backward_impl(s, s->pad_p, s->pad_g, 0, s->pad_feat);
    if (st != GF_OK) return st;
    return gf::crop_grads_now(s, grads, accumulate);
}

size_t gf_smp_feature_width(const gf_smp *s) {
    if (!s) return 0;
    return s->cfg.physics ? gf::feature_width(s->ucfg) : (size_t)s->ucfg.nChanels;
}

/* Host-only graph preparation of ONE molecule (no device needed): receptive fields phi[l][v] as
 * [L+1][V][cap+1] ints (slot 0 = size) and, optionally, the WL features [V][F(D+1)].  Lets the host logic be tested
 * on a CPU-only box and inspected by callers. */
gf_status gf_smp_prepare_molecule_host(const gf_smp_config *cfg, int V, const int *adj, const double *feature,
                                       int *phi_out, double *wl_out) {
    if (!cfg || V <= 0 || !adj || !feature || !phi_out) return fail(nullptr, GF_ERR_INVALID, "gf_smp_prepare_molecule_host: bad argument");
    gfsmp::Config c;
    c.nLevels = cfg->nLevels;
    c.nChanels = cfg->nChanels;
    c.nFeatures = cfg->nFeatures;
    c.nDepth = cfg->nDepth;
    c.max_receptive_field = cfg->max_receptive_field;
    c.has_WL_ordering = cfg->has_WL_ordering;
    c.physics = cfg->physics ? 1 : 0;
    gfsmp::Molecule m;
    gfsmp::prepare_molecule(c, V, adj, feature, &m);
    const int cap = c.max_receptive_field;
    for (int l = 0; l <= c.nLevels; ++l)
        for (int v = 0; v < V; ++v) {
            int *p = phi_out + ((size_t)l * V + v) * (cap + 1);
            const std::vector<int> &f = m.phi[l][v];
            p[0] = (int)f.size();
            for (int i = 0; i < cap; ++i) p[1 + i] = i < (int)f.size() ? f[i] : -1;
        }
    if (wl_out)
        for (size_t i = 0; i < m.wl.size(); ++i) wl_out[i] = m.wl[i];
    return GF_OK;
}

/* 1 (default): fused level kernels where the shape allows; 0: the op-by-op pipeline (promotion, RisiContraction_18,
 * MatMul, VectorAddTensor, LeakyReLU3D as separate kernels).  Both produce the same results within fp32 rounding. */
gf_status gf_smp_device_bytes(const gf_smp *s, size_t *in_use, size_t *reserved) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    size_t u = 0, r = 0;
    for (const gf_smp::Block &b : s->pool) {
        r += b.bytes;
        if (b.used) u += b.bytes;
    }
    if (in_use) *in_use = u;
    if (reserved) *reserved = r;
    return GF_OK;
}

gf_status gf_smp_set_fused(gf_smp *s, int on) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    s->fused = on ? 1 : 0;
    s->forwarded = false;
    return GF_OK;
}

/* introspection for parity tests: receptive field phi_level(v) of molecule mol; returns its size */
int gf_smp_receptive_field(const gf_smp *s, int mol, int level, int v, int *out, int capacity) {
    if (!s || !s->prepared || mol < 0 || mol >= s->lay.nMol || level < 0 || level > s->cfg.nLevels) return -1;
    const gfsmp::Molecule &M = s->lay.mols[mol];
    if (v < 0 || v >= M.V) return -1;
    const std::vector<int> &f = M.phi[level][v];
    for (int i = 0; i < (int)f.size() && i < capacity; ++i) out[i] = f[i];
    return (int)f.size();
}

/* introspection for parity tests: the level-`level` activation f_l[v] of molecule `mol` ([s][s][C], what level[l]->f[v]->value
 * holds in the reference after forward(), SMP_omega.h:667-669) or its reduced adjacency ([s][s], level[l]->adj[v], :556-581),
 * copied to a HOST buffer.  Returns the element count, or -1 (bad argument / capacity too small / not forwarded).  Blocking. */
static long long smp_read_node(gf_smp *s, int mol, int level, int v, float *out, size_t capacity, bool adjacency) {
    if (!s || !s->prepared || !out || mol < 0 || mol >= s->lay.nMol || level < 0 || level > s->cfg.nLevels) return -1;
    if (adjacency ? level < 1 : !s->forwarded) return -1;
    const gfsmp::LevelLayout &h = s->lay.level[level];
    int n = -1;
    if (level == 0) {
        const int g = s->lay.mol_first_vertex[mol] + v;
        if (v >= 0 && g < s->lay.mol_first_vertex[mol + 1]) n = g;
    } else {
        for (int i = 0; i < h.nNodes && n < 0; ++i)
            if (h.node_mol[i] == mol && h.node_vertex[i] == v) n = i;
    }
    if (n < 0) return -1;
    const size_t sz = (size_t)h.node_s[n], C = (size_t)s->cfg.level_channels(level);  // (physics towers halve per level)
    const size_t Cu = (size_t)s->ucfg.level_channels(level);                              // (the caller's channels: the padded ones are cropped)
    const size_t count = adjacency ? sz * sz : sz * sz * Cu;
    if (count > capacity) return -1;
    const float *src = adjacency ? s->lv[level].adj + h.node_row[n] : s->lv[level].f + (size_t)h.node_row[n] * C;
    if (!adjacency && Cu != C) {
        if (hipMemcpy2DAsync(out, Cu * sizeof(float), src, C * sizeof(float), Cu * sizeof(float), sz * sz, hipMemcpyDeviceToHost, s->ctx->stream) != hipSuccess)
            return -1;
    } else if (hipMemcpyAsync(out, src, count * sizeof(float), hipMemcpyDeviceToHost, s->ctx->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(s->ctx->stream) != hipSuccess) return -1;
    return (long long)count;
}
long long gf_smp_read_activation(gf_smp *s, int mol, int level, int v, float *out, size_t capacity) {
    return smp_read_node(s, mol, level, v, out, capacity, false);
}
long long gf_smp_read_reduced_adjacency(gf_smp *s, int mol, int level, int v, float *out, size_t capacity) {
    return smp_read_node(s, mol, level, v, out, capacity, true);
}

/* counts used by the bench to report algorithmic work: rows = sum s^2, ppos = sum s^3 at a level */
gf_status gf_smp_level_sizes(const gf_smp *s, int level, long long *nodes, long long *rows, long long *ppos) {
    if (!s) return fail(nullptr, GF_ERR_INVALID, "null smp handle");
    if (!s->prepared || level < 0 || level > s->cfg.nLevels) return fail(s->ctx, GF_ERR_INVALID, "gf_smp_level_sizes: bad level");
    const gfsmp::LevelLayout &h = s->lay.level[level];
    if (nodes) *nodes = h.nNodes;
    if (rows) *rows = h.rows;
    if (ppos) *ppos = h.ppos;
    return GF_OK;
}
long long gf_smp_level_pairs(const gf_smp *s, int level) {
    if (!s || !s->prepared || level < 0 || level > s->cfg.nLevels) return -1;
    return s->lay.level[level].pairs;
}
long long gf_smp_level_covered_rows(const gf_smp *s, int level) {
    if (!s || !s->prepared || level < 0 || level > s->cfg.nLevels) return -1;
    const gfsmp::LevelLayout &h = s->lay.level[level];
    if (level == 0 || !s->lv[level].rowflag || h.rows == 0) return h.rows;
    gf_smp *ms = const_cast<gf_smp *>(s);   // (cached per prepared batch; see the header for the threading contract)
    if (ms->h_covered.size() != (size_t)s->cfg.nLevels + 1) ms->h_covered.assign((size_t)s->cfg.nLevels + 1, -1);
    if (ms->h_covered[level] >= 0) return ms->h_covered[level];
    std::vector<unsigned char> fl((size_t)h.rows);
    hipStream_t up = s->upload ? s->upload : s->ctx->stream;
    if (hipMemcpyAsync(&fl[0], s->lv[level].rowflag, fl.size(), hipMemcpyDeviceToHost, up) != hipSuccess || hipStreamSynchronize(up) != hipSuccess)
        return -1;
    long long n = 0;
    for (size_t i = 0; i < fl.size(); ++i) n += (fl[i] >> 1) & 1;
    ms->h_covered[level] = n;
    return n;
}
long long gf_smp_level_present_rows(const gf_smp *s, int level) {
    if (!s || !s->prepared || level < 0 || level > s->cfg.nLevels) return -1;
    const gfsmp::LevelLayout &h = s->lay.level[level];
    if (level >= 1 && s->lay.device_tables && s->tab_stats) {
        gf_smp *ms = const_cast<gf_smp *>(s);   // (lazily: the statistics words the table kernels left on the device)
        if (ms->h_tab_stats.empty()) {
            ms->h_tab_stats.assign((size_t)4 * (s->cfg.nLevels + 1), 0u);
            hipStream_t up = s->upload ? s->upload : s->ctx->stream;
            if (hipMemcpyAsync(&ms->h_tab_stats[0], s->tab_stats, sizeof(unsigned) * ms->h_tab_stats.size(), hipMemcpyDeviceToHost, up) != hipSuccess ||
                hipStreamSynchronize(up) != hipSuccess) {
                ms->h_tab_stats.clear();
                return -1;
            }
        }
        unsigned long long n = 0;
        std::memcpy(&n, &s->h_tab_stats[(size_t)4 * level + 2], sizeof(n));
        return (long long)n;
    }
    if (level == 0 || h.pi.size() != (size_t)h.rows) return h.rows;
    long long n = 0;
    for (size_t i = 0; i < h.pi.size(); ++i) n += h.pi[i] >= 0;
    return n;
}

}  // extern "C"
