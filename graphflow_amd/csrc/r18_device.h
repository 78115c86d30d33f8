// r18_device.h -- device-side building blocks shared by contract18.hip and smp_fused.hip: 16-byte channel quads,
// buffer addressing, the c-group butterfly, the gated-adjacency LDS image and the small A x table product.
#ifndef GF_R18_DEVICE_H_INCLUDED
#define GF_R18_DEVICE_H_INCLUDED

#include <hip/hip_runtime.h>

namespace gf {
namespace dev {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kK = 18;

using f4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ f4 splat(float x) { return f4{x, x, x, x}; }

// Buffer addressing (cdna guide T8/T20): a wave-uniform 128-bit descriptor in SGPRs + a loop-invariant 32-bit
// per-lane byte offset + a scalar row offset.  Keeps the streaming loops free of 64-bit VALU address arithmetic and
// of the VGPR pairs that flat addressing would pin for every load.  The descriptor base must be provably uniform
// (built from kernel arguments and blockIdx only).
using u4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, size_t bytes) {
    const unsigned n = bytes > 0xfffffffcull ? 0xfffffffcu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ f4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff_bytes, soff_bytes, 0);
}

// Streaming (non-temporal) forms for data that is touched ONCE by a kernel -- the big operands it streams in and the results it streams
// out: `nt` on BOTH sides of a streaming kernel is worth 8 - 10 % of a copy's rate on MI355X (tools/micro/copy_probe.hip: 5.5 -> 6.0 - 6.3
// TB/s; `nt` on the loads alone or on the stores alone: nothing), tables and workspaces that are re-read soon keep the default policy.
// GF_NT=0 builds the plain forms (A/B runs: tools/ab_lib.sh).
#ifndef GF_NT
#define GF_NT 1
#endif
__device__ __forceinline__ f4 ld4_nt(const float *p) {
#if GF_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
#else
    return ld4(p);
#endif
}
__device__ __forceinline__ void st4_nt(float *p, f4 v) {
#if GF_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
#else
    st4(p, v);
#endif
}
__device__ __forceinline__ f4 buf_ld4_nt(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, GF_NT ? 2 : 0));   // (aux bit 1 = nt)
}
__device__ __forceinline__ void buf_st4_nt(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff_bytes, soff_bytes, GF_NT ? 2 : 0);
}

__device__ __forceinline__ f4 shfl_xor4(f4 v, int m) {
    f4 r;
    r.x = __shfl_xor(v.x, m);
    r.y = __shfl_xor(v.y, m);
    r.z = __shfl_xor(v.z, m);
    r.w = __shfl_xor(v.w, m);
    return r;
}

// x + (x of lane ^ 16) and x + (x of lane ^ 32) on the VALU.  gfx950's v_permlane16_swap / v_permlane32_swap exchange the
// odd 16-lane rows (the upper 32 lanes) of one register with the even rows (the lower 32 lanes) of another; fed two copies of
// x they return {rows 0,0,2,2 | rows 1,1,3,3} ({lower, lower | upper, upper}), whose sum is the butterfly step in every
// lane.  __shfl_xor compiles to ds_bpermute_b32, which goes through the LDS crossbar: the slab kernels issue 16 of those
// per row, and that -- not HBM -- was what bounded them.
__device__ __forceinline__ float xor16_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the c-groups of a wave (lanes that share the same channel quad): xor-butterfly over lane bits >= log2(LPC).
template <int LPC>
__device__ __forceinline__ f4 reduce_cgroups(f4 v) {
#pragma unroll
    for (int m = LPC; m < 16; m <<= 1) v += shfl_xor4(v, m);
    if (LPC <= 16) {
        v.x = xor16_sum(v.x);
        v.y = xor16_sum(v.y);
        v.z = xor16_sum(v.z);
        v.w = xor16_sum(v.w);
    }
    if (LPC <= 32) {
        v.x = xor32_sum(v.x);
        v.y = xor32_sum(v.y);
        v.z = xor32_sum(v.z);
        v.w = xor32_sum(v.w);
    }
    return v;
}

// LDS carve shared by the four kernels: gated adjacency with row stride N+1 (bank-conflict-free column walks),
// its row sums r[d], then tot = sum A+, tr = trace A+.
struct AdjLds {
    float *A;   // [N][N+1]
    float *r;   // [Np]
    float *st;  // [4]: tot, tr
    __device__ __forceinline__ float at(int d, int e, int N) const { return A[d * (N + 1) + e]; }
};

__host__ __device__ __forceinline__ int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ __forceinline__ int adj_lds_floats(int N) { return pad4(N * (N + 1)) + pad4(N) + 4; }

// Loads A[g] (transposed when TR) with RisiContraction_18's `adj_value > 0` gate (RisiContraction_18.h:90,345) and
// derives r[d] = sum_e A+[d][e] (always the row sums of the UNtransposed matrix), tot and tr.
template <bool TR>
__device__ __forceinline__ AdjLds load_adjacency(float *smem, const float *__restrict__ Ag, int N) {
    AdjLds L;
    L.A = smem;
    L.r = smem + pad4(N * (N + 1));
    L.st = L.r + pad4(N);
    const int tid = threadIdx.x;
    for (int i = tid; i < N * N; i += kThreads) {
        const int d = i / N, e = i - d * N;
        float a = Ag[i];
        a = (a > 0.f) ? a : 0.f;
        if (TR)
            L.A[e * (N + 1) + d] = a;
        else
            L.A[d * (N + 1) + e] = a;
    }
    __syncthreads();
    if (tid < N) {
        float s = 0.f;
        for (int e = 0; e < N; ++e) s += TR ? L.A[e * (N + 1) + tid] : L.A[tid * (N + 1) + e];
        L.r[tid] = s;
    }
    __syncthreads();
    if (tid < 64) {  // tot and tr by one wave: strided partials, then a 6-step butterfly
        float t = 0.f, d = 0.f;
        for (int i = tid; i < N; i += 64) {
            t += L.r[i];
            d += L.A[i * (N + 1) + i];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            t += __shfl_xor(t, m);
            d += __shfl_xor(d, m);
        }
        if (tid == 0) {
            L.st[0] = t;
            L.st[1] = d;
        }
    }
    __syncthreads();
    return L;
}

// The adjacency image alone, for callers that have the gated row sums precomputed (the SMP driver: `rsum` of the level) and
// never read tot / tr: no reductions and NO barrier in here -- the caller's next __syncthreads() publishes the image (the
// combine kernels run one workgroup per (node, x), 180 000 of them per level: three barriers and two serial folds each were
// a fifth of their time).
template <bool TR>
__device__ __forceinline__ AdjLds load_adjacency_lite(float *smem, const float *__restrict__ Ag, const float *__restrict__ rsum, int N) {
    AdjLds L;
    L.A = smem;
    L.r = smem + pad4(N * (N + 1));
    L.st = L.r + pad4(N);
    const int tid = threadIdx.x;
    for (int i = tid; i < N * N; i += kThreads) {
        const int d = i / N, e = i - d * N;
        float a = Ag[i];
        a = (a > 0.f) ? a : 0.f;
        if (TR)
            L.A[e * (N + 1) + d] = a;
        else
            L.A[d * (N + 1) + e] = a;
    }
    for (int i = tid; i < N; i += kThreads) L.r[i] = rsum[i];
    return L;
}

// out[y] (y = first; first+step; ...) = sum_e M[y][e] * T[e][channel quad], for up to three tables at once.
// M is the LDS adjacency (already transposed if the caller needs A^T); rows are walked with stride N+1.
template <int NT, int CW>
__device__ __forceinline__ void small_matvec(const AdjLds &L, int N, int y, int fl, const float *const (&T)[NT],
                                             f4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = splat(0.f);
    const float *row = L.A + y * (N + 1);
    for (int e = 0; e < N; ++e) {
        const float w = row[e];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += w * ld4(T[t] + e * CW + 4 * fl);
    }
}

// sum_{y in [lo,hi)} w(y) * ld4(base + y*stride): the loads of a batch of 8 are all issued before the first add, so a
// length-N reduction costs ceil(N/8) memory latencies instead of N.
template <typename W>
__device__ __forceinline__ f4 batched_sum(const float *base, size_t stride, int lo, int hi, W weight) {
    f4 s = splat(0.f);
    for (int y0 = lo; y0 < hi; y0 += 8) {
        f4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = (y0 + j < hi) ? y0 + j : lo;  // clamped re-read, weight forced to 0 below
            t[j] = ld4(base + (size_t)y * stride);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += ((y0 + j < hi) ? weight(y0 + j) : 0.f) * t[j];
    }
    return s;
}


// Where a workgroup works.  Uniform batches (pair_node == nullptr): graph g = blk / N, index i = blk % N, everything
// at g * N^k.  Ragged batches (the SMP driver): one workgroup per (node, index) "pair"; the node table gives its size
// and the offsets of its P block (in positions = units of C floats), of its rows (N^2 per node: adjacency, Out/G rows,
// N x N workspace tables) and of its first pair (per-(node, index) partial scalars).
struct Ragged {
    const int *pair_node;        // [pairs] or nullptr
    const int *node_s;           // [nodes]
    const long long *node_p;     // [nodes] sum of s^3 before the node
    const long long *node_row;   // [nodes] sum of s^2 before the node
    const long long *node_pair;  // [nodes] sum of s   before the node
    long long pair_base;         // first pair of this launch
    int N;                       // uniform size (uniform batches), maximum size in the launch (ragged)
};
struct Where {
    int N, i, win, node;
    size_t pbase, rowbase, pairbase;
};
__device__ __forceinline__ Where locate(const Ragged &R, int nwin) {
    Where w;
    long long blk = blockIdx.x;
    w.win = (int)(blk % nwin);
    blk /= nwin;
    if (R.pair_node) {
        const long long e = R.pair_base + blk;
        const int n = R.pair_node[e];
        w.N = R.node_s[n];
        w.node = n;
        w.pairbase = (size_t)R.node_pair[n];
        w.i = (int)(e - R.node_pair[n]);
        w.pbase = (size_t)R.node_p[n];
        w.rowbase = (size_t)R.node_row[n];
    } else {
        const int N = R.N;
        const size_t g = (size_t)(blk / N);
        w.N = N;
        w.node = (int)g;
        w.i = (int)(blk % N);
        w.pbase = g * N * N * N;
        w.rowbase = g * N * N;
        w.pairbase = g * N;
    }
    return w;
}


}  // namespace dev
}  // namespace gf
#endif
