// gemm_lds.h -- the LDS operand image of the fp32 MFMA kernels (mixers.hip: the general tiled GEMM; smp_level_c64.hip: the
// weight-gradient kernel of the fused SMP level).  Tile constants, the even/odd k order that makes a lane's operands
// contiguous, the XOR slot swizzle and the lane -> element maps of the transposing stores.
#pragma once
#include <hip/hip_runtime.h>

namespace gf {
namespace lds_image {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int kThreads = 256;
constexpr int LDS_ROW = BK + 4;  // 36 floats = 144 B: 16-byte aligned rows, conflict-free ds_read_b128 across 16 rows
using f16v = __attribute__((ext_vector_type(16))) float;
using f4v = __attribute__((ext_vector_type(4))) float;

// Position of k inside an LDS row: even k first, then odd k.  The f32 MFMA 32x32x2 gives lane (i, h = lane >> 5) the
// operand element k = 2 j + h at step j, so with this order each lane's 16 operands of a BK = 32 tile are contiguous
// (four ds_read_b128 instead of sixteen ds_read_b32).
__device__ __forceinline__ int kpos(int k) { return (k & 1) * (BK / 2) + (k >> 1); }

// Word offset of (row, kp) inside an operand image.  The eight 16-byte slots of a row are XOR-swizzled by a function of
// the row: unswizzled, the transposing stores (four scalar ds_write_b32 per float4, 36-word row stride, bank =
// word % 32) put all 32 lanes of a store group on two banks (SQ_LDS_BANK_CONFLICT was 74 % of the LDS-active cycles);
// with it they are 2-way (B) / 4-way (A^T), the ds_read_b128 fragment reads stay conflict-free (checked by
// enumeration over the instruction's four 16-lane groups), and a slot still holds four consecutive kp.
__device__ __forceinline__ int lds_swz(int row) { return (__builtin_popcount(row & 28) & 1) | ((row >> 4) & 2); }
// k row taken by the q-th group of 16 lanes in the transposing B store: (0,2,1,3) inside every four, so that one 32-lane
// store group holds k and k+2 (kp differs by 1: disjoint banks) instead of k and k+1 (kp differs by 16: same banks)
__device__ __forceinline__ int bscat_k(int q) { return (q & ~3) | ((q & 1) << 1) | ((q >> 1) & 1); }
// row taken by the q-th group of 8 lanes in the float4-along-k stores (two ds_write_b64 per float4).  Pairing rows r and
// r + 4 in a 16-lane store group would make these stores conflict-free too (they are 2-way now), but measured no gain.
__device__ __forceinline__ int rowst_m(int q) { return q; }
// transposing A store (TA): float4 idx -> (m, k).  Eight lanes take 32 consecutive m of one k row (128 B of global
// memory), the next eight lanes the row k + 2, ... so that a 32-lane store group holds four different kp & 3 and eight
// different swizzled slots: 2-way conflicts instead of 4-way with 32 lanes on one k row.
__device__ __forceinline__ void ascat_mk(int idx, int *m, int *k) {
    const int a = (idx & 7) + 8 * ((idx >> 5) & 3), kq = (idx >> 3) & 3, kh = idx >> 7;
    *m = 4 * a;
    *k = 2 * kq + (kh & 1) + 8 * (kh >> 1);
}
__device__ __forceinline__ int lds_at(int row, int kp) { return row * LDS_ROW + ((((kp >> 2) ^ lds_swz(row)) << 2) | (kp & 3)); }

}  // namespace lds_image
}  // namespace gf
