// smp_fused.hip -- fused SMP level: promotion + RisiContraction_18 + K-projection + bias + LeakyReLU without ever
// materialising the promoted stack P (sum s^3 C floats) or the 18-slice contraction output Q (18 sum s^2 C floats).
//
// Same mathematics as the op-by-op pipeline of smp.hip (GraphFlow/SMP_omega.h:630-670), regrouped:
//   Q_k = (N x N table) x (factor depending on A)  for every one of the 18 cases, and the
//   K-projection is linear, so   sum_k Q_k K^(k)   is evaluated as
//     tables  T = [S_ab | S_bc | T6 | T10]                   four N x N x C tables per node, built by ONE pass that gathers
//                                                             the promoted tensors straight from f_{l-1}
//     compact G15 = Fd K15, G16 = Fc K16 on the sum-s rows of the level below: the diagonal tables D_bb[x,y] = P[x,y,y] and
//             D_ac[x,y] = P[x,y,x] are gathers of f_{l-1}[w_x][p,p] and f_{l-1}[w_x][p,centre]  (see diag_gather_fwd)
//     GEMMs   O_loc = tot [S_ab|S_bc][K0;K2] + tr S_ab K6 + [T6|T10][K5;K9]     (tot, tr as per-row factors of the operand)
//             Z = [S_ab|S_bc][K8;K12]   Z' = S_ab K11                              8 C x C block products on the rows, not 18
//             V = [rowsum_a|colsum_b|D8|D11][K1;K3;K7;K10]  (per (node,x))   S = [total|s14|s15|s18][K4;K13;K14;K17] (per node)
//     combine f_l[x,y] = LeakyReLU(b + O_loc[x,y] + sum_e A[y,e] (Z[x,e] + Z'[e,x] + G15[x,e] + G16[e,x]) + r[y] V[x] + A[x,y] S)
// The reverse sweep mirrors it: combine-backward -> compact gradients -> block GEMMs (dT, dK) -> smp_bwd_gather, the
// consumer gather of df_{l-1} that evaluates dP from the table gradients on the fly (dP is not materialised either; the
// two-kernel form tables-backward -> promote_backward remains for GF_SMP_BWD_GATHER=0 and receptive fields > 32).
// HBM traffic per level drops from about (2 S + 40 R) C floats to about 19 R C, GEMM flops from 18 to 8 units
// (R = sum s^2, S = sum s^3).
#include <algorithm>
#include <type_traits>

#include "r18_device.h"
#include "smp_internal.h"

using namespace gf::dev;

namespace gf {
namespace {

bool env_is(const char *name, char v) {
    const char *e = std::getenv(name);
    return e && e[0] == v;
}

// Totals of two tables over the four c-groups (16-lane rows) of a wave in ONE butterfly: v_permlane16_swap of (u, v) leaves
// {u.r0, v.r0, u.r2, v.r2} and {u.r1, v.r1, u.r3, v.r3}, whose sum pairs rows 0+1 and 2+3 of u in the even rows and of v in
// the odd rows; the xor-32 step finishes both.  Even c-groups end up with the total of u, odd ones with the total of v:
// half the lane traffic of reducing each table to all lanes.
__device__ __forceinline__ f4 fold_two_tables(f4 u, f4 v) {
    f4 z;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[k]), __float_as_uint(v[k]), false, false);
        z[k] = xor32_sum(__uint_as_float(r[0]) + __uint_as_float(r[1]));
    }
    return z;
}

// x of lane ^ 8 (a rotation by eight inside each 16-lane row: one DPP move per dword)
__device__ __forceinline__ f4 dpp_xor8(f4 v) {
    f4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
    return r;
}

__device__ __forceinline__ f4 dpp_ror4(f4 v) {   // x of lane + 4 inside each 16-lane row (row_ror:4)
    f4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x124 /* row_ror:4 */, 0xf, 0xf, true));
    return r;
}

constexpr float kAlphaF = 0.01f;
__device__ __forceinline__ float lreluf(float z) { return z > 0.f ? z : kAlphaF * z; }

// column blocks of the table matrix T [rows][4C]
enum { T_SAB = 0, T_SBC = 1, T_T6 = 2, T_T10 = 3, T_COLS = 4 };
// column blocks of the projected matrix O [rows][3C].  O_LOC = tot O_tot + tr O_tr + O_dir: the per-node factors tot and
// tr (the level's rowscale table) are applied to the T operand inside the GEMM, so the three row-local products
// accumulate into one block.
enum { O_LOC = 0, O_Z = 1, O_ZP = 2, O_COLS = 3 };
// stacked weight layout: position p holds block K^(kperm[p]); groups are contiguous
//   [0,2) tot | [2,3) tr | [3,5) dir | [5,8) Z | [8,10) Z' | [10,14) V | [14,18) S
__constant__ int c_kperm[18] = {0, 2, 6, 5, 9, 8, 12, 11, 15, 16, 1, 3, 7, 10, 4, 13, 14, 17};

// ---------------------------------------------------------------------------------------------------------------
// T1w: tables-forward (s <= 4 NI, NI in {1, 2, 4, 8}).  One WAVE per (node, b): a workgroup covers four consecutive b of one
// node and shares the node's selection maps and row sums in LDS.  Lane = (c-group cg, channel quad fl) as in r18_fwd_slab;
// every row a of the slab P[:, b, :, :] is GATHERED straight from f_{l-1}[src(n, a)] through the selection map pi_a
// (P[a][b][c] = F_a[pi_a(b)][pi_a(c)] or 0), one row prefetched ahead.  A wave walks all rows a itself, so the sums over a
// stay in its registers: no cross-wave reduction, one barrier in the whole kernel.  (A workgroup-per-pair variant with the
// rows split over four waves paid about ten barriers of prologue/epilogue and was slower in every size class: 3.3 -> 2.6 ms
// for s <= 16, 0.73 -> 0.54 ms for 16 < s <= 32 at cfg3.)
// ---------------------------------------------------------------------------------------------------------------
#ifndef GF_TF_DEPTH2
#define GF_TF_DEPTH2 0
#endif
#ifndef GF_TF_OCC
#define GF_TF_OCC 4   // waves per SIMD of the classes s <= 16 (five: the per-node kernel spills, 0.94 -> 1.08 ms)
#endif
#ifndef GF_TF_WIDE
#define GF_TF_WIDE 16   // classes NI >= this run eight waves per node (16: none)
#endif
#ifndef GF_TF_D2NI
#define GF_TF_D2NI 4
#endif
// VEC (round 4, ALLOK only): the sums over b that smp_vectors used to collect in a pass of its own -- rowsum_a[x] = sum_b S_ab[x, b],
// D8[x] = sum_b P[x, b, b] and the per-node scalars -- are kept as the rows go by: every wave adds its (a, b) terms to LDS accumulators
// of its OWN (two 16-byte read-modify-writes per row, plain DS operations in wave order: no atomics; the lanes outside c-group 0 work
// on a slot nobody reads, so the row loop stays branch-free), the four waves' partials are folded in a fixed order after one barrier.
// tables-forward 0.90 -> 1.04 ms, smp_vectors (0.38 ms a cfg3 step, 1 GB of traffic) gone.  Measured and not kept: one read-modify-write
// per row with c-group 2 taking the diagonal element over lane ^ 32 (1.19 ms: the exchange sits on the row's critical path); the
// row sums only, D8 gathered from the compact diagonal table in the epilogue as smp_vectors did (1.10 ms).
// LPC (round 4): lanes per position -- 16 (64-channel windows) or 8 (32-channel windows: at C = 32 every lane has channels and a wave
// load covers eight positions; the sixteen-lane mapping left half of the lanes idle there).
template <int NI, bool ALLOK, bool VEC = false, int LPC = 16>  // ALLOK: C is a multiple of the window (4 LPC channels) -- every lane has channels
__global__ __launch_bounds__(NI >= GF_TF_WIDE ? 2 * kThreads : kThreads, NI <= 4 ? GF_TF_OCC : 2) void smp_tables_fwd_w(  // (NI = 4 sat at 130 VGPRs: capped to 128 -> 4 waves per SIMD)
    const float *__restrict__ fprev, const float *__restrict__ rsum,
                                                             float *__restrict__ T, float *__restrict__ Vt,
                                                             float *__restrict__ scal, const long long *__restrict__ pair_src_row,
                                                             const int *__restrict__ pair_src_s, const short *__restrict__ pi,
                                                             const int4 *__restrict__ recs,  // two per node, in launch order (build_tf_records)
                                                             int C, int nwin,
                                                             int zeros_kept,  // != 0: the structurally-zero rows (a, b) hold their zeros
                                                             const unsigned char *__restrict__ rowflag,  // with zeros_kept: bit 1 =
                                                             // row (b, c) has data in S_bc / T10 (the others are not stored either)
                                                             float *__restrict__ St) {  // VEC: [nodes][4C] per-node scalars
    static_assert(!VEC || ALLOK, "the folded vector sums need every lane to have channels");
    constexpr int PPW = 64 / LPC, CWIN = 4 * LPC;   // positions per wave load, channels per window
    static_assert(LPC == 16 || ((LPC == 8 || LPC == 4) && ALLOK), "eight / four lanes per position: whole 32- / 16-channel windows only");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = lane / LPC, fl = lane % LPC;
    // launch order: molecule-major inside the size class, one contiguous run of it per XCD (blockIdx % 8): the source
    // tensors f_{l-1}[w] of a molecule are gathered by ~s consumers each, which then share an L2
    unsigned tile;
    {
        const unsigned nb = gridDim.x, qq = nb / 8, r = nb % 8, x = blockIdx.x % 8;
        tile = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + blockIdx.x / 8;
    }
    // ONE workgroup per node (round 3; it was one per four b): measured with every row but a wave's own skipped, two thirds of the
    // kernel's time were the workgroup's fixed costs -- three dependent table reads to find the node, the staging of its maps, the
    // barrier -- paid s / 4 times per node.  The node's record is one 32-byte read, the maps are staged once, and each wave walks
    // b = wave, wave + 4, ... on its own.
    const int win = (int)(tile % nwin);
    const int4 r0 = recs[2 * (tile / nwin)], r1 = recs[2 * (tile / nwin) + 1];
    const int N = r0.y;
    const size_t rowbase = ((size_t)(unsigned)r1.y << 32) | (unsigned)r1.x, pairbase = ((size_t)(unsigned)r1.w << 32) | (unsigned)r1.z;
    const int f = win * CWIN + 4 * fl;
    const bool fok = f < C;
    const int fld = fok ? f : 0;
    constexpr bool allok = ALLOK;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    // The address chain of a row is LDS reads only, and few of them: per row a the source tensor's descriptor words, per (a, c) the
    // BYTE offset of column pi_a(c) inside a row of the source (kAbsent -- beyond any source tensor, the buffer load returns 0 --
    // where pi_a(c) < 0 and in the padding c >= N), rows padded to the class's 4 NI positions so that a lane's NI reads are
    // unconditional and a constant 16 bytes apart.  (Round 3: the maps were shorts read one at a time behind a lane mask, each waited
    // for, multiplied and selected before its request went out: ~45 VALU and five LDS round trips per row in front of the loads.)
    constexpr int ST = PPW * NI;
    constexpr int kAbsent = 0x40000000;
    float *sR = smem;                                               // [N]
    int4 *sRow = reinterpret_cast<int4 *>(smem + ((N + 3) & ~3));   // [N] {address lo, hi, bytes, s_w} of f_{l-1}[src(n, a)]
    int *sOff = reinterpret_cast<int *>(sRow + N);                  // [N][ST]
    const int nthreads = (NI >= GF_TF_WIDE ? 2 : 1) * kThreads;   // (eight waves per node in the wide classes)
    for (int i = tid; i < N; i += nthreads) {
        sR[i] = rsum[pairbase + i];
        const int sw = pair_src_s[pairbase + i];
        const unsigned long long addr = reinterpret_cast<unsigned long long>(fprev + pair_src_row[pairbase + i] * C);
        sRow[i] = make_int4((int)(unsigned)addr, (int)(unsigned)(addr >> 32), sw * sw * C * 4, sw);
    }
    for (int i = tid; i < N * ST; i += nthreads) {
        const int a = i / ST, c = i % ST;
        const int p = (c < N) ? pi[rowbase + a * N + c] : -1;
        sOff[i] = p >= 0 ? p * C * 4 : kAbsent;
    }
    unsigned char *sFlag = reinterpret_cast<unsigned char *>(sOff + N * ST);  // [N][N] the node's row flags
    const bool skip_bc = zeros_kept && rowflag;
    if (skip_bc)
        for (int i = tid; i < N * N; i += nthreads) sFlag[i] = rowflag[rowbase + i];
    // VEC: per wave, [N][rowsum | D8][64 floats] accumulators and a 64-float slot the lanes outside c-group 0 read and write instead
    f4 *sAcc = reinterpret_cast<f4 *>(smem + ((((N + 3) & ~3) + 4 * N + N * ST + ((N * N + 3) >> 2) + 3) & ~3));
    const int nwv = nthreads / 64;
    f4 *wacc = sAcc + (size_t)wave * N * (2 * LPC), *wdummy = sAcc + (size_t)nwv * N * (2 * LPC) + wave * (2 * LPC);
    if constexpr (VEC)
        for (int i = lane; i < N * (2 * LPC); i += 64) wacc[i] = splat(0.f);
    __syncthreads();

    float rc[NI];
    int cc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * PPW + cg;
        cc[i] = (c < N) ? c : -1;
        rc[i] = (c < N && fok) ? sR[c] : 0.f;
    }
    for (int b = wave; b < N; b += nthreads / 64) {
    f4 sbc[NI], t10[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) sbc[i] = t10[i] = splat(0.f);
    f4 dgsum = splat(0.f);

    // Row a of the slab through a buffer descriptor of the source tensor f_{l-1}[src(n, a)] (wave-uniform base and row offset,
    // 32-bit lane offsets): a structurally-zero position gets an out-of-range offset and the hardware returns 0 -- no 64-bit
    // address arithmetic and no selects in the loop.  Only rows with pi_a(b) >= 0 are ever requested (`present` below), and every
    // lane requests: no branch, nothing for the compiler to lose count of the memory queue over.
    const int lane_off = fok ? fld * 4 : kAbsent;        // (lanes without channels -- C % 64 != 0 -- fetch nothing)
    const int diag_off = (cg < 2) ? lane_off : kAbsent;  // (the diagonal entries are c-groups 0 and 1's)
    auto load_row = [&](int a, f4(&v)[NI], f4 &dg) {
        const int4 ri = sRow[a];
        const int *orow = sOff + a * ST;
        const int ob = orow[b];
        int oc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) oc[i] = orow[cg + PPW * i];
        const int od = orow[(cg == 0) ? b : a];  // pi_a(b) for c-group 0, pi_a(a) for the others
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ri.x), hi = __builtin_amdgcn_readfirstlane((unsigned)ri.y);
        const int bytes = __builtin_amdgcn_readfirstlane(ri.z), sw = __builtin_amdgcn_readfirstlane(ri.w);
        const float *src = reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (size_t)(unsigned)bytes);
        const int rowoff = __builtin_amdgcn_readfirstlane(ob) * sw;  // row pi_a(b) of the source: pi_a(b) C 4 bytes x s_w
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = buf_ld4(rs, oc[i] + lane_off, rowoff);
        dg = buf_ld4(rs, od + diag_off, rowoff);
    };

    // two row buffers used alternately (the loop is unrolled by two: no register copies between rows)
    f4 bufA[NI], bufB[NI], dA, dB;
    // (after fold_two_tables the EVEN 16-lane rows of the wave hold the S_ab total, the odd rows the T6 total)
    float *const tcol = T + (rowbase + b) * (size_t)(T_COLS * C) + f + (((lane >> 4) & 1) ? T_T6 : T_SAB) * C;  // (+ a N ldt per row)
    const size_t tstep = (size_t)N * (T_COLS * C);
    // Rows (a, b) whose source does not contain vertex b (pi_a(b) < 0) are structurally zero -- about 40 % of them at QM9
    // sizes: they are never loaded or summed, only their two table entries are written as zeros.  `present` is wave-uniform
    // (b is the wave's, a the loop's): bit a = row a has data.  The row of the node's own vertex (a == b) always has.
    unsigned present = (unsigned)__ballot(lane < N && sOff[(lane < N ? lane : 0) * ST + b] != kAbsent);
    present = __builtin_amdgcn_readfirstlane(present);
    if (allok) {
        if (!zeros_kept)  // (the zeros of these rows are already in the buffer, written once for this batch: DevLevel::t_zeros)
            for (unsigned z = ~present & (N >= 32 ? 0xffffffffu : ((1u << N) - 1u)); z; z &= z - 1)
                st4(tcol + (__builtin_ctz(z)) * tstep, splat(0.f));
    } else if (fok && cg < 2) {
        for (unsigned z = ~present & (N >= 32 ? 0xffffffffu : ((1u << N) - 1u)); z; z &= z - 1)
            st4(T + (rowbase + (size_t)__builtin_ctz(z) * N + b) * (size_t)(T_COLS * C) + f + (cg ? T_T6 : T_SAB) * C, splat(0.f));
    }
    auto row_step = [&](int a, int an, const f4(&cur)[NI], const f4 &dcur, f4(&nxt)[NI], f4 &dnxt, auto own_row) {
        load_row(an >= 0 ? an : a, nxt, dnxt);
        const float ra = sR[a];
        f4 sab = splat(0.f), t6 = splat(0.f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const f4 v = cur[i];
            sbc[i] += v;
            t10[i] += ra * v;
            sab += v;
            t6 += rc[i] * v;
        }
        if constexpr (LPC <= 8) {   // the two c-groups of a 16-lane row first (lane ^ 8: a rotation by eight inside the row, on the VALU)
            sab += dpp_xor8(sab);
            t6 += dpp_xor8(t6);
        }
        if constexpr (LPC == 4) {   // ... four of them: the pairs' sums rotated by four (v[l] + v[l+8] + v[l+4] + v[l+12])
            sab += dpp_ror4(sab);
            t6 += dpp_ror4(t6);
        }
        const f4 both = fold_two_tables(sab, t6);  // even 16-lane rows: S_ab[a,b], odd rows: T6[a,b]
        dgsum += dcur;
        if (allok) {
            // every lane stores (c-groups 0/2 the S_ab block, 1/3 the T6 block; the pairs write identical values to the same
            // address): a store that all paths issue can be COUNTED by the compiler, so waiting for the next row's loads
            // (vmcnt is in order over loads and stores) need not include it; lane-conditional stores cannot (-5 %)
            gf_st_s<32>(reinterpret_cast<f4 *>(tcol + a * tstep), both);  // table row (a, b)
            if constexpr (VEC) {   // c-group 0 holds S_ab[a, b] and P[a, b, b]: into the wave's accumulators of row a
                f4 *slot = (cg == 0) ? wacc + a * (2 * LPC) + fl : wdummy + fl;
                const f4 r = slot[0] + both, d8 = slot[LPC] + dcur;
                slot[0] = r;
                slot[LPC] = d8;
            }
            if constexpr (decltype(own_row)::value) {  // a == b: the peeled first row
                if (cg == 0) st4(scal + ((pairbase + b) * 4 + 3) * (size_t)C + f, dcur);
                if (cg == 2 * (16 / LPC)) st4(scal + ((pairbase + b) * 4 + 1) * (size_t)C + f, both);   // (a c-group of an even row other than 0)
            }
        } else if (fok) {
            float *trow = T + (rowbase + (size_t)a * N + b) * (size_t)(T_COLS * C) + f;  // table row (a, b)
            if (cg == 0) {
                st4(trow + T_SAB * C, both);
                if (a == b) st4(scal + ((pairbase + b) * 4 + 3) * (size_t)C + f, dcur);
            } else if (cg == 1) {
                st4(trow + T_T6 * C, both);
            } else if (cg == 2) {
                if (a == b) st4(scal + ((pairbase + b) * 4 + 1) * (size_t)C + f, both);
            }
        }
    };
    {
        auto pop = [&]() {  // next present row, or -1
            if (!present) return -1;
            const int a = __builtin_ctz(present);
            present &= present - 1;
            return a;
        };
        if (ALLOK) {
            // Row a = b (the node's own vertex: always present) first, outside the loop, with its two extra stores; the loop over
            // the other rows then has no conditional store and is entered with the memory queue its back edge leaves (a row's
            // requests and one store): the compiler counts the queue (vmcnt(N)) instead of draining it -- stores included --
            // at the top of every pair of rows.  (Row order changes the order of the sums over a, not their terms.)
            present &= ~(1u << b);
            if constexpr (GF_TF_DEPTH2 && NI <= GF_TF_D2NI) {
                // TWO rows requested ahead (three buffers in rotation; round 3 -- the offset maps above freed the registers): a wave
                // spent ~2 us per row, the round trip of a gather under load, with one row in flight
                f4 bufC[NI], dC;
                int a1 = pop(), a2 = pop();
                load_row(b, bufA, dA);
                load_row(a1 >= 0 ? a1 : b, bufB, dB);
                row_step(b, a2, bufA, dA, bufC, dC, std::true_type{});
                while (a1 >= 0) {  // (a1 waits in B, a2 in C)
                    const int a3 = pop();
                    row_step(a1, a3, bufB, dB, bufA, dA, std::false_type{});
                    if (a2 < 0) break;
                    const int a4 = pop();
                    row_step(a2, a4, bufC, dC, bufB, dB, std::false_type{});
                    if (a3 < 0) break;
                    const int a5 = pop();
                    row_step(a3, a5, bufA, dA, bufC, dC, std::false_type{});
                    a1 = a4;
                    a2 = a5;
                }
            } else {
            int a = b, an = pop();
            load_row(a, bufA, dA);
            row_step(a, an, bufA, dA, bufB, dB, std::true_type{});
            while (an >= 0) {
                a = an;
                an = pop();
                row_step(a, an, bufB, dB, bufA, dA, std::false_type{});
                if (an < 0) break;
                a = an;
                an = pop();
                row_step(a, an, bufA, dA, bufB, dB, std::false_type{});
            }
            }
        } else {
            int a = pop();
            load_row(a, bufA, dA);
            for (;;) {
                int an = pop();
                row_step(a, an, bufA, dA, bufB, dB, std::false_type{});
                if (an < 0) break;
                a = an;
                an = pop();
                row_step(a, an, bufB, dB, bufA, dA, std::false_type{});
                if (an < 0) break;
                a = an;
            }
        }
    }
    f4 cs = splat(0.f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (cc[i] >= 0) {
            cs += sbc[i];
            if (fok && (!skip_bc || (sFlag[b * N + cc[i]] & 2))) {
                float *trow = T + (rowbase + (size_t)b * N + cc[i]) * (size_t)(T_COLS * C) + f;  // table row (b, c)
                gf_st_s<32>(reinterpret_cast<f4 *>(trow + T_SBC * C), sbc[i]);
                gf_st_s<32>(reinterpret_cast<f4 *>(trow + T_T10 * C), t10[i]);
            }
        }
    }
    cs = reduce_cgroups<LPC>(cs);
    // diagonal sums: lanes of c-group 0 hold sum_a P[a,b,b], c-group 1 holds sum_a P[a,b,a]
    const f4 dactot = shfl_xor4(dgsum, LPC);  // c-group 0 lanes receive c-group 1's sum
    if (cg == 0 && fok) {
        float *v = Vt + (pairbase + b) * 4 * (size_t)C + f;
        st4(v + 1 * C, cs);
        st4(v + 3 * C, dactot);
        float *sc = scal + (pairbase + b) * 4 * (size_t)C + f;
        st4(sc + 0 * C, cs);
        st4(sc + 2 * C, dgsum);
    }
    }  // b
    if constexpr (VEC) {
        __syncthreads();   // (every wave's accumulators and partial scalars are complete and visible to the workgroup)
        const int node = r0.x;
        for (int i = tid; i < N * (2 * LPC); i += nthreads) {   // (row a, rowsum | D8, float4 q): the waves' partials in wave order
            const int a = i / (2 * LPC), rem = i % (2 * LPC);
            f4 v = sAcc[(size_t)a * (2 * LPC) + rem];
            for (int w = 1; w < nwv; ++w) v += sAcc[((size_t)w * N + a) * (2 * LPC) + rem];
            st4(Vt + (pairbase + a) * 4 * (size_t)C + (rem / LPC) * 2 * C + win * CWIN + 4 * (rem % LPC), v);
        }
        if (tid < 4 * LPC) {   // per-node scalars: the sum over b of the partials, in the order of b (as smp_vectors formed it)
            const int k = tid / LPC, q = tid % LPC;
            const auto one = [](int) { return 1.f; };
            st4(St + (size_t)node * 4 * C + k * C + win * CWIN + 4 * q,
                batched_sum(scal + pairbase * 4 * (size_t)C + k * C + win * CWIN + 4 * q, (size_t)4 * C, 0, N, one));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// tables-forward for the FEW nodes above 32 positions (33 .. 64; round 6): smp_tables_fwd_w keeps a wave's sums over a in registers by
// consumer position -- 4 NI positions, NI <= 8 -- and a 48-atom molecule's level-3 fields reach 35.  One workgroup per (node, b), plain
// loops, every sum in index order; the same outputs as smp_tables_fwd_w<NI, true, false> (the sums over b are smp_vectors', launched
// over these nodes behind it):
//   T[(a, b)] = [S_ab | . | T6 | .]  for the rows with an image (pi_a(b) >= 0; the others: zeros unless the level keeps them masked)
//   T[(b, c)] = [. | S_bc | . | T10] for the rows some source covers (all of them when the level does not mask)
//   Vt[(n, b)] blocks 1, 3 = sum_c S_bc[b, c], sum_a P[a, b, a];  scal[(n, b)] = {sum_c S_bc[b, c], S_ab[b, b], sum_a P[a, b, b], P[b, b, b]}
// with P[a, b, c] = f_{l-1}[w_a][pi_a(b)][pi_a(c)] or 0 (SMP_omega.h:641-651, RisiContraction_18.h:98-322 factorised as in DESIGN.md 4.5).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kBigN = 64;
__global__ __launch_bounds__(256) void smp_tables_fwd_big(const float *__restrict__ fprev, const float *__restrict__ rsum, float *__restrict__ T,
                                                          float *__restrict__ Vt, float *__restrict__ scal,
                                                          const long long *__restrict__ pair_src_row, const int *__restrict__ pair_src_s,
                                                          const short *__restrict__ pi, const int *__restrict__ pair_node,
                                                          const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                                          const long long *__restrict__ node_pair, long long pair0, int C, int zeros_kept,
                                                          const unsigned char *__restrict__ rowflag) {
    const long long e = pair0 + blockIdx.x;
    const int n = pair_node[e], N = node_s[n];
    const size_t rowbase = (size_t)node_row[n], pairbase = (size_t)node_pair[n];
    const int b = (int)(e - (long long)pairbase), nl = C >> 2, tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) float big_smem[];
    float *sBC = big_smem;                       // [N][C] S_bc[b, c], then: P[a, b, b]
    float *sAC = sBC + (size_t)N * C;            // [N][C] P[a, b, a]
    float *sR = sAC + (size_t)N * C;             // [N]
    long long *sSrc = reinterpret_cast<long long *>(sR + ((N + 3) & ~3));   // [N] first float of f_{l-1}[w_a]
    int *sSw = reinterpret_cast<int *>(sSrc + N);                           // [N]
    short *sPi = reinterpret_cast<short *>(sSw + N);                        // [N][N] pi_a(x)
    for (int i = tid; i < N; i += 256) {
        sR[i] = rsum[pairbase + i];
        sSrc[i] = pair_src_row[pairbase + i] * C;
        sSw[i] = pair_src_s[pairbase + i];
    }
    for (int i = tid; i < N * N; i += 256) sPi[i] = pi[rowbase + i];
    __syncthreads();
    const bool skip_bc = zeros_kept && rowflag;
    // rows (b, c): sums over the sources a
    for (int it = tid; it < N * nl; it += 256) {
        const int c = it / nl, f = 4 * (it - c * nl);
        f4 sbc = splat(0.f), t10 = splat(0.f);
        for (int a = 0; a < N; ++a) {
            const int pb = sPi[a * N + b], pc = sPi[a * N + c];
            if (pb >= 0 && pc >= 0) {
                const f4 v = ld4(fprev + sSrc[a] + ((size_t)pb * sSw[a] + pc) * C + f);
                sbc += v;
                t10 += sR[a] * v;
            }
        }
        st4(sBC + (size_t)c * C + f, sbc);
        if (!skip_bc || (rowflag[rowbase + (size_t)b * N + c] & 2)) {
            float *trow = T + (rowbase + (size_t)b * N + c) * (size_t)(T_COLS * C) + f;
            st4(trow + T_SBC * C, sbc);
            st4(trow + T_T10 * C, t10);
        }
    }
    __syncthreads();
    f4 cs = splat(0.f);
    if (tid < nl)
        for (int c = 0; c < N; ++c) cs += ld4(sBC + (size_t)c * C + 4 * tid);
    __syncthreads();   // (sBC is reused below)
    // rows (a, b): sums over c; the two diagonal elements of the row on the way
    for (int it = tid; it < N * nl; it += 256) {
        const int a = it / nl, f = 4 * (it - a * nl);
        const int pb = sPi[a * N + b];
        f4 sab = splat(0.f), t6 = splat(0.f), dbb = splat(0.f), dac = splat(0.f);
        if (pb >= 0) {
            const float *src = fprev + sSrc[a] + (size_t)pb * sSw[a] * C + f;
            for (int c = 0; c < N; ++c) {
                const int pc = sPi[a * N + c];
                if (pc >= 0) {
                    const f4 v = ld4(src + (size_t)pc * C);
                    sab += v;
                    t6 += sR[c] * v;
                    if (c == b) dbb = v;
                    if (c == a) dac = v;
                }
            }
        }
        if (pb >= 0 || !zeros_kept) {
            float *trow = T + (rowbase + (size_t)a * N + b) * (size_t)(T_COLS * C) + f;
            st4(trow + T_SAB * C, sab);
            st4(trow + T_T6 * C, t6);
        }
        st4(sBC + (size_t)a * C + f, dbb);
        st4(sAC + (size_t)a * C + f, dac);
        if (a == b) {   // the node's own vertex: always present
            st4(scal + ((pairbase + b) * 4 + 3) * (size_t)C + f, dbb);
            st4(scal + ((pairbase + b) * 4 + 1) * (size_t)C + f, sab);
        }
    }
    __syncthreads();
    if (tid < nl) {
        const int f = 4 * tid;
        f4 dgsum = splat(0.f), dactot = splat(0.f);
        for (int a = 0; a < N; ++a) {
            dgsum += ld4(sBC + (size_t)a * C + f);
            dactot += ld4(sAC + (size_t)a * C + f);
        }
        float *v = Vt + (pairbase + b) * 4 * (size_t)C + f;
        st4(v + 1 * C, cs);
        st4(v + 3 * C, dactot);
        float *sc = scal + (pairbase + b) * 4 * (size_t)C + f;
        st4(sc + 0 * C, cs);
        st4(sc + 2 * C, dgsum);
    }
}
static size_t tables_big_lds(int N, int C) {
    return sizeof(float) * (2 * (size_t)N * C + ((N + 3) & ~3)) + sizeof(long long) * N + sizeof(int) * N + sizeof(short) * (size_t)N * N + 32;
}

// rowsum_a[x] = sum_b S_ab[x,b], D8[x] = sum_b Dbb[x,b] per (node, x); scalars per node = sum over b of the partials.
// Items = (x, float4 lane); the s loads of an item are issued in batches of 8 (batched_sum).
__global__ __launch_bounds__(256) void smp_vectors(const float *__restrict__ T, float *__restrict__ Vt,
                                                   const float *__restrict__ scal, float *__restrict__ St,
                                                   const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                                   const long long *__restrict__ node_pair, int C,
                                                   const float *__restrict__ Fdc, const long long *__restrict__ pair_src_pair,
                                                   const short *__restrict__ pi) {
    const int n = blockIdx.x;
    const int s = node_s[n], nl = C / 4;
    const size_t rowbase = (size_t)node_row[n], pairbase = (size_t)node_pair[n];
    const auto one = [](int) { return 1.f; };
    for (int i = threadIdx.x; i < s * nl; i += blockDim.x) {
        const int fl = i % nl, x = i / nl;
        const float *t = T + (rowbase + (size_t)x * s) * (size_t)(T_COLS * C) + T_SAB * C + 4 * fl;
        f4 rs = splat(0.f), d8 = splat(0.f);
        {  // D8[x] = sum_b P[x,b,b] = sum over the images pi_x(b) of f_{l-1}[w_x][p,p]  (compact table Fdc); the rows (x, b) with
           // pi_x(b) < 0 are structurally zero in S_ab as well: only the others are read (half of the 0.73 GB block at QM9 sizes)
            const float *fd = Fdc + (size_t)pair_src_pair[pairbase + x] * 2 * C + 4 * fl;
            const short *map = pi + rowbase + (size_t)x * s;
            for (int b0 = 0; b0 < s; b0 += 8) {
                f4 v[8], u[8];
                bool ok[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = (b0 + j < s) ? map[b0 + j] : -1;
                    ok[j] = p >= 0;
                    v[j] = ld4(fd + (size_t)(ok[j] ? p : 0) * 2 * C);
                    u[j] = ld4(ok[j] ? t + (size_t)(b0 + j) * (T_COLS * C) : fd);  // (absent: a line the lane has just read)
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (ok[j]) d8 += v[j], rs += u[j];
            }
        }
        st4(Vt + (pairbase + x) * 4 * (size_t)C + 0 * C + 4 * fl, rs);
        st4(Vt + (pairbase + x) * 4 * (size_t)C + 2 * C + 4 * fl, d8);
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x)  // 4C floats = C float4
        st4(St + (size_t)n * 4 * C + 4 * i, batched_sum(scal + pairbase * 4 * (size_t)C + 4 * i, (size_t)4 * C, 0, s, one));
}

// ---------------------------------------------------------------------------------------------------------------
// Compact-diagonal variant.  D_bb[x,y] = P[x,y,y] = f_{l-1}[w_x][p,p] and D_ac[x,y] = P[x,y,x] = f_{l-1}[w_x][p,c] with
// p = pi_x(y) and c = the position of w_x's own vertex in its receptive field: both tables are gathers of sum-s vectors per
// source node.  So D_bb K15 and D_ac K16 are computed ONCE on those vectors ([pairs of level l-1] rows instead of
// [rows of level l]: 24x fewer at level 3 of cfg3), combine-forward gathers the products, and the reverse sweep collects
// their gradients with a consumer gather.  T and dT lose two of their six blocks and the row GEMMs two of ten products.
//   Fdc[pr] = [ f[w][p,p] | f[w][p,c_w] ]   (pr = node_pair(l-1)[w] + p)        Gc = [ Fd K15 | Fc K16 ]
// ---------------------------------------------------------------------------------------------------------------
__global__ void diag_gather_fwd(const float *__restrict__ fprev, float *__restrict__ Fdc, const int *__restrict__ node_s,
                                const long long *__restrict__ node_row, const long long *__restrict__ node_pair,
                                const int *__restrict__ node_center, int C) {
    const int w = blockIdx.x;
    const int s = node_s[w], c = node_center[w], nl = C / 4;
    const float *src = fprev + (size_t)node_row[w] * C;
    float *dst = Fdc + (size_t)node_pair[w] * 2 * C;
    for (int i = threadIdx.x; i < s * nl; i += blockDim.x) {
        const int fl = i % nl, p = i / nl;
        st4(dst + (size_t)p * 2 * C + 4 * fl, ld4(src + ((size_t)p * s + p) * C + 4 * fl));
        st4(dst + (size_t)p * 2 * C + C + 4 * fl, ld4(src + ((size_t)p * s + c) * C + 4 * fl));
    }
}

// dGc[pr] = [ sum_cons dU_n[a, inv(p)] | sum_cons dU_n[inv(p), a] ]  over the consumers (n, a) of source node w, in consumer
// order (deterministic).  dU_n[x, e] is the Z block of dO at row (x, e) (written by combine-backward).
// (individual __restrict__ kernel parameters, also for the launch that shares this body with smp_reduce_pairs: handed over in a struct the
//  pointers lose their no-alias guarantee -- measured in round 6: 0.20 -> 0.26 ms per cfg3 step for this kernel)
#define GF_DIAGB_PARAMS                                                                                                               \
    const float *__restrict__ dO, float *__restrict__ dGc, const int *__restrict__ prev_s, const long long *__restrict__ prev_pair,       \
        const long long *__restrict__ cons_ptr, const long long *__restrict__ cons_row, const int *__restrict__ cons_s,                  \
        const int *__restrict__ cons_a, const long long *__restrict__ cons_inv_off, const short *__restrict__ inv, int C, int ocols,     \
        const float *__restrict__ nodefac, /* or null: slice-dropout factors [nodes][18] of the CONSUMERS' level */                       \
        const long long *__restrict__ cons_pair, const int *__restrict__ pair_node /* (with nodefac: consumer -> node) */
#define GF_DIAGB_ARGS dO, dGc, prev_s, prev_pair, cons_ptr, cons_row, cons_s, cons_a, cons_inv_off, inv, C, ocols, nodefac, cons_pair, pair_node
__device__ __forceinline__ void diag_gather_bwd_body(GF_DIAGB_PARAMS, int w) {
    const int sw = prev_s[w], nl = C / 4;
    const long long c0 = cons_ptr[w], c1 = cons_ptr[w + 1];
    const size_t ldo = (size_t)ocols * C;
    float *dst = dGc + (size_t)prev_pair[w] * 2 * C;
    for (int i = threadIdx.x; i < sw * nl; i += blockDim.x) {
        const int fl = i % nl, p = i / nl;
        f4 a15 = splat(0.f), a16 = splat(0.f);
        for (long long e0 = c0; e0 < c1; e0 += 4) {  // four consumers' loads in flight (clamped address + select), same order
            f4 u[4], v[4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = (e0 + j < c1) ? e0 + j : e0;
                const int ip = inv[cons_inv_off[e] + p];
                ok[j] = e0 + j < c1 && ip >= 0;
                const int s = cons_s[e], a = cons_a[e], ix = ok[j] ? ip : 0;
                const float *base = dO + (size_t)cons_row[e] * ldo + O_Z * C + 4 * fl;
                u[j] = ld4(base + ((size_t)a * s + ix) * ldo);
                v[j] = ld4(base + ((size_t)ix * s + a) * ldo);
                if (nodefac) {   // (uniform) the consumer node's factors of slices 15 and 16
                    const float *nf = nodefac + (size_t)pair_node[cons_pair[e]] * 18;
                    u[j] = u[j] * nf[15];
                    v[j] = v[j] * nf[16];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok[j]) {
                    a15 += u[j];
                    a16 += v[j];
                }
        }
        st4(dst + (size_t)p * 2 * C + 4 * fl, a15);
        st4(dst + (size_t)p * 2 * C + C + 4 * fl, a16);
    }
}
__global__ void diag_gather_bwd(GF_DIAGB_PARAMS) { diag_gather_bwd_body(GF_DIAGB_ARGS, (int)blockIdx.x); }

// zeros into the S_ab / T6 blocks of the rows (a, b), and the S_bc / T10 blocks of the rows (b, c), that tables-forward never
// writes: once per prepared batch (DevLevel::t_zeros; rowflag bits 0 / 1 = the row has data in the first / second pair of blocks)
constexpr int kZeroFillRows = 64;   // rows per workgroup: eight per pass (a thread was launched per (row, float4): 62 M threads at cfg3's level 3)
template <int CB>   // channels: 64 or 32
__global__ __launch_bounds__(256) void tables_zero_fill(float *__restrict__ T, const unsigned char *__restrict__ rowflag, long long rows) {
    constexpr int NQ = CB / 4, RPP = 256 / (2 * NQ), NP = kZeroFillRows / RPP;   // float4 per block, rows per pass, passes
    const int q = threadIdx.x % (2 * NQ);  // 2 NQ float4 = two CB-column blocks
    const long long row0 = (long long)blockIdx.x * kZeroFillRows + threadIdx.x / (2 * NQ);
    int fl[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {   // (all flags first: one round trip)
        const long long row = row0 + RPP * p;
        fl[p] = row < rows ? rowflag[row] : 3;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long long row = row0 + RPP * p;
        if (!(fl[p] & 1)) st4(T + row * (T_COLS * CB) + (q < NQ ? T_SAB * CB + 4 * q : T_T6 * CB + 4 * (q - NQ)), splat(0.f));
        if (!(fl[p] & 2)) st4(T + row * (T_COLS * CB) + (q < NQ ? T_SBC * CB + 4 * q : T_T10 * CB + 4 * (q - NQ)), splat(0.f));
    }
}

// ---- slice dropout on the fused level (round 5; RisiContraction_18_dropout, GraphFlow/RisiContraction_18_dropout.h:106-132, 465-471) ----
// The reference zeroes (train) or scales by nKept / 18 (test) single slices k of a node's contraction output.  The factorised level never
// forms a slice, but every slice is one block product K^(k) applied to one table, so a node's slice factor m_k is a per-node factor on that
// product: the eight row products take it through an eight-column row-factor table (smp_rowpanel_split / smp_wgrad_direct, NF = 8), the
// vector / scalar products through their operands (Vt, St and their gradients are scaled block by block), the two compact diagonal
// products where the consumer gathers them (combine-forward, diag_gather_bwd).
//   nodefac[n][k] = bit k of keep[n] ? scale : 0        rowfac8[row] = (tot m0, tot m2, tr m6, m5, m9, m8, m12, m11) of the row's node
__global__ __launch_bounds__(64) void build_dropout_factors(const unsigned *__restrict__ keep, float scale, const float2 *__restrict__ node_scale,
                                                            const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                                            float *__restrict__ nodefac, float *__restrict__ rowfac8) {
    const int n = blockIdx.x, s = node_s[n];
    const unsigned bits = keep[n];
    auto m = [&](int k) { return ((bits >> k) & 1u) ? scale : 0.f; };
    if (threadIdx.x < 18) nodefac[(size_t)n * 18 + threadIdx.x] = m((int)threadIdx.x);
    const float2 tt = node_scale[n];
    const f4 lo = {tt.x * m(0), tt.x * m(2), tt.y * m(6), m(5)}, hi = {m(9), m(8), m(12), m(11)};
    float *dst = rowfac8 + (size_t)node_row[n] * 8;
    for (int i = threadIdx.x; i < s * s; i += blockDim.x) {
        st4(dst + (size_t)i * 8, lo);
        st4(dst + (size_t)i * 8 + 4, hi);
    }
}
// X [rows][4 C] *= the node's factors of slices (k0, k1, k2, k3), block by block; row_node == nullptr: row r belongs to node r
__global__ void scale_node_blocks(float *__restrict__ X, const int *__restrict__ row_node, const float *__restrict__ nodefac, int k0, int k1,
                                  int k2, int k3, int C, long long rows) {
    const int nq = C / 4;   // float4 per block
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 4 * nq) return;
    const long long row = i / (4 * nq);
    const int blk = (int)(i % (4 * nq)) / nq;
    const int node = row_node ? row_node[row] : (int)row;
    const int k = blk == 0 ? k0 : blk == 1 ? k1 : blk == 2 ? k2 : k3;
    const float f = nodefac[(size_t)node * 18 + k];
    f4 v = ld4(X + i * 4);
    st4(X + i * 4, v * f);
}

// stacked[p] = K^(kperm[p])  (gather; the gradients take the inverse permutation in smp_fold_level).  Block k of the level weight is
// K[(k C + ci) C + co] in the SMP_omega layout [18C][C] and K[co 18C + k C + ci] in the CustomMatMulTensor layout [C][18C]
// (custom != 0, SMP_2D_ver8): the stacked copy is [ci][co] either way, so the block GEMMs do not care.
__device__ __forceinline__ size_t weight_index(int k, int r, int C, int custom) {
    const int ci = r / C, co = r % C;
    return custom ? (size_t)co * 18 * C + (size_t)k * C + ci : ((size_t)k * C + ci) * C + co;
}
// every level's stacked copy in one launch (the parameters are fixed for the whole forward pass)
constexpr int kStackLevels = 8;
struct StackAll {
    const float *K[kStackLevels];
    float *stacked[kStackLevels];
    int n;
};
__global__ void stack_weights_all(StackAll a, int C, int custom) {
    const int CC = C * C;
    const float *K = a.K[blockIdx.y];
    float *st = a.stacked[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 18 * CC; i += gridDim.x * blockDim.x)
        st[i] = K[weight_index(c_kperm[i / CC], i % CC, C, custom)];
}

// One pass over the per-(node,x) partials of combine-backward for a block of nodes [n0, n1):
//   dSout[n] = sum_x dSpart[(n,x)]
//   colpart[block] = sum over the block's pairs of dbpart    (folded in order by smp_fold_level)
// 256 threads = row groups x C/4 float4 lanes (C % 4 == 0, C <= 1024); the groups are folded through LDS in a fixed order.
#define GF_REDP_PARAMS                                                                                                              \
    const float *__restrict__ dSpart, const float *__restrict__ dbpart, float *__restrict__ dSout, float *__restrict__ colpart,          \
        const int *__restrict__ node_s, const long long *__restrict__ node_pair, int Cr, int nodes, int nodes_per_block
#define GF_REDP_ARGS dSpart, dbpart, dSout, colpart, node_s, node_pair, Cr, nodes, nodes_per_block
__device__ __forceinline__ void reduce_pairs_body(GF_REDP_PARAMS, int bid, float *red) {
    const int C = Cr;
    const int n0 = bid * nodes_per_block, n1 = (n0 + nodes_per_block < nodes) ? n0 + nodes_per_block : nodes;
    const int nl = C / 4, ng = 256 / nl;
    const int g = threadIdx.x / nl, fl = threadIdx.x % nl;
    const auto one = [](int) { return 1.f; };
    if (g < ng) {
        for (int n = n0 + g; n < n1; n += ng)
            st4(dSout + (size_t)n * C + 4 * fl, batched_sum(dSpart + (size_t)node_pair[n] * C + 4 * fl, (size_t)C, 0, node_s[n], one));
        const long long p0 = node_pair[n0], p1 = (n1 < nodes) ? node_pair[n1] : node_pair[n1 - 1] + node_s[n1 - 1];
        const int cnt = (int)((p1 - p0 - g + ng - 1) / ng);
        st4(red + g * C + 4 * fl, batched_sum(dbpart + ((size_t)p0 + g) * C + 4 * fl, (size_t)ng * C, 0, cnt > 0 ? cnt : 0, one));
    }
    __syncthreads();
    if (g == 0) {
        f4 t = ld4(red + 4 * fl);
        for (int k = 1; k < ng; ++k) t += ld4(red + k * C + 4 * fl);
        st4(colpart + (size_t)bid * C + 4 * fl, t);
    }
}
__global__ __launch_bounds__(256) void smp_reduce_pairs(GF_REDP_PARAMS) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    reduce_pairs_body(GF_REDP_ARGS, (int)blockIdx.x, red);
}
// Both in ONE launch (round 6): they are independent (each reads what combine-backward left), small and latency-bound -- the column
// partials' workgroups first, then a workgroup per source node of the level below.  GF_SMP_FUSE_SMALL=0: two launches.
__global__ __launch_bounds__(256) void smp_reduce_pairs_and_diag_gather(GF_REDP_PARAMS, int nb, GF_DIAGB_PARAMS) {
    __shared__ __attribute__((aligned(16))) float red[1024];
    if ((int)blockIdx.x < nb) reduce_pairs_body(GF_REDP_ARGS, (int)blockIdx.x, red);
    else diag_gather_bwd_body(GF_DIAGB_ARGS, (int)blockIdx.x - nb);
}

// End of a level's reverse sweep: every partial image of its weight and bias gradients folded in ONE launch, in a fixed
// order, straight into the caller's gradient buffers:
//   stacked positions [0,8)  <- the row-range images of smp_wgrad_c64          (n = 8 C^2)
//   [8,9), [9,10)            <- dK15, dK16 on the compact rows                  (n = C^2 each)
//   [10,14), [14,18)         <- the per-(node,x) and per-node products          (n = 4 C^2 each)
//   bias                     <- the column partials of smp_reduce_pairs         (n = C)
// dK_l[weight_index(kperm[p], .)] += sum (the inverse of stack_weights_all's permutation), db_l += sum.
// 256 threads = 64 outputs x 4 split quarters: a quarter sums its run of images in order (8 loads in flight), the four
// quarters are then added in order through LDS -- the summation tree depends on the image count only.
constexpr int kFoldGroups = 6;
struct FoldArgs {
    const float *part[kFoldGroups];
    int splits[kFoldGroups];
    unsigned n[kFoldGroups], first[kFoldGroups];  // outputs [first, first + n) of the stacked index space (bias at 18 C^2)
    int ngroups;
};
__global__ __launch_bounds__(256) void smp_fold_level(FoldArgs a, float *__restrict__ dK, float *__restrict__ db, int C, int custom,
                                                      unsigned total) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const unsigned i = blockIdx.x * 64u + e;
    float acc = 0.f;
    if (i < total) {
        int g = 0;
#pragma unroll
        for (int k = 1; k < kFoldGroups; ++k)
            if (k < a.ngroups && i >= a.first[k]) g = k;
        const unsigned j = i - a.first[g];
        if (j < a.n[g]) {
            const int S = a.splits[g], per = (S + 3) / 4;
            const int s0 = qtr * per, s1 = (s0 + per < S) ? s0 + per : S;
            const float *p = a.part[g] + j;
            const size_t stride = a.n[g];
            for (int sb = s0; sb < s1; sb += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (sb + k < s1) ? p[(size_t)(sb + k) * stride] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            }
        }
    }
    red[qtr][e] = acc;
    __syncthreads();
    if (qtr == 0 && i < total) {
        const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const unsigned CC = (unsigned)C * C;
        if (i < 18 * CC)
            dK[weight_index(c_kperm[i / CC], (int)(i % CC), C, custom)] += v;
        else
            db[i - 18 * CC] += v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// combine-forward.  Workgroup per (node, four consecutive x) -- the quads of tables-forward: f_l[x, y, :] for all y.
// A workgroup lives for little more than its memory latencies (one barrier between the gather of U = Z + Z'^T + compact
// terms and the A-products), so four x per workgroup put four times the loads in flight per latency and read the
// adjacency once instead of four times.  Items = (x, e) / (x, y) pairs dealt to the sixteen row groups.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCombX = 4;
constexpr int kFusedMaxN = 32;  // receptive-field cap of the fused levels (smp_fused_supported)
struct QuadWhere {
    int N, x0, cnt, node, win;
    size_t rowbase, pairbase;
};
__device__ __forceinline__ QuadWhere locate_quad(const int *quad_node, const int *quad_b0, const int *node_s,
                                                 const long long *node_row, const long long *node_pair, int nwin) {
    QuadWhere w;
    const int q = (int)(blockIdx.x / nwin);
    w.win = (int)(blockIdx.x % nwin);
    w.node = quad_node[q];
    w.x0 = quad_b0[q];
    w.N = node_s[w.node];
    w.cnt = (w.N - w.x0 < kCombX) ? w.N - w.x0 : kCombX;
    w.rowbase = (size_t)node_row[w.node];
    w.pairbase = (size_t)node_pair[w.node];
    return w;
}

template <int LPC, int MAXN = kFusedMaxN>   // MAXN: largest receptive field of the launch (64: the nodes above 32 positions, round 6)
__global__ __launch_bounds__(kThreads) void smp_combine_fwd(const float *__restrict__ O, const float *__restrict__ A,
                                                            const float *__restrict__ Vout, const float *__restrict__ Sout,
                                                            const float *__restrict__ bias, float *__restrict__ F,
                                                            const int *__restrict__ quad_node, const int *__restrict__ quad_b0,
                                                            const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                                            const long long *__restrict__ node_pair, int C, int nwin,
                                                            const float *__restrict__ Gc, const long long *__restrict__ pair_src_pair,
                                                            const short *__restrict__ pi, const float *__restrict__ rsum,
                                                            int ocols,  // 3: O = [O_loc | Z | Z']; 2: O = [O_loc | U], U = Z + Z'^T
                                                            const float *__restrict__ nodefac = nullptr) {  // (or null) slice dropout: [nodes][18]
    // factors; the compact products G15 / G16 take theirs here (as smp_combine_fwd_panels)
    constexpr int CW = 4 * LPC;
    const size_t ldo = (size_t)ocols * C;
    constexpr int NGRP = kThreads / LPC;
    const int tid = threadIdx.x;
    const int grp = tid / LPC, fl = tid % LPC;
    const QuadWhere W = locate_quad(quad_node, quad_b0, node_s, node_row, node_pair, nwin);
    const int N = W.N, items = W.cnt * N;
    const size_t rowbase = W.rowbase, pairbase = W.pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;
    const int fc = fok ? f : 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency_lite<false>(smem, A + rowbase, rsum + pairbase, N);  // (published by the barrier below)
    float *sU = smem + adj_lds_floats(N);  // [kCombX][N][CW]  Z[x,e] + Z'[e,x] (+ compact terms)
    // Phase-2 item `it` = (xi, y) reads the O_loc block of the row whose U block phase-1 item `it` = (xi, e = y) reads:
    // both are fetched here, so a workgroup pays one HBM round trip instead of one either side of the barrier.
    constexpr int MAXIT = (kCombX * MAXN + NGRP - 1) / NGRP;  // fused levels: N <= 32 (64 for the launch over the big nodes, smp_fused_supported)
    f4 oloc[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int it = grp + k * NGRP;
        if (it >= items) break;
        const int xi = it / N, e = it - xi * N, x = W.x0 + xi;
        const float *orow = O + (rowbase + (size_t)x * N + e) * ldo + fc;
        f4 u = ld4(orow + O_Z * C);
        oloc[k] = ld4(orow + O_LOC * C);
        if (ocols == 3) u += ld4(O + (rowbase + (size_t)e * N + x) * ldo + O_ZP * C + fc);
        {  // + D_bb[x,e] K15 + D_ac[e,x] K16, gathered from the compact products of the level below
            const int pxe = pi[rowbase + (size_t)x * N + e], pex = pi[rowbase + (size_t)e * N + x];
            const f4 g15 = ld4(Gc + (size_t)(pair_src_pair[pairbase + x] + (pxe >= 0 ? pxe : 0)) * 2 * C + fc);
            const f4 g16 = ld4(Gc + (size_t)(pair_src_pair[pairbase + e] + (pex >= 0 ? pex : 0)) * 2 * C + C + fc);
            const float c15 = nodefac ? nodefac[(size_t)W.node * 18 + 15] : 1.f, c16 = nodefac ? nodefac[(size_t)W.node * 18 + 16] : 1.f;
            if (pxe >= 0) u += c15 * g15;
            if (pex >= 0) u += c16 * g16;
        }
        st4(sU + (size_t)it * CW + 4 * fl, fok ? u : splat(0.f));
    }
    float *sV = sU + (size_t)kCombX * N * CW;  // [kCombX][CW]  Vout rows of the quad's x
    if (grp < W.cnt) st4(sV + grp * CW + 4 * fl, ld4(Vout + (pairbase + W.x0 + grp) * (size_t)C + fc));
    const f4 sout = ld4(Sout + (size_t)W.node * C + fc);
    const f4 bb = ld4(bias + fc);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int it = grp + k * NGRP;
        if (it >= items) break;
        const int xi = it / N, y = it - xi * N, x = W.x0 + xi;
        const float *const Tt[1] = {sU + (size_t)xi * N * CW};
        f4 m[1];
        small_matvec<1, CW>(L, N, y, fl, Tt, m);
        const f4 vout = ld4(sV + xi * CW + 4 * fl);
        const f4 z = bb + oloc[k] + m[0] + L.r[y] * vout + L.at(x, y, N) * sout;
        if (fok) {
            f4 out;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = lreluf(z[j]);
            st4(F + (rowbase + (size_t)x * N + y) * (size_t)C + f, out);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// combine-backward.  Workgroup per (node, four consecutive x): from dF[x,:,:] produce dO[(x,y)] (the O_loc block), dZ[(x,e)],
// dZ'[(e,x)] and the per-(node,x) partials dVout, dS-part, db-part.
// ---------------------------------------------------------------------------------------------------------------
template <int LPC, int UB = 4>   // UB: rows whose loads a thread issues together, ahead of their stores (measured at cfg3: 2 -> 0.72, 4 -> 0.70, 8 -> 0.81 ms)
__global__ __launch_bounds__(kThreads) void smp_combine_bwd(const float *__restrict__ F, const float *__restrict__ dF,
                                                            const float *__restrict__ node_dF,  // [nodes][C] or null: (dF null) dF is
                                                            // the same C-vector at every (x,y) of a node (readout broadcast)
                                                            const float *__restrict__ A, float *__restrict__ dO,
                                                            float *__restrict__ dVout, float *__restrict__ dSpart,
                                                            float *__restrict__ dbpart, const int *__restrict__ quad_node,
                                                            const int *__restrict__ quad_b0, const int *__restrict__ node_s,
                                                            const long long *__restrict__ node_row,
                                                            const long long *__restrict__ node_pair, int C, int nwin,
                                                            const float *__restrict__ rsum, int ocols,
                                                            float *__restrict__ dzmax) {  // or null: [workgroups][CW] largest |dz| per column
    // of this workgroup's rows (C = 64: the weight gradients' column exponents, smp_wgrad_column_bounds)
    constexpr int CW = 4 * LPC;
    constexpr int NGRP = kThreads / LPC;
    const int tid = threadIdx.x;
    const int grp = tid / LPC, fl = tid % LPC;
    const QuadWhere W = locate_quad(quad_node, quad_b0, node_s, node_row, node_pair, nwin);
    const int N = W.N, items = W.cnt * N;
    const size_t rowbase = W.rowbase, pairbase = W.pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;
    const int fc = fok ? f : 0;
    const size_t ldo = (size_t)ocols * C;
    f4 dzm = splat(0.f);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency_lite<true>(smem, A + rowbase, rsum + pairbase, N);  // L.A[e][y] = A+[y][e]; see the barrier below
    float *sDz = smem + adj_lds_floats(N);  // [kCombX][N][CW]
    const f4 gnode = node_dF ? ld4(node_dF + (size_t)W.node * C + fc) : splat(0.f);
    for (int it0 = grp; it0 < items; it0 += UB * NGRP) {
        f4 fv[UB], g[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int it = it0 + u * NGRP;
            if (it < items) {
                const int xi = it / N, y = it - xi * N;
                const size_t row = rowbase + (size_t)(W.x0 + xi) * N + y;
                fv[u] = ld4(F + row * C + fc);
                g[u] = !node_dF ? ld4(dF + row * C + fc) : dF ? gnode + ld4(dF + row * C + fc) : gnode;   // (a tower's level below the top: both)
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int it = it0 + u * NGRP;
            if (it < items) {
                const int xi = it / N, y = it - xi * N;
                const size_t row = rowbase + (size_t)(W.x0 + xi) * N + y;
                f4 dz;
#pragma unroll
                for (int j = 0; j < 4; ++j) dz[j] = fok ? g[u][j] * (fv[u][j] > 0.f ? 1.f : kAlphaF) : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) dzm[j] = fmaxf(dzm[j], fabsf(dz[j]));
                st4(sDz + (size_t)it * CW + 4 * fl, dz);
                if (fok) st4(dO + row * ldo + O_LOC * C + f, dz);
            }
        }
    }
    __syncthreads();
    for (int it = grp; it < items; it += NGRP) {
        const int xi = it / N, e = it - xi * N, x = W.x0 + xi;
        const float *const Tt[1] = {sDz + (size_t)xi * N * CW};
        f4 m[1];
        small_matvec<1, CW>(L, N, e, fl, Tt, m);  // dU[e] = sum_y A+[y][e] dz[y]
        if (fok) {
            st4(dO + (rowbase + (size_t)x * N + e) * ldo + O_Z * C + f, m[0]);
            if (ocols == 3) st4(dO + (rowbase + (size_t)e * N + x) * ldo + O_ZP * C + f, m[0]);
        }
    }
    if (fok) {
        for (int it = grp; it < W.cnt * 3; it += NGRP) {  // (x, which) : r[y] | A+[x][y] | 1 weighted sums over y
            const int xi = it / 3, k = it - xi * 3, x = W.x0 + xi;
            const float *dz = sDz + (size_t)xi * N * CW + 4 * fl;
            f4 acc = splat(0.f);
            for (int y = 0; y < N; ++y) {
                const float w = (k == 0) ? L.r[y] : (k == 1) ? L.A[y * (N + 1) + x] : 1.f;
                acc += w * ld4(dz + y * CW);
            }
            float *dst = (k == 0) ? dVout : (k == 1) ? dSpart : dbpart;
            st4(dst + (pairbase + x) * (size_t)C + f, acc);
        }
    }
    if (dzmax) {  // (uniform) the sixteen row groups' maxima through the image of dz, which nobody reads any more
        __syncthreads();
        st4(sDz + (size_t)grp * CW + 4 * fl, dzm);
        __syncthreads();
        if (grp == 0) {
            f4 m = dzm;
            for (int g = 1; g < NGRP; ++g) {
                const f4 v = ld4(sDz + (size_t)g * CW + 4 * fl);
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
            st4(dzmax + (size_t)blockIdx.x * CW + 4 * fl, m);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// tables-backward.  Workgroup per (node, b): dP[:, b, :, :] from the table gradients.  Streaming phase as r18_bwd_slab.
//   dP[a,b,c] = X[a] + Y[c] + G5[a] r[c] + G9[c] r[a] + [c==b] Z1[a] + [c==a] Z2[a]
//   X[a]  = dS_ab[a,b] + d rowsum[a] + d colsum[b] + d total + [a==b] d s14
//   Y[c]  = dS_bc[b,c]     G5[a] = dT6[a,b]     G9[c] = dT10[b,c]
//   Z1[a] = dDbb[a,b] + dD8[a] + d s15 + [a==b] d s18          Z2[a] = dDac[a,b] + dD11[b]
// ---------------------------------------------------------------------------------------------------------------
template <int LPC, int NI>
__global__ __launch_bounds__(kThreads, 3) void smp_tables_bwd(const float *__restrict__ dT, const float *__restrict__ dVt,
                                                              const float *__restrict__ dSt, const float *__restrict__ A,
                                                              float *__restrict__ dP, const short *__restrict__ pi, Ragged R,
                                                              int C, int nwin) {
    constexpr int PPW = 64 / LPC;
    constexpr int CW = 4 * LPC;
    constexpr int NGRP = kThreads / LPC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = lane / LPC, fl = lane % LPC;
    const int grp = tid / LPC;
    const Where W = locate(R, nwin);
    const int N = W.N, b = W.i;
    const size_t rowbase = W.rowbase, pbase = W.pbase, pairbase = W.pairbase;
    const int f = W.win * CW + 4 * fl;
    const bool fok = f < C;
    const int fc = fok ? f : 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AdjLds L = load_adjacency<false>(smem, A + rowbase, N);
    float *sX = smem + adj_lds_floats(N);
    float *sG5 = sX + N * CW;
    float *sZ1 = sG5 + N * CW;
    float *sZ2 = sZ1 + N * CW;
    // Selection maps of the node: dP[a][b][c] is only ever read by the consumer gather where pi_a(b) and pi_a(c) exist
    // (the promoted tensor is zero elsewhere, about two thirds of it at QM9 sizes), so only those positions are written.
    short *sPi = reinterpret_cast<short *>(sZ2 + N * CW);
    for (int i = tid; i < N * N; i += kThreads) sPi[i] = pi[rowbase + i];
    {
        const float *ds = dSt + (size_t)W.node * 4 * C + fc;
        const f4 dtotal = ld4(ds + 0 * C), ds14 = ld4(ds + 1 * C), ds15 = ld4(ds + 2 * C), ds18 = ld4(ds + 3 * C);
        const float *dvb = dVt + (pairbase + b) * 4 * (size_t)C + fc;
        const f4 dcol = ld4(dvb + 1 * C), dd11 = ld4(dvb + 3 * C);
        for (int a = grp; a < N; a += NGRP) {
            const float *t = dT + (rowbase + (size_t)a * N + b) * (size_t)(T_COLS * C) + fc;  // table row (a, b)
            const float *dva = dVt + (pairbase + a) * 4 * (size_t)C + fc;
            f4 x = ld4(t + T_SAB * C) + ld4(dva + 0 * C) + dcol + dtotal;
            // (the D_bb / D_ac table gradients reach df_{l-1} through the compact path: dFdc in the consumer gather)
            f4 z1 = ld4(dva + 2 * C) + ds15;
            if (a == b) {
                x += ds14;
                z1 += ds18;
            }
            const f4 m = fok ? splat(1.f) : splat(0.f);
            st4(sX + a * CW + 4 * fl, x * m);
            st4(sG5 + a * CW + 4 * fl, ld4(t + T_T6 * C) * m);
            st4(sZ1 + a * CW + 4 * fl, z1 * m);
            st4(sZ2 + a * CW + 4 * fl, dd11 * m);
        }
    }
    __syncthreads();

    f4 yv[NI], g9[NI];
    float rc[NI];
    int coff[NI], cxs[NI];
    bool live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * PPW + cg;
        const bool ok = c < N;
        const int cx = ok ? c : 0;
        cxs[i] = cx;
        live[i] = ok && fok;
        coff[i] = cx * C + fc;
        rc[i] = L.r[cx];
        const float *t = dT + (rowbase + (size_t)b * N + cx) * (size_t)(T_COLS * C) + fc;  // table row (b, c)
        yv[i] = ld4(t + T_SBC * C);
        g9[i] = ld4(t + T_T10 * C);
    }
    const int ib = b / PPW, cgb = b % PPW;
    float *dPg = dP + pbase * C + (size_t)b * N * C;
    const size_t rowStride = (size_t)N * N * C;
    for (int a = wave; a < N; a += kWaves) {
        const short *map = sPi + a * N;
        if (map[b] < 0) continue;  // wave-uniform: the whole row (a, b, :) of the promoted tensor is structurally zero
        float *row = dPg + a * rowStride;
        const f4 xa = ld4(sX + a * CW + 4 * fl), g5a = ld4(sG5 + a * CW + 4 * fl);
        const float ra = L.r[a];
        const int ia = a / PPW, cga = a % PPW;
        const f4 z1 = ld4(sZ1 + a * CW + 4 * fl) * ((cg == cgb) ? 1.f : 0.f);
        const f4 z2 = ld4(sZ2 + a * CW + 4 * fl) * ((cg == cga) ? 1.f : 0.f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f4 o = xa + yv[i] + g5a * rc[i] + g9[i] * ra;
            if (i == ib) o += z1;
            if (i == ia) o += z2;
            if (live[i] && map[cxs[i]] >= 0) st4(row + coff[i], o);
        }
    }
}

template <int LPC>
size_t tables_bwd_lds(int N) {
    constexpr int CW = 4 * LPC;
    return sizeof(float) * ((size_t)adj_lds_floats(N) + 4 * (size_t)N * CW) + sizeof(short) * (size_t)N * N + 16;
}
template <int LPC>
size_t combine_lds(int N) {
    constexpr int CW = 4 * LPC;
    return sizeof(float) * ((size_t)adj_lds_floats(N) + (size_t)kCombX * (N + 1) * CW);  // + the forward's Vout rows
}

// threads of a workgroup-per-node kernel whose items are (position, channel quad): the level's largest node decides (a level of
// 3-vertex fields left 200 of 256 threads idle, and a quarter of the workgroups a CU could hold)
static int node_block(const gfsmp::LevelLayout &h, int C) {
    const int smax = h.buckets.empty() ? 1 : h.buckets.back().s;
    int t = (smax * (C / 4) + 63) / 64 * 64;
    return t < 64 ? 64 : t > 256 ? 256 : t;
}

struct SizeClass {
    long long lo, hi;  // pair range
    int smax, ni;
};

// merge the level's size buckets into the slab kernels' NI classes (LPC = 16 only: C % 64 == 0 is the fused path's domain)
std::vector<SizeClass> classes_of(const gfsmp::LevelLayout &h, int ppw) {
    std::vector<SizeClass> out;
    size_t k = 0;
    for (int cls = 1; cls <= 8 && k < h.buckets.size(); cls *= 2) {
        const size_t k0 = k;
        int smax = 0;
        // (never beyond 32 positions, whatever the lanes per position: the nodes above take smp_tables_fwd_big -- big_part)
        while (k < h.buckets.size() && h.buckets[k].s <= cls * ppw && h.buckets[k].s <= 32) smax = h.buckets[k++].s;
        if (k == k0) continue;
        SizeClass c;
        c.lo = h.node_pair[h.buckets[k0].first_node];
        c.hi = (k < h.buckets.size()) ? h.node_pair[h.buckets[k].first_node] : (long long)h.pairs;
        c.smax = smax;
        c.ni = cls;
        out.push_back(c);
    }
    return out;
}

// The nodes of a level above 32 positions (nodes are numbered by size: they come last), see kFusedMaxField: first node / row / pair /
// workgroup-kernel quad of that range and its extents.  nodes == 0: none.
struct BigPart {
    int n0 = 0, nodes = 0, quad0 = 0, quads = 0, smax = 0;
    long long row0 = 0, pair0 = 0, pairs = 0;
};
static BigPart big_part(const gfsmp::LevelLayout &h) {
    BigPart b;
    b.n0 = h.nNodes, b.row0 = h.rows, b.pair0 = h.pairs, b.quad0 = (int)h.quad_node.size();
    long long q = 0;
    for (const gfsmp::Bucket &bk : h.buckets) {
        if (bk.s > 32) {
            b.n0 = bk.first_node;
            break;
        }
        q += (long long)bk.count * ((bk.s + 3) / 4);   // (quads are in node order, ceil(s / 4) per node: smp_prep.cpp)
    }
    if (b.n0 < h.nNodes) {
        b.nodes = h.nNodes - b.n0;
        b.row0 = h.node_row[(size_t)b.n0], b.pair0 = h.node_pair[(size_t)b.n0];
        b.pairs = h.pairs - b.pair0;
        b.quad0 = (int)q, b.quads = (int)h.quad_node.size() - (int)q;
        b.smax = h.buckets.back().s;
    }
    return b;
}

Ragged ragged_for(const gf_smp::DevLevel &d, long long lo, int smax) {
    Ragged R = {d.pair_node, d.node_s, d.node_p, d.node_row, d.node_pair, lo, smax};
    return R;
}

// channel counts that are whole 32-channel windows but not whole 64-channel ones (C = 32, 96): the workgroup-per-(node, x) kernels run
// with eight lanes per row instead of sixteen half-idle ones
static bool smp_half_window(int C) { return C % 32 == 0 && C % 64 != 0; }
// ... and whole 16-channel windows only (C = 16, 48, ...): four lanes per position in tables-forward (round 5: the 16-channel kernel family)
static bool smp_quarter_window(int C) { return C == 16; }

// tables-forward keeps smp_vectors' sums itself (C % 64 == 0; GF_SMP_TF_VEC=0: the separate pass)
static bool smp_tables_fold_vectors(const gf_smp *s) {
    return ((s->cfg.nChanels & 63) == 0 || smp_half_window(s->cfg.nChanels) || smp_quarter_window(s->cfg.nChanels)) && !env_is("GF_SMP_TF_VEC", '0');
}

// the eight-lanes-per-position classes (C % 32 == 0, C % 64 != 0): a wave load covers eight positions, NI = 1, 2, 4 for s <= 8, 16, 32
// launch names of tables-forward: one per template instantiation, as rocprof lists kernels by symbol (the bench's "dominant kernel" is the
// symbol with the largest time per step; it sums these four for the kernel's own roofline line)
template <int NI>
constexpr const char *tables_fwd_name() {
    return NI == 1 ? "smpf_tables_fwd_ni1" : NI == 2 ? "smpf_tables_fwd_ni2" : NI == 4 ? "smpf_tables_fwd_ni4" : "smpf_tables_fwd_ni8";
}
template <int NI, int LPC>   // LPC = 8 (32-channel windows) or 4 (16-channel windows): 64 / LPC positions per wave load
gf_status launch_tables_fwd_wn(gf_smp *s, int l, const SizeClass &c) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int C = s->cfg.nChanels, nwin = C / (4 * LPC);
    constexpr int PPW = 64 / LPC;
    const int n_lo = h.pair_node[(size_t)c.lo], n_hi = (c.hi < (long long)h.pairs) ? h.pair_node[(size_t)c.hi] : h.nNodes;
    if (n_hi <= n_lo) return GF_OK;
    const size_t lds = sizeof(float) * ((c.smax + 3) & ~3) + 16 * (size_t)c.smax + sizeof(int) * (size_t)c.smax * PPW * NI + 16 +
                       (size_t)c.smax * c.smax + 16;
    const int nwv = kThreads / 64;
    const size_t lds_v = ((lds + 15) & ~(size_t)15) + 16 + (size_t)nwv * ((size_t)c.smax * (32 * LPC) + 32 * LPC);
    gf_status st = opt_in_lds(ctx, smp_tables_fwd_w<NI, true, true, LPC>, lds_v);
    if (st != GF_OK) return st;
    const int flags = d.t_zeros ? 1 : 0;
    GF_LAUNCH(ctx, tables_fwd_name<NI>(), (smp_tables_fwd_w<NI, true, true, LPC>), dim3((unsigned)((n_hi - n_lo) * nwin)), dim3(kThreads), lds_v,
              s->lv[l - 1].f, d.rsum, d.Q, d.Vt, d.scal, d.pair_src_row, d.pair_src_s, d.pi, d.tf_recs + 2 * (size_t)n_lo, C, nwin,
              flags, flags ? d.rowflag : (const unsigned char *)nullptr, d.St);
    return GF_OK;
}

template <int NI>
gf_status launch_tables_fwd_w(gf_smp *s, int l, const SizeClass &c) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const int C = s->cfg.nChanels, nwin = (C + 63) / 64;
    // nodes of the class: nodes are sorted by size, so the class is a contiguous node range -- and the same range of positions in
    // the level's (class, molecule) order, which is the order of the records (build_tf_records)
    const int n_lo = h.pair_node[(size_t)c.lo], n_hi = (c.hi < (long long)h.pairs) ? h.pair_node[(size_t)c.hi] : h.nNodes;
    if (n_hi <= n_lo) return GF_OK;
    const size_t lds = sizeof(float) * ((c.smax + 3) & ~3) + 16 * (size_t)c.smax + sizeof(int) * (size_t)c.smax * 4 * NI + 16 +
                       (size_t)c.smax * c.smax + 16;
    const int flags = ((d.t_zeros && (C & 63) == 0) ? 1 : 0);
    if (smp_tables_fold_vectors(s)) {   // the sums over b of smp_vectors kept by the kernel itself (VEC)
        const int nwv = (NI >= GF_TF_WIDE ? 2 : 1) * kThreads / 64;
        const size_t lds_v = ((lds + 15) & ~(size_t)15) + 16 + (size_t)nwv * ((size_t)c.smax * 512 + 512);
        gf_status st = opt_in_lds(ctx, smp_tables_fwd_w<NI, true, true>, lds_v);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, tables_fwd_name<NI>(), (smp_tables_fwd_w<NI, true, true>), dim3((unsigned)((n_hi - n_lo) * nwin)), dim3((NI >= GF_TF_WIDE ? 2 : 1) * kThreads), lds_v,
                  s->lv[l - 1].f, d.rsum, d.Q, d.Vt, d.scal, d.pair_src_row, d.pair_src_s, d.pi, d.tf_recs + 2 * (size_t)n_lo, C, nwin,
                  flags, flags ? d.rowflag : (const unsigned char *)nullptr, d.St);
    } else if ((C & 63) == 0)
        GF_LAUNCH(ctx, tables_fwd_name<NI>(), (smp_tables_fwd_w<NI, true>), dim3((unsigned)((n_hi - n_lo) * nwin)), dim3((NI >= GF_TF_WIDE ? 2 : 1) * kThreads), lds,
                  s->lv[l - 1].f, d.rsum, d.Q, d.Vt, d.scal, d.pair_src_row, d.pair_src_s, d.pi, d.tf_recs + 2 * (size_t)n_lo, C, nwin,
                  flags, flags ? d.rowflag : (const unsigned char *)nullptr, (float *)nullptr);
    else
        GF_LAUNCH(ctx, tables_fwd_name<NI>(), (smp_tables_fwd_w<NI, false>), dim3((unsigned)((n_hi - n_lo) * nwin)), dim3((NI >= GF_TF_WIDE ? 2 : 1) * kThreads), lds,
                  s->lv[l - 1].f, d.rsum, d.Q, d.Vt, d.scal, d.pair_src_row, d.pair_src_s, d.pi, d.tf_recs + 2 * (size_t)n_lo, C, nwin,
                  flags, (const unsigned char *)nullptr, (float *)nullptr);
    return GF_OK;
}

template <int NI>
gf_status launch_tables_bwd(gf_smp *s, int l, const SizeClass &c, const float *dT) {
    gf_ctx *ctx = s->ctx;
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels, nwin = (C + 63) / 64;
    const size_t lds = tables_bwd_lds<16>(c.smax);
    gf_status st = opt_in_lds(ctx, smp_tables_bwd<16, NI>, lds);
    if (st != GF_OK) return st;
    GF_LAUNCH(ctx, "smpf_tables_bwd", (smp_tables_bwd<16, NI>), dim3((unsigned)((c.hi - c.lo) * nwin)), dim3(kThreads), lds, dT,
              d.dVt, d.dSt, d.adj, s->P, d.pi, ragged_for(d, c.lo, c.smax), C, nwin);
    return GF_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// consumer gather with tables-backward folded in.  Per SOURCE node w of level l-1:
//   df_{l-1}[w][p,q] = [p==q] dFd[p] + [q==c_w] dFc[p] + sum over consumers (n,a), in consumer order, of dP_n[a, b, c]
// with b = inv(p), c = inv(q) both present, and dP_n[a,b,c] EVALUATED from the table gradients by the formula above
// smp_tables_bwd (same expression; only the two diagonal terms are summed separately) instead of being written by one
// kernel and read back by the next.  A lane owns (p, 4 channels) and keeps the accumulators of its row in registers.
// The [c == b] and [c == a] terms of dP land on fixed positions of the row: c == b means q == p (inv is injective), and c == a means
// q == c_w (the source's own vertex sits at position a of the consumer): they are summed over the consumers in two extra
// accumulators, which start from the compact-path gradients of the same two positions, and join the row at the end.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGatherMaxS = 64;   // (32 until round 6)

// ---------------------------------------------------------------------------------------------------------------
// Nothing but memory requests and sums inside the consumer loop (round 3).  Round 2's kernel (a workgroup per source, consumer lists
// staged in LDS behind barriers, one launch per size class) spent three dependent round trips per consumer at three waves per SIMD: it
// was latency-bound (rocprof: the loop's instructions accounted for a tenth of its time), not issue-bound.  Here:
//   * everything wave-uniform about a consumer comes from two per-batch tables built on the device at prepare time
//     (build_gather_records): an 8-dword header per consumer entry (size, index, row / pair bases, node, r[a]) and a 4-dword record
//     per (entry, source position q): byte offset of the consumer's row (b = 0, c = inv(q)), r[c], presence m, m r[a].  They are
//     read with SCALAR loads, the header one consumer ahead: no LDS, no barrier, no readfirstlane, no per-lane address arithmetic
//     per position -- a position's two requests are buffer loads with a per-lane row offset (b s ldt) and a scalar column offset;
//   * the lane's own index b = inv(p) is requested one consumer ahead;
//   * ALL requests of a consumer (ten row terms + two per position, QB = SW positions at a time for SW <= 16) are issued before
//     the first sum: one round trip per consumer, 26 - 42 KB in flight per wave;
//   * absent positions (uniform) read the consumer's row c = 0 and are multiplied by m = 0; absent b (per lane) sends every
//     request of the lane out of range (zeros): no branch anywhere in the loop, so the memory queue is counted, not drained.
// Held against the two-kernel form (tables-backward writes dP, promote_backward gathers it: GF_SMP_BWD_GATHER=0) by
// tests/test_smp_gpu.py::test_folded_backward_gather_equals_the_two_kernel_path.
// ---------------------------------------------------------------------------------------------------------------
struct GatherTables {
    const int4 *hdr;   // [entries][2]: {s, a, row lo, row hi} {pair base lo, hi, node, r[a] bits}
    const int4 *qrec;  // gather_pad(s_w) records per entry: {row-c byte offset (0 when absent), r[c] bits, m bits, m r[a] bits}
};

// block per source node w: its consumer entries' headers and records (positions past s_w and absent positions: m = 0, row 0)
__global__ __launch_bounds__(64) void build_gather_records(const int *__restrict__ prev_s, const long long *__restrict__ cons_ptr,
                                     const long long *__restrict__ cons_qbase, const int *__restrict__ cons_s,
                                     const int *__restrict__ cons_a, const long long *__restrict__ cons_row,
                                     const long long *__restrict__ cons_pair, const int *__restrict__ pair_node,
                                     const long long *__restrict__ cons_inv_off, const short *__restrict__ inv,
                                     const float *__restrict__ rsum, int ldt_bytes, int4 *__restrict__ hdr, int4 *__restrict__ qrec) {
    const int w = blockIdx.x, sw = prev_s[w];
    const int pad = gfsmp::gather_pad(sw);
    const long long c0 = cons_ptr[w], c1 = cons_ptr[w + 1], qb = cons_qbase[w];
    for (long long ce = c0 + threadIdx.x; ce < c1; ce += blockDim.x) {
        const long long row = cons_row[ce], pe = cons_pair[ce], pb = pe - cons_a[ce];
        hdr[2 * ce] = make_int4(cons_s[ce], cons_a[ce], (int)(unsigned)(row & 0xffffffffll), (int)(row >> 32));
        hdr[2 * ce + 1] = make_int4((int)(unsigned)(pb & 0xffffffffll), (int)(pb >> 32), pair_node[pe], __float_as_int(rsum[pe]));
    }
    const long long n = (c1 - c0) * pad;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const long long ce = c0 + i / pad;
        const int q = (int)(i % pad);
        const long long pe = cons_pair[ce], pb = pe - cons_a[ce];
        const int c = q < sw ? (int)inv[cons_inv_off[ce] + q] : -1;
        // A position q without an image (or padding) multiplies what it loads by m = 0 -- but it still loads: from the row of the
        // FIRST position that has an image, a row (b, c') every b of this entry shares a source with (the entry's own), i.e. one that
        // is written every step (the backward products do not store the S_bc / T10 gradients of rows no source covers).
        int cfirst = 0;
        for (int k = 0; k < sw; ++k) {
            const int ck = (int)inv[cons_inv_off[ce] + k];
            if (ck >= 0) {
                cfirst = ck;
                break;
            }
        }
        const float ra = rsum[pe], rho = c >= 0 ? rsum[pb + c] : 0.f, m = c >= 0 ? 1.f : 0.f;
        qrec[qb + i] = make_int4((c >= 0 ? c : cfirst) * ldt_bytes, __float_as_int(rho), __float_as_int(m), __float_as_int(m * ra));
    }
}

// Scalar loads written out (s_load_dwordx4 + an explicit wait that "produces" the loaded values, so that nothing reads them before
// it).  Left to the compiler, a uniform global load becomes a scalar load only when it can prove that no store of the kernel
// clobbers it -- which it gives up on behind a scheduling barrier, behind a struct of pointers, and in every path of a switch but
// the first (observed): the fallback is a per-lane load plus a readfirstlane waterfall per use, ten times the instructions.
typedef int si4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ si4 s_load4(const int4 *p) {  // p must be wave-uniform
    si4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ void s_wait2(si4 &a, si4 &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }
template <int N>
__device__ __forceinline__ void s_wait(si4 (&r)[N]) {
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]));
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]));
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]));
    else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]));
    else if constexpr (N == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]), "+s"(r[5]));
    else {
        static_assert(N == 8, "batch sizes of the gather");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(r[4]), "+s"(r[5]), "+s"(r[6]), "+s"(r[7]));
    }
}

__device__ __forceinline__ f4 gbuf_ld4_once(__amdgpu_buffer_rsrc_t r, int voff_bytes) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, 0, (GF_NT_SITES & 128) ? 2 : 0));
}
constexpr int kOor = 0x40000000;  // a lane offset no consumer's tables reach (<= 32 x 32 rows of 16 C bytes): the load returns 0

#define GF_GATHER_PARAMS                                                                                                          \
    const float *__restrict__ dT, const float *__restrict__ dVt, const float *__restrict__ dSt, float *__restrict__ dfprev,            \
        const float *__restrict__ dFdc, const int *__restrict__ prev_s, const long long *__restrict__ prev_row,                      \
        const long long *__restrict__ prev_pair, const int *__restrict__ prev_center, const long long *__restrict__ cons_ptr,        \
        const long long *__restrict__ cons_inv_off, const long long *__restrict__ cons_qbase, const short *__restrict__ inv,         \
        GatherTables G, int C
#define GF_GATHER_ARGS dT, dVt, dSt, dfprev, dFdc, prev_s, prev_row, prev_pair, prev_center, cons_ptr, cons_inv_off, cons_qbase, inv, G, C

// one source node w: SWP = gfsmp::gather_pad(s_w) accumulators and records per consumer; QB positions requested together.
// (The pointers stay individual __restrict__ kernel parameters: handed over in a struct they lose the no-alias guarantee against
//  the store of df, and the uniform table loads are then no longer selected as scalar loads.)
// One wave-sized work item: lanes [64 chunk, 64 chunk + 64) of source w's (row p, channel quad) space.  RS = records per consumer
// entry of this source (SWP, or 32 for the sources of 17 .. 32 positions, which run as two items of sixteen positions each:
// positions [qoff, qoff + 16)).
template <int SWP, int QB, int RS = SWP>
__device__ __forceinline__ void gather_source(GF_GATHER_PARAMS, int w, int chunk, int qoff = 0) {
    static_assert(SWP % QB == 0, "whole batches");
    const int sw = prev_s[w], cw = prev_center[w];
    const int nl = C >> 2, items = sw * nl;
    const long long c0 = cons_ptr[w], c1 = cons_ptr[w + 1];
    const int C4 = C * 4, ldtb = T_COLS * C4;  // bytes of a C-block, of a row of dT
    float *dst = dfprev + prev_row[w] * C;
    const float *dfd = dFdc + (size_t)prev_pair[w] * 2 * C;
    const long long ioff0 = c0 < c1 ? cons_inv_off[c0] : 0;  // (the entries of a source are consecutive in inv: sw shorts each)
    const int4 *qr0 = G.qrec + cons_qbase[w];
    {
        const int it = chunk * 64 + (int)(threadIdx.x & 63);
        const bool live = it < items;
        const int p = live ? it / nl : 0, f4b = 16 * (live ? it % nl : 0);
        f4 acc[SWP], accd = splat(0.f), accc = splat(0.f);
#pragma unroll
        for (int q = 0; q < SWP; ++q) acc[q] = splat(0.f);
        if (live) {
            accd = ld4(dfd + (size_t)p * 2 * C + (f4b >> 2));
            accc = ld4(dfd + (size_t)p * 2 * C + C + (f4b >> 2));
        }
        if (c0 < c1) {
            si4 h0 = s_load4(G.hdr + 2 * c0), h1 = s_load4(G.hdr + 2 * c0 + 1);
            s_wait2(h0, h1);
            int b = inv[ioff0 + p];  // (idle lanes read position 0 and are sent out of range below: no branch around a request)
            b = live ? b : -1;
            for (long long ce = c0; ce < c1; ++ce) {
                // header and lane index of the NEXT consumer (clamped re-read at the end: unconditional)
                const long long cn = ce + 1 < c1 ? ce + 1 : ce;
                si4 n0 = s_load4(G.hdr + 2 * cn), n1 = s_load4(G.hdr + 2 * cn + 1);  // (waited for at the end of this consumer)
                int bn = inv[ioff0 + (cn - c0) * sw + p];
                bn = live ? bn : -1;
                const int s = h0.x, a = h0.y;
                const long long urow = ((long long)h0.w << 32) | (unsigned)h0.z, upb = ((long long)h1.y << 32) | (unsigned)h1.x;
                const int unode = h1.z;
                const __amdgpu_buffer_rsrc_t rT = make_rsrc(dT + (size_t)urow * (T_COLS * C), (size_t)s * s * ldtb);
                const __amdgpu_buffer_rsrc_t rV = make_rsrc(dVt + (size_t)upb * 4 * C, (size_t)s * 4 * C4);
                const __amdgpu_buffer_rsrc_t rS = make_rsrc(dSt + (size_t)unode * 4 * C, (size_t)4 * C4);
                const bool has = b >= 0;
                const int tab = has ? (a * s + b) * ldtb + f4b : kOor;  // row (a, b) of the consumer's tables
                const int va = has ? a * 4 * C4 + f4b : kOor, vb = has ? b * 4 * C4 + f4b : kOor;
                const int vs = has ? f4b : kOor, vd = (has && p == cw) ? f4b : kOor;  // a == b  <=>  p is the source's own vertex
                const int tb = has ? b * s * ldtb + f4b : kOor;                          // row (b, 0)
                // (the S_ab / T6 gradient blocks of row (a, b) are read by this source alone, once: streamed -- GF_NT_SITES 128 -- so that they
                //  do not push the S_bc / T10 blocks, which every source of the consumer re-reads, out of the L2)
                const f4 l0 = gbuf_ld4_once(rT, tab + T_SAB * C4), g5 = gbuf_ld4_once(rT, tab + T_T6 * C4);
                const f4 l1 = buf_ld4(rV, va, 0), l4 = buf_ld4(rV, va + 2 * C4, 0);
                const f4 l2 = buf_ld4(rV, vb + C4, 0), z2 = buf_ld4(rV, vb + 3 * C4, 0);
                const f4 l3 = buf_ld4(rS, vs, 0), l5 = buf_ld4(rS, vs + 2 * C4, 0);
                const f4 l6 = buf_ld4(rS, vd + C4, 0), l7 = buf_ld4(rS, vd + 3 * C4, 0);
                const int4 *qr = qr0 + (ce - c0) * RS + qoff;
                f4 x = splat(0.f);
#pragma unroll
                for (int q0 = 0; q0 < SWP; q0 += QB) {
                    f4 y[QB], g9[QB];
                    si4 rec[QB];
#pragma unroll
                    for (int j = 0; j < QB; ++j) rec[j] = s_load4(qr + q0 + j);
                    s_wait(rec);  // (behind the row terms' requests; also lands the next header)
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        y[j] = buf_ld4(rT, tb + T_SBC * C4, rec[j].x);
                        g9[j] = buf_ld4(rT, tb + T_T10 * C4, rec[j].x);
                    }
                    if (q0 == 0) x = ((l0 + l1) + (l2 + l3)) + l6;
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        const float rho = __int_as_float(rec[j].y), m = __int_as_float(rec[j].z), mra = __int_as_float(rec[j].w);
                        acc[q0 + j] += m * (x + y[j]) + g5 * rho + g9[j] * mra;
                    }
                    // (no sched_barrier between the batches: the intrinsic counts as a memory clobber, after which the uniform
                    //  record loads of the loop are no longer selected as scalar loads but as per-lane loads + waterfall loops)
                }
                accd += (l4 + l5) + l7;
                accc += z2;
                s_wait2(n0, n1);
                h0 = n0;
                h1 = n1;
                b = bn;
            }
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < SWP; ++q)
                if (qoff + q < sw) {
                    f4 o = acc[q];
                    if (qoff + q == p) o += accd;
                    if (qoff + q == cw) o += accc;
                    gf_st_s<64>(reinterpret_cast<f4 *>(dst + ((size_t)p * sw + qoff + q) * C + (f4b >> 2)), o);
                }
        }
    }
}

// ONE launch per level, molecule by molecule (gfsmp::LevelLayout::gather_items): the rows (b, c) of a consumer are re-read by each of its
// ~s sources, of whatever size -- launched per size class, a molecule's table gradients were fetched from HBM once per class (rocprof:
// 1.24 -> 1.53 ms for the same kernel when the classes of the launch order were refined).  Every wave of the grid takes one work item
// of the level's list: (source, 64-lane chunk of its (p, channel quad) rows, eight positions q) -- a source of more than eight positions
// is several items, all sources molecule-major (smp_prep.cpp).  Waves are independent (no LDS, no barrier): a workgroup is just four
// consecutive items, whatever their sources' sizes; nothing is launched for rows a source does not have (idle waves of a
// workgroup-per-source grid cost 0.15 - 0.3 ms of wave launches per step).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void smp_bwd_gather_all(GF_GATHER_PARAMS,
                                                                                                      const int2 *__restrict__ items,
                                                                                                      int n_items) {
    // launch order: each XCD (blockIdx % 8) takes a contiguous run of the list
    unsigned blk;
    {
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = blockIdx.x % 8;
        blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    }
    const int i = __builtin_amdgcn_readfirstlane((int)(blk * 4 + (threadIdx.x >> 6)));
    if (i >= n_items) return;
    const int2 item = items[i];
    const int w = __builtin_amdgcn_readfirstlane(item.x), chunk = __builtin_amdgcn_readfirstlane(item.y & 0xffff);
    const int qc = __builtin_amdgcn_readfirstlane(item.y >> 16);   // which eight positions q of the source (sources above eight positions)
    const int sw = prev_s[w];
    // Round 4: at most EIGHT accumulators per item -- a source of more positions runs as ceil(s_w / 8) items of eight positions q each
    // (the row terms of a consumer are requested by each of them: L1 hits).  128 registers, four waves per SIMD instead of three: the
    // kernel waits on its gathers (holding it to two waves per SIMD costs 20 %), and the sixteen-accumulator path set its budget.
    switch (gfsmp::gather_pad(sw)) {
        case 1: gather_source<1, 1>(GF_GATHER_ARGS, w, chunk); break;
        case 2: gather_source<2, 2>(GF_GATHER_ARGS, w, chunk); break;
        case 4: gather_source<4, 4>(GF_GATHER_ARGS, w, chunk); break;
        case 5: gather_source<5, 5>(GF_GATHER_ARGS, w, chunk); break;
        case 6: gather_source<6, 6>(GF_GATHER_ARGS, w, chunk); break;
        case 8: gather_source<8, 8>(GF_GATHER_ARGS, w, chunk); break;
        case 10:
            if (qc == 0) gather_source<8, 8, 10>(GF_GATHER_ARGS, w, chunk, 0);
            else gather_source<2, 2, 10>(GF_GATHER_ARGS, w, chunk, 8);
            break;
        case 12:
            if (qc == 0) gather_source<8, 8, 12>(GF_GATHER_ARGS, w, chunk, 0);
            else gather_source<4, 4, 12>(GF_GATHER_ARGS, w, chunk, 8);
            break;
        case 16: gather_source<8, 8, 16>(GF_GATHER_ARGS, w, chunk, 8 * qc); break;
        case 32: gather_source<8, 8, 32>(GF_GATHER_ARGS, w, chunk, 8 * qc); break;
        default: gather_source<8, 8, 64>(GF_GATHER_ARGS, w, chunk, 8 * qc); break;   // (sources of 33 .. 64 positions: level 4 of a 48-atom molecule)
    }
}

}  // namespace

// the dedicated C = 64 kernels of the block products (weights resident in LDS, rows in registers; output-stationary weight gradients).
// GF_SMP_ROWPANEL=0 (read per call: the parity tests switch it) selects the grouped tiled GEMM launches every other channel count uses.
// Round 4: the split-operand row-panel products also run at C = 32 (32 x 32 blocks: one column half, two k-chunks per lane; weight
// gradients on smp_wgrad_direct<32>); the fp32-pipe variants, the panel combine and the small-product kernels stay C = 64 only.
static bool smp_c64_kernels(const gf_smp *s) {
    if (env_is("GF_SMP_ROWPANEL", '0')) return false;
    return s->cfg.nChanels == 64 || ((s->cfg.nChanels == 32 || s->cfg.nChanels == 16) && smp_split_products(s->ctx) && s->wbound != nullptr);
}
// compact projected matrix O = [O_loc | U] (2C) instead of [O_loc | Z | Z'] (3C): the dedicated C = 64 product kernels gather the
// transposed rows themselves; the tiled launches keep the three-block layout
bool smp_compact_o(const gf_smp *s) { return smp_c64_kernels(s); }
// SMP_2D_ver7's extra products inside the row-panel kernels (32 / 16 channels, prebuilt images, no slice dropout); else as GEMMs on T
static bool smp_extras_in_kernel(const gf_smp *s, int l) {
    const int C = s->cfg.nChanels;
    return s->n_extra && (C == 32 || C == 16) && smp_c64_kernels(s) && s->lv[l].wimg_ready && s->lv[l].wimg && !s->drop_on && smp_split_products(s->ctx) &&
           s->lv[l].trow && !env_is("GF_SMP_EXTRAS_IN_KERNEL", '0');
}
gf_status smp_fused_backward_level_grouped(gf_smp *s, int l, float *dKl, float *dbl);

// every level's block-permuted weight copy in one launch (gf_smp_forward, before the first level)
gf_status smp_fused_stack_all(gf_smp *s, const std::vector<const float *> &K) {
    const int L = s->cfg.nLevels, C = s->cfg.nChanels;
    for (int l0 = 1; l0 <= L; l0 += kStackLevels) {
        StackAll a;
        a.n = 0;
        for (int l = l0; l <= L && a.n < kStackLevels; ++l) {
            if (!(s->fused && smp_fused_supported(s, l))) continue;
            a.K[a.n] = K[l];
            a.stacked[a.n] = s->lv[l].Wst;
            ++a.n;
        }
        if (a.n > 0)
            GF_LAUNCH(s->ctx, "smpf_stack_w", stack_weights_all, dim3(64, a.n), dim3(256), 0, a, C, s->cfg.custom_matmul);
    }
    // ... and the split product kernels' weight images of every level, both directions (the backward pass reuses them)
    for (int l = 1; l <= L; ++l) s->lv[l].wimg_ready = false;
    if (smp_panel_channels(C) && smp_compact_o(s) && smp_split_products(s->ctx)) {
        std::vector<const float *> w, x;
        std::vector<void *> im;
        for (int l = 1; l <= L; ++l)
            if (s->fused && smp_fused_supported(s, l) && s->lv[l].wimg) {
                w.push_back(s->lv[l].Wst);
                x.push_back((s->n_extra && s->extra_w) ? s->extra_w + (size_t)(l - 1) * 3 * C * C : nullptr);   // (images 18 .. 20: SMP_2D_ver7's extra products)
                im.push_back(s->lv[l].wimg);
                s->lv[l].wimg_ready = true;
            }
        if (!w.empty()) {
            gf_status st = smp_split_build_images(s->ctx, w.data(), im.data(), (int)w.size(), C, x.data());
            if (st != GF_OK) return st;
        }
    }
    return GF_OK;
}

// zeros into the blocks of T that tables-forward skips (DevLevel::t_zeros), unless they are there already
gf_status smp_fused_ensure_zero_fill(gf_smp *s, int l) {
    gf_smp::DevLevel &d = s->lv[l];
    if (!d.t_zeros || d.t_filled || !d.rowflag) return GF_OK;
    const long long rows = s->lay.level[l].rows;
    if (s->cfg.nChanels == 16)
        GF_LAUNCH(s->ctx, "smpf_tables_fill", tables_zero_fill<16>, dim3((unsigned)((rows + kZeroFillRows - 1) / kZeroFillRows)), dim3(256), 0, d.Q,
                  d.rowflag, rows);
    else if (s->cfg.nChanels == 64)
        GF_LAUNCH(s->ctx, "smpf_tables_fill", tables_zero_fill<64>, dim3((unsigned)((rows + kZeroFillRows - 1) / kZeroFillRows)), dim3(256), 0, d.Q,
                  d.rowflag, rows);
    else
        GF_LAUNCH(s->ctx, "smpf_tables_fill", tables_zero_fill<32>, dim3((unsigned)((rows + kZeroFillRows - 1) / kZeroFillRows)), dim3(256), 0, d.Q,
                  d.rowflag, rows);
    d.t_filled = true;
    return GF_OK;
}

bool smp_fused_supported(const gf_smp *s, int l) {
    const int C = s->cfg.nChanels;
    if (C % 4 != 0 || C > 1024) return false;
    if (s->cfg.nContractions != 18) return false;  // SMP_2D_ver6 / ver7 (_10 / _50): op-by-op levels
    if (!s->cfg.square()) return false;            // a tower at its own halving channel counts (GF_SMP_PAD_CHANNELS=0): op-by-op levels
    const gfsmp::LevelLayout &h = s->lay.level[l];
    if (h.buckets.empty()) return false;
    if (s->drop_on) {   // RisiContraction_18_dropout: fused where the per-product row factors exist (round 5: the split row-panel kernels at
        // 32 / 16 channels -- the towers' padded widths -- with the panel combine-forward and the level's device-built statistics), else op by op
        const gf_smp::DevLevel &d = s->lv[l];
        if (!((C == 32 || C == 16) && smp_c64_kernels(s) && d.rowfac8 && d.nodefac && d.fwd_pan && d.dzmax && d.row_max && d.trow && s->bwd_gather) ||
            env_is("GF_SMP_FUSED_DROPOUT", '0'))
            return false;
    }
    if (h.buckets.back().s <= 32) return true;  // 8 * PPW at LPC = 16
    // Round 6: fields of 33 .. 64 positions at 64 / 32 / 16 (padded) channels -- the nodes above 32 run tables-forward on smp_tables_fwd_big and the two combine steps
    // on the workgroup kernels (big_part); the gather wants the SOURCES (level l - 1) within 32, the split row-panel products their packed tables
    const gf_smp::DevLevel &d = s->lv[l];
    const gfsmp::LevelLayout &hp = s->lay.level[l - 1];
    return h.buckets.back().s <= kFusedMaxField && smp_panel_channels(C) && smp_tables_fold_vectors(s) && smp_c64_kernels(s) &&
           smp_split_products(s->ctx) && s->bwd_gather && !hp.buckets.empty() && hp.buckets.back().s <= kGatherMaxS && d.trow && d.trowf && d.rowflag &&
           d.dzmax && d.row_max && d.fwd_pan && !env_is("GF_SMP_BIG_FIELDS", '0');
}

// Q buffer of the level ([rows][18C]) is carved as  T [rows][6C] | O / dO [rows][3C] | dT [rows][6C]
gf_status smp_fused_forward_level(gf_smp *s, int l, const float *Kl, const float *bl) {
    gf_ctx *ctx = s->ctx;
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels, nwin = (C + 63) / 64;
    const int rows = (int)h.rows, pairs = (int)h.pairs, nodes = h.nNodes;
    float *T = d.Q, *O = d.Q + (size_t)h.rows * T_COLS * C;
    gf_status st;
    // what this pass runs the level's products on decides the layout of O / dO (two or three blocks) at C = 32: the reverse sweep
    // follows the forward's choice, not the option's value at the time it runs (round-4 advice)
    s->lv[l].fwd_c64 = smp_c64_kernels(s);
    // The structurally-zero rows of the S_ab / T6 blocks (half of the rows at QM9 sizes, 0.73 GB of zeros a step at cfg3) are the
    // same rows every step of a prepared batch and nothing else writes there: their zeros go in once, tables-forward skips them.
    // Round 4: and they are not even written that once while every reader of T skips them -- the split product kernels and the packed
    // weight-gradient kernel read an absent block from a page of zeros, the sums over b never visit one -- which is the default path; a
    // reader that does not mask (fp32 pipe, tiled GEMMs) gets the fill before it runs (here, or ensure_zero_fill in the reverse sweep).
    if ((C == 64 || ((C == 32 || C == 16) && smp_tables_fold_vectors(s))) && d.rowflag && !env_is("GF_SMP_MASK_ZEROS", '0')) {
        s->lv[l].t_zeros = true;
        const bool readers_mask = smp_c64_kernels(s) && smp_split_products(ctx) && d.trowf && d.trow && (long long)rows < (1ll << 29);
        if (!readers_mask) {
            st = smp_fused_ensure_zero_fill(s, l);
            if (st != GF_OK) return st;
        }
    } else {
        s->lv[l].t_zeros = false;
    }
    const bool drop = s->drop_on;   // (smp_fused_supported: only where the factor tables exist)
    if (drop)
        GF_LAUNCH(ctx, "smpf_dropout_factors", build_dropout_factors, dim3(nodes), dim3(64), 0, d.keep_mask, s->drop_scale,
                  reinterpret_cast<const float2 *>(d.node_scale), d.node_s, d.node_row, d.nodefac, d.rowfac8);
    const bool lanes8 = smp_half_window(C) && smp_tables_fold_vectors(s);   // eight lanes per position (C = 32): classes of 8 NI positions
    const bool lanes4 = smp_quarter_window(C) && smp_tables_fold_vectors(s);   // four (C = 16): classes of 16 NI positions
    const std::vector<SizeClass> cls = classes_of(h, lanes4 ? 16 : lanes8 ? 8 : 4);
    for (const SizeClass &c : cls) {
        if (lanes4) {
            switch (c.ni) {
                case 1: st = launch_tables_fwd_wn<1, 4>(s, l, c); break;
                default: st = launch_tables_fwd_wn<2, 4>(s, l, c); break;
            }
        } else if (lanes8) {
            switch (c.ni) {
                case 1: st = launch_tables_fwd_wn<1, 8>(s, l, c); break;
                case 2: st = launch_tables_fwd_wn<2, 8>(s, l, c); break;
                default: st = launch_tables_fwd_wn<4, 8>(s, l, c); break;
            }
        } else {
            switch (c.ni) {
                case 1: st = launch_tables_fwd_w<1>(s, l, c); break;
                case 2: st = launch_tables_fwd_w<2>(s, l, c); break;
                case 4: st = launch_tables_fwd_w<4>(s, l, c); break;
                default: st = launch_tables_fwd_w<8>(s, l, c); break;
            }
        }
        if (st != GF_OK) return st;
    }
    const BigPart big = big_part(h);   // the nodes above 32 positions (none at QM9 sizes): their own tables-forward, see smp_tables_fwd_big
    if (big.nodes > 0) {
        const size_t lds_b = tables_big_lds(big.smax, C);
        st = opt_in_lds(ctx, smp_tables_fwd_big, lds_b);
        if (st != GF_OK) return st;
        const int flags = d.t_zeros ? 1 : 0;   // (as the launchers of smp_tables_fwd_w at these channel counts)
        GF_LAUNCH(ctx, "smpf_tables_fwd_big", smp_tables_fwd_big, dim3((unsigned)big.pairs), dim3(256), lds_b, s->lv[l - 1].f, d.rsum, d.Q, d.Vt, d.scal,
                  d.pair_src_row, d.pair_src_s, d.pi, d.pair_node, d.node_s, d.node_row, d.node_pair, big.pair0, C, flags,
                  flags ? d.rowflag : (const unsigned char *)nullptr);
    }
    {  // Fdc = [f[w][p,p] | f[w][p,c_w]] of the level below (read by smp_vectors and by the compact products)
        const gf_smp::DevLevel &pv = s->lv[l - 1];
        GF_LAUNCH(ctx, "smpf_diag_gather", diag_gather_fwd, dim3(s->lay.level[l - 1].nNodes), dim3(node_block(s->lay.level[l - 1], C)), 0, pv.f, d.Fdc, pv.node_s,
                  pv.node_row, pv.node_pair, pv.node_center, C);
    }
    if (!smp_tables_fold_vectors(s))
        GF_LAUNCH(ctx, "smpf_vectors", smp_vectors, dim3(nodes), dim3(node_block(h, C)), 0, T, d.Vt, d.scal, d.St, d.node_s, d.node_row,
                  d.node_pair, C, d.Fdc, d.pair_src_pair, d.pi);
    else if (big.nodes > 0)   // (smp_tables_fwd_w folds these sums itself; the big nodes' kernel leaves them to smp_vectors)
        GF_LAUNCH(ctx, "smpf_vectors", smp_vectors, dim3(big.nodes), dim3(256), 0, T, d.Vt, d.scal, d.St + (size_t)big.n0 * 4 * C, d.node_s + big.n0,
                  d.node_row + big.n0, d.node_pair + big.n0, C, d.Fdc, d.pair_src_pair, d.pi);
    const size_t CC = (size_t)C * C;
    if (drop) {   // the vector / scalar slices' factors ride on the operands: Vt blocks (1, 3, 7, 10), St blocks (4, 13, 14, 17)
        const long long nv = (long long)pairs * C, ns = (long long)nodes * C;
        GF_LAUNCH(ctx, "smpf_dropout_scale", scale_node_blocks, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, d.Vt, d.pair_node, d.nodefac, 1, 3, 7, 10, C,
                  (long long)pairs);
        GF_LAUNCH(ctx, "smpf_dropout_scale", scale_node_blocks, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, d.St, (const int *)nullptr, d.nodefac, 4, 13, 14,
                  17, C, (long long)nodes);
    }
    {
        // V = Vt [K1;K3;K7;K10], S = St [K4;K13;K14;K17], Gc = [Fd K15 | Fc K16]: the small products of the level, ONE launch
        const int prevPairs = (int)s->lay.level[l - 1].pairs;
        const GemmSpec sm[4] = {
            {d.Vt, d.Wst + 10 * CC, d.Vout, pairs, C, 4 * C, 4 * C, C, C, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}},
            {d.St, d.Wst + 14 * CC, d.Sout, nodes, C, 4 * C, 4 * C, C, C, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}},
            {d.Fdc, d.Wst + 8 * CC, d.Gc, prevPairs, C, C, 2 * C, C, 2 * C, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}},
            {d.Fdc + C, d.Wst + 9 * CC, d.Gc + C, prevPairs, C, C, 2 * C, C, 2 * C, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}},
        };
        if (smp_panel_channels(C) && d.wimg_ready) {   // wave per 32-row panel on the f16 pipe (smp_small_split)
            const int prog[3] = {0, 2, 0}, nrows[3] = {pairs, prevPairs, nodes}, pos0[3] = {10, 8, 14};
            const float *in[3] = {d.Vt, d.Fdc, d.St};
            float *out[3] = {d.Vout, d.Gc, d.Sout};
            st = smp_small_split_c64(ctx, false, 3, prog, in, out, nrows, pos0, d.wimg, "smpf_small_nn", C);
        } else {
            st = gemm_grouped_free(ctx, false, sm, 4, "smpf_small_nn");
        }
        if (st != GF_OK) return st;
    }
    const int ocols = smp_compact_o(s) ? 2 : O_COLS;
    const int ldt = T_COLS * C, ldo = O_COLS * C;   // (the tiled launches below always use the three-block layout)
    // block GEMMs: A = T column range, B = stacked weights, C = O column block -- one grouped launch (every row panel
    // of T is fetched from HBM once and shared through L2 by the three products), separate launches as a fallback.
    //   O_LOC = tot [S_ab|S_bc][K0;K2] + tr S_ab K6 + [T6|T10][K5;K9]   (three K pieces, the first two row-scaled)
    {
        const long long tC = C, wCC = (long long)CC;
        GemmSpec sp[3] = {
            {T, d.Wst, O + O_LOC * C, rows, C, 5 * C, ldt, C, ldo, 3, {T_SAB * tC, T_SAB * tC, T_T6 * tC, 0}, {0 * wCC, 2 * wCC, 3 * wCC, 0},
             {2 * C, C, 2 * C, 0}, d.rowscale, 2, {0, 1, -1, -1}},
            // Z = [S_ab|S_bc][K8;K12] (stack 5,6), Z' = S_ab K11 (stack 7); the D_bb K15 / D_ac K16 terms come from Gc
            {T + T_SAB * C, d.Wst + 5 * CC, O + O_Z * C, rows, C, 2 * C, ldt, C, ldo, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0,
             {-1, -1, -1, -1}},
            {T + T_SAB * C, d.Wst + 7 * CC, O + O_ZP * C, rows, C, C, ldt, C, ldo, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0,
             {-1, -1, -1, -1}},
        };
        if (smp_c64_kernels(s)) {
            st = smp_rowpanel_products_c64(ctx, true, T, drop ? d.rowfac8 : d.rowscale, d.Wst, O, rows, d.trow, d.trowf, false,
                                           d.wimg_ready ? d.wimg : nullptr, C, drop ? 8 : 2, smp_extras_in_kernel(s, l) ? 3 : 0);  // weights in LDS
            if (st != GF_OK) return st;
        } else if (gemm_grouped_supported(sp, 3, false, false)) {
            st = gemm_grouped_rows(ctx, false, false, sp, 3, rows);
            if (st != GF_OK) return st;
        } else {
            struct G { int tcol, kb, wpos, ocol, scol, acc; };
            const G gs[5] = {{T_SAB, 2, 0, O_LOC, 0, 0}, {T_SAB, 1, 2, O_LOC, 1, 1}, {T_T6, 2, 3, O_LOC, -1, 1},
                             {T_SAB, 2, 5, O_Z, -1, 0}, {T_SAB, 1, 7, O_ZP, -1, 0}};
            for (const G &g : gs) {
                st = gemm_rs(ctx, false, false, rows, C, g.kb * C, T + g.tcol * C, ldt, 0, d.Wst + g.wpos * CC, C, 0, O + g.ocol * C, ldo,
                             0, 1, g.acc, d.rowscale, 2, g.scol);
                if (st != GF_OK) return st;
            }
        }
    }
    if (s->n_extra && !smp_extras_in_kernel(s, l)) {   // SMP_2D_ver7 on the 18-slice level: O_loc += S_ab X_a + S_bc X_b + tr S_bc X_c (gf_smp::n_extra), plain fp32 GEMMs on T
        if (!s->extra_w) return fail(ctx, GF_ERR_INVALID, "fused level %d: the extra products' weights are not bound", l);
        st = smp_fused_ensure_zero_fill(s, l);   // (these readers do not mask the absent S_ab blocks)
        if (st != GF_OK) return st;
        const float *X = s->extra_w + (size_t)(l - 1) * 3 * CC;
        const int ldO = ocols * C;
        static_assert(T_SBC == T_SAB + 1, "[S_ab | S_bc] [X_a; X_b] as one product of depth 2 C");
        // ONE launch: three K pieces (S_ab X_a, S_bc X_b, tr S_bc X_c), accumulated into O_loc; two launches where the segmented form
        // does not apply (16 channels: a piece is shorter than the GEMM's k-step)
        const GemmSpec xs = {T, X, O + O_LOC * C, rows, C, 3 * C, ldt, C, ldO, 3, {(long long)T_SAB * C, (long long)T_SBC * C, (long long)T_SBC * C, 0},
                             {0, (long long)CC, 2 * (long long)CC, 0}, {C, C, C, 0}, d.rowscale, 2, {-1, -1, 1, -1}};
        if (gemm_grouped_supported(&xs, 1, false, false)) {
            st = gemm_grouped_rows(ctx, false, false, &xs, 1, rows, 1);
        } else {
            st = gemm_rs(ctx, false, false, rows, C, 2 * C, T + T_SAB * C, ldt, 0, X, C, 0, O + O_LOC * C, ldO, 0, 1, 1, nullptr, 0, -1);
            if (st == GF_OK) st = gemm_rs(ctx, false, false, rows, C, C, T + T_SBC * C, ldt, 0, X + 2 * CC, C, 0, O + O_LOC * C, ldO, 0, 1, 1, d.rowscale, 2, 1);
        }
        if (st != GF_OK) return st;
    }
    if (smp_panel_channels(C) && ocols == 2 && d.fwd_pan && (long long)rows * 512 < 0x3fffffffll)
    {   // wave per row panel, the adjacency product on the matrix pipe; the top level leaves the readout's partial sums behind, the
        // others the per-channel maxima the level above scales its weight-gradient operands with
        float *psum = (l == s->cfg.nLevels || s->cfg.physics) ? d.psum : nullptr;   // (a tower reads every level out)
        st = smp_combine_fwd_panels_c64(s, l, O, bl, psum, d.pmax, drop ? d.nodefac : nullptr);
        if (st == GF_OK && psum) s->lv[l].psum_ready = true;
        if (st == GF_OK && d.pmax && big.nodes == 0) s->lv[l].pmax_ready = true;   // (maxima of the panels only: a level with bigger nodes is scanned)
        if (st == GF_OK && big.nodes > 0) {   // the nodes above 32 positions: workgroup per (node, four x), a 64-position build
            const size_t lds = combine_lds<16>(big.smax);
            st = opt_in_lds(ctx, smp_combine_fwd<16, kFusedMaxField>, lds);
            if (st != GF_OK) return st;
            GF_LAUNCH(ctx, "smpf_combine_fwd_big", (smp_combine_fwd<16, kFusedMaxField>), dim3((unsigned)(big.quads * nwin)), dim3(kThreads), lds, O, d.adj,
                      d.Vout, d.Sout, bl, d.f, d.quad_node + big.quad0, d.quad_b0 + big.quad0, d.node_s, d.node_row, d.node_pair, C, nwin, d.Gc,
                      d.pair_src_pair, d.pi, d.rsum, ocols, drop ? d.nodefac : (const float *)nullptr);
        }
        return st;
    }
    if (drop) return fail(ctx, GF_ERR_UNSUPPORTED, "fused level %d: slice dropout needs the panel combine-forward", l);
    {   // (a level beyond the panel kernel's 32-bit offsets, or another channel count: the workgroup kernel for every node)
        const size_t lds = combine_lds<16>(h.buckets.back().s);
        if (big.nodes > 0) {   // (a 64-position build where the level has nodes above 32 positions)
            st = opt_in_lds(ctx, smp_combine_fwd<16, kFusedMaxField>, lds);
            if (st != GF_OK) return st;
            GF_LAUNCH(ctx, "smpf_combine_fwd", (smp_combine_fwd<16, kFusedMaxField>), dim3((unsigned)(h.quad_node.size() * nwin)), dim3(kThreads), lds, O,
                      d.adj, d.Vout, d.Sout, bl, d.f, d.quad_node, d.quad_b0, d.node_s, d.node_row, d.node_pair, C, nwin, d.Gc,
                      d.pair_src_pair, d.pi, d.rsum, ocols);
            return GF_OK;
        }
        st = opt_in_lds(ctx, smp_combine_fwd<16>, lds);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smpf_combine_fwd", (smp_combine_fwd<16>), dim3((unsigned)(h.quad_node.size() * nwin)), dim3(kThreads), lds, O,
                  d.adj, d.Vout, d.Sout, bl, d.f, d.quad_node, d.quad_b0, d.node_s, d.node_row, d.node_pair, C, nwin, d.Gc,
                  d.pair_src_pair, d.pi, d.rsum, ocols);
    }
    return GF_OK;
}

// Remainder of a level's reverse sweep after combine-backward, with the small work batched:
//   smp_reduce_pairs            dSout per node + column partials of the bias gradient          (was 3 launches)
//   diag_gather_bwd             dGc
//   ONE NT launch               dFdc = [dG15 K15^T | dG16 K16^T], dVt = dVout [K1;K3;K7;K10]^T, dSt = dSout [K4;..]^T   (was 3)
//   smp_wgrad_c64               row-range images of the eight row products
//   ONE TN launch               dK15, dK16, Vt^T dVout, St^T dSout, each split over its own rows    (was 4 + their folds)
//   smp_fold_level              every image folded, un-stacked and added into dK_l / db_l          (was ~8 launches)
//   row-panel products          dT
gf_status smp_fused_backward_level_grouped(gf_smp *s, int l, float *dKl, float *dbl) {
    gf_ctx *ctx = s->ctx;
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const gf_smp::DevLevel &d = s->lv[l], &pv = s->lv[l - 1];
    const int C = s->cfg.nChanels;
    const int rows = (int)h.rows, pairs = (int)h.pairs, nodes = h.nNodes;
    const int prevNodes = s->lay.level[l - 1].nNodes, prevPairs = (int)s->lay.level[l - 1].pairs;
    float *T = d.Q, *dO = d.Q + (size_t)h.rows * T_COLS * C, *dT = dO + (size_t)h.rows * O_COLS * C;
    bool x_wgrad_done = false;   // the extra products' weight gradients came out of the level's weight-gradient kernel
    bool x_bwd_done = false;     // ... and their share of dT out of the backward row-panel kernel
    const size_t CC = (size_t)C * C;
    const int ldt = T_COLS * C, ldo = O_COLS * C;
    const GemmSpec none = {nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}};
    auto spec = [&](const float *A, const float *B, float *Cm, int M, int N, int K, int lda, int ldb, int ldc) {
        GemmSpec z = none;
        z.A = A; z.B = B; z.C = Cm; z.M = M; z.N = N; z.K = K; z.lda = lda; z.ldb = ldb; z.ldc = ldc;
        return z;
    };
    gf_status st;
    const int npb = (nodes + 255) / 256, nb = (nodes + npb - 1) / npb;   // <= 256 column partials per level
    float *colpart = s->colpart + (size_t)l * 256 * C;
    const int ocols = d.fwd_c64 ? 2 : O_COLS;
    const bool drop = s->drop_on;
    const float *nfac = drop ? d.nodefac : (const float *)nullptr;
    if (!env_is("GF_SMP_FUSE_SMALL", '0') && prevNodes > 0) {
        GF_LAUNCH(ctx, "smpf_diag_gather_bwd", smp_reduce_pairs_and_diag_gather, dim3((unsigned)(nb + prevNodes)), dim3(256), 0, d.dSpart, d.dbpart, d.dSout,
                  colpart, d.node_s, d.node_pair, C, nodes, npb, nb, dO, d.dGc, pv.node_s, pv.node_pair, d.cons_ptr, d.cons_row, d.cons_s, d.cons_a,
                  d.cons_inv_off, d.inv, C, ocols, nfac, d.cons_pair, d.pair_node);
    } else {
        GF_LAUNCH(ctx, "smpf_reduce_pairs", smp_reduce_pairs, dim3(nb), dim3(256), 0, d.dSpart, d.dbpart, d.dSout, colpart, d.node_s,
                  d.node_pair, C, nodes, npb);
        GF_LAUNCH(ctx, "smpf_diag_gather_bwd", diag_gather_bwd, dim3(prevNodes), dim3(node_block(s->lay.level[l - 1], C)), 0, dO, d.dGc, pv.node_s, pv.node_pair,
                  d.cons_ptr, d.cons_row, d.cons_s, d.cons_a, d.cons_inv_off, d.inv, C, ocols, nfac, d.cons_pair, d.pair_node);
    }
    {
        const GemmSpec nt[4] = {spec(d.dGc, d.Wst + 8 * CC, d.dFdc, prevPairs, C, C, 2 * C, C, 2 * C),
                                spec(d.dGc + C, d.Wst + 9 * CC, d.dFdc + C, prevPairs, C, C, 2 * C, C, 2 * C),
                                spec(d.dVout, d.Wst + 10 * CC, d.dVt, pairs, 4 * C, C, C, C, 4 * C),
                                spec(d.dSout, d.Wst + 14 * CC, d.dSt, nodes, 4 * C, C, C, C, 4 * C)};
        if (smp_panel_channels(C) && d.wimg_ready) {
            const int prog[3] = {1, 2, 1}, nrows[3] = {pairs, prevPairs, nodes}, pos0[3] = {10, 8, 14};
            const float *in[3] = {d.dVout, d.dGc, d.dSout};
            float *out[3] = {d.dVt, d.dFdc, d.dSt};
            st = smp_small_split_c64(ctx, true, 3, prog, in, out, nrows, pos0, d.wimg, "smpf_small_nt", C);
        } else
        st = gemm_grouped_free(ctx, true, nt, 4, "smpf_small_nt");
        if (st != GF_OK) return st;
    }
    if (drop) {   // (the gradients of the scaled operands: the same factors once more)
        const long long nv = (long long)pairs * C, ns = (long long)nodes * C;
        GF_LAUNCH(ctx, "smpf_dropout_scale", scale_node_blocks, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, d.dVt, d.pair_node, d.nodefac, 1, 3, 7, 10, C,
                  (long long)pairs);
        GF_LAUNCH(ctx, "smpf_dropout_scale", scale_node_blocks, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, d.dSt, (const int *)nullptr, d.nodefac, 4, 13, 14,
                  17, C, (long long)nodes);
    }
    FoldArgs fa;
    fa.ngroups = 6;
    float *ws = static_cast<float *>(ctx->ws);
    size_t ws_floats = ctx->ws_bytes / sizeof(float), used = 0;
    FoldGroup rowg;
    const bool stationary = d.fwd_c64;
    if (stationary && (C == 32 || C == 16)) {   // smp_wgrad_direct<32 | 16>: one partial image of the eight products per workgroup
        const int splits = smp_wgrad_direct_splits(ctx, rows);
        if ((size_t)splits * 8 * CC > ws_floats) return fail(ctx, GF_ERR_NOMEM, "fused level: workspace too small for %d weight-gradient images", splits);
        // SMP_2D_ver7 on the 18-slice level: its three extra products ride in the same kernel (their operands are fragments it already
        // holds); three more images per workgroup behind the eight, folded straight into dX below
        float *xpart = nullptr;
        if (s->n_extra && s->extra_g && !drop && smp_wgrad_extra_supported(2) &&
            (size_t)splits * 8 * CC + ((size_t)splits + (splits + 31) / 32) * 3 * CC <= ws_floats)
            xpart = ws + (size_t)splits * 8 * CC;
        unsigned *words = s->wbound + (size_t)l * smp_wgrad_direct_words_c32();   // (the same scratch layout at 16 channels)
        const unsigned *chan = nullptr;
        if (d.dzmax && d.row_max) {   // per-channel maxima of f_{l-1} and of this level's dz (combine-backward's per-workgroup maxima)
            GF_HIP_TRY(ctx, hipMemsetAsync(words, 0, sizeof(unsigned) * 64, ctx->stream));
            // (combine-backward's maxima: one row of a window's width per workgroup -- 32-channel windows at C = 32)
            // (the largest |f_{l-1}| per channel from the per-panel maxima combine-forward of the level below left behind, as at C = 64 --
            //  f_{l-1} itself was read for them: 94 MB at cfg3's level 3 -- or from f_0)
            const bool pm = pv.pmax && pv.pmax_ready;
            st = smp_wgrad_channel_maxima_ld(ctx, pm ? pv.pmax : pv.f, pm ? (long long)pv.fwd_npanels : (long long)s->lay.level[l - 1].rows, C, d.dzmax,
                                             d.dz_rows, d.dz_ld, C, words);
            if (st != GF_OK) return st;
            if (d.dz_rows2 > 0) {   // (the big nodes' workgroups: maxima of another row width, folded into the same words -- atomicMax)
                st = smp_wgrad_channel_maxima_ld(ctx, pv.f, 0, C, d.dzmax + d.dz_off2, d.dz_rows2, d.dz_ld2, C, words);
                if (st != GF_OK) return st;
            }
            chan = words;
        }
        if (!chan) {   // (host-built tables: the exact column bounds are maxima over ALL of T -- the absent blocks need their zeros)
            st = smp_fused_ensure_zero_fill(s, l);
            if (st != GF_OK) return st;
        }
        st = smp_wgrad_partials_direct_c32(ctx, T, dO, drop ? d.rowfac8 : d.rowscale, rows, splits, ws, d.trow, d.trowf, words, chan,
                                           (float)h.buckets.back().s, d.row_max, drop ? 8 : 2, C, xpart);
        if (st != GF_OK) return st;
        if (xpart) {
            st = splitk_fold(ctx, xpart, s->extra_g + (size_t)(l - 1) * 3 * CC, 3 * CC, splits, 0);
            if (st != GF_OK) return st;
            x_wgrad_done = true;
        }
        rowg.part = ws;
        rowg.splits = splits;
        rowg.n = 8 * CC;
        used = (size_t)splits * 8 * CC;
    } else if (stationary) {
        unsigned *wb = (s->wbound && ocols == 2 && smp_split_products(ctx)) ? s->wbound + (size_t)l * smp_wgrad_bound_words() : nullptr;
        if (wb && !d.dzmax) wb = nullptr;
        WgradScales sc;
        if (wb) {   // the operand columns' exponents come from the largest |f_{l-1}| and |dz_l| of every channel: the per-panel maxima
            // combine-forward of the level below left behind (or f_0 itself) and the per-workgroup maxima of this level's
            // combine-backward -- 17 - 19 MB of partials at cfg3's level 3, reduced by one small launch
            const bool pm = pv.pmax && pv.pmax_ready;
            st = smp_wgrad_channel_maxima(ctx, pm ? pv.pmax : pv.f, pm ? (long long)pv.fwd_npanels : (long long)s->lay.level[l - 1].rows, d.dzmax,
                                          d.dz_rows, wb);
            if (st != GF_OK) return st;
            sc.chan = wb;
            sc.smax = (float)h.buckets.back().s;
            sc.max_tot = d.max_tot, sc.max_tr = d.max_tr, sc.row_max = d.row_max;
        }
        {   // (a weight-gradient kernel that reads the absent blocks of T needs their zeros: see smp_fused_forward_level)
            const bool packed = ocols == 2 && sc.any() && smp_split_products(ctx) && d.trowf && (long long)rows < (1ll << 29) &&
                                !env_is("GF_SMP_MASK_ZEROS", '0');
            if (!packed) {
                st = smp_fused_ensure_zero_fill(s, l);
                if (st != GF_OK) return st;
            }
        }
        st = smp_wgrad_partials_c64(ctx, T, dO, d.rowscale, rows, ws, ws_floats, &rowg, ocols == 2 ? d.trow : nullptr, sc, d.trowf);
        if (st != GF_OK) return st;
        used = (size_t)rowg.splits * rowg.n;
    } else {  // other channel counts: the grouped split-K launch (its own ordered reduction) into the stacked image, one "image"
        struct G { int tcol, kb, wpos, ocol, scol; };
        const G gs[5] = {{T_SAB, 2, 0, O_LOC, 0}, {T_SAB, 1, 2, O_LOC, 1}, {T_T6, 2, 3, O_LOC, -1}, {T_SAB, 2, 5, O_Z, -1}, {T_SAB, 1, 7, O_ZP, -1}};
        GemmSpec sp[5];
        for (int i = 0; i < 5; ++i) {
            sp[i] = spec(T + gs[i].tcol * C, dO + gs[i].ocol * C, d.dWst + gs[i].wpos * CC, gs[i].kb * C, C, rows, ldt, ldo, C);
            sp[i].rs = gs[i].scol >= 0 ? d.rowscale : nullptr;
            sp[i].rs_ld = 2;
            sp[i].scol[0] = gs[i].scol;
        }
        if (C <= 64 && gemm_grouped_supported(sp, 5, true, false)) {
            st = gemm_grouped_splitk(ctx, sp, 5, rows, d.dWst, 0);
            if (st != GF_OK) return st;
        } else {
            for (int i = 0; i < 5; ++i) {
                st = gemm_rs(ctx, true, false, gs[i].kb * C, C, rows, T + gs[i].tcol * C, ldt, 0, dO + gs[i].ocol * C, ldo, 0,
                             d.dWst + gs[i].wpos * CC, C, 0, 1, 0, d.rowscale, 2, gs[i].scol);
                if (st != GF_OK) return st;
            }
        }
        rowg.part = d.dWst;
        rowg.splits = 1;
        rowg.n = 8 * CC;
        // (the split launches above used the workspace from its base; their reductions have been issued, and the launch
        //  below is ordered behind them on the same stream)
    }
    FoldGroup small[4];
    {
        const GemmSpec tn[4] = {spec(d.Fdc, d.dGc, nullptr, C, C, prevPairs, 2 * C, 2 * C, C),
                                spec(d.Fdc + C, d.dGc + C, nullptr, C, C, prevPairs, 2 * C, 2 * C, C),
                                spec(d.Vt, d.dVout, nullptr, 4 * C, C, pairs, 4 * C, C, C),
                                spec(d.St, d.dSout, nullptr, 4 * C, C, nodes, 4 * C, C, C)};
        GemmSpec tnc[4];
        for (int i = 0; i < 4; ++i) {
            tnc[i] = tn[i];
            tnc[i].C = ws + used;  // (alignment check only: the launcher places the images itself)
        }
        st = gemm_grouped_free_tn(ctx, tnc, 4, ws + used, ws_floats - used, small, "smpf_small_tn");
        if (st != GF_OK) return st;
    }
    const FoldGroup *grp[6] = {&rowg, &small[0], &small[1], &small[2], &small[3], nullptr};
    const unsigned firsts[6] = {0u, (unsigned)(8 * CC), (unsigned)(9 * CC), (unsigned)(10 * CC), (unsigned)(14 * CC), (unsigned)(18 * CC)};
    for (int g = 0; g < 5; ++g) {
        fa.part[g] = grp[g]->part;
        fa.splits[g] = grp[g]->splits;
        fa.n[g] = (unsigned)grp[g]->n;
        fa.first[g] = firsts[g];
    }
    fa.part[5] = colpart;
    fa.splits[5] = nb;
    fa.n[5] = (unsigned)C;
    fa.first[5] = firsts[5];
    const unsigned total = (unsigned)(18 * CC + C);
    GF_LAUNCH(ctx, "smpf_fold", smp_fold_level, dim3((total + 63) / 64), dim3(256), 0, fa, dKl, dbl, C, s->cfg.custom_matmul, total);
    st = smp_dp_level_done(s, l);  // data-parallel: dK_l and db_l are final -- their all-reduce runs beside what follows
    if (st != GF_OK) return st;
    // table gradients dT from dO
    if (d.fwd_c64) {
        // (with the consumer gather reading dT, the gradients of the structurally-zero S_ab / T6 rows have no reader: not written)
        st = smp_rowpanel_products_c64(ctx, false, dO, drop ? d.rowfac8 : d.rowscale, d.Wst, dT, rows, ocols == 2 ? d.trow : nullptr, d.trowf,
                                       smp_fused_gather_enabled(s, l), d.wimg_ready ? d.wimg : nullptr, C, drop ? 8 : 2,
                                       (ocols == 2 && smp_extras_in_kernel(s, l)) ? 3 : 0);
        x_bwd_done = ocols == 2 && smp_extras_in_kernel(s, l);
        if (st != GF_OK) return st;
    } else {
        const long long oC = C, wCC = (long long)CC;
        GemmSpec dg[3] = {
            {dO, d.Wst, dT + T_SAB * C, rows, C, 4 * C, ldo, C, ldt, 4, {O_LOC * oC, O_LOC * oC, O_Z * oC, O_ZP * oC},
             {0 * wCC, 2 * wCC, 5 * wCC, 7 * wCC}, {C, C, C, C}, d.rowscale, 2, {0, 1, -1, -1}},
            {dO, d.Wst, dT + T_SBC * C, rows, C, 2 * C, ldo, C, ldt, 2, {O_LOC * oC, O_Z * oC, 0, 0}, {1 * wCC, 6 * wCC, 0, 0}, {C, C, 0, 0},
             d.rowscale, 2, {0, -1, -1, -1}},
            {dO, d.Wst, dT + T_T6 * C, rows, 2 * C, C, ldo, C, ldt, 1, {O_LOC * oC, 0, 0, 0}, {3 * wCC, 0, 0, 0}, {C, 0, 0, 0}, nullptr, 0,
             {-1, -1, -1, -1}},
        };
        if (gemm_grouped_supported(dg, 3, false, true)) {
            st = gemm_grouped_rows(ctx, false, true, dg, 3, rows);
            if (st != GF_OK) return st;
        } else {
            struct H { int ocol, wpos, kb, tcol, acc, scol; };
            const H hs[6] = {{O_Z, 5, 2, T_SAB, 0, -1}, {O_LOC, 0, 1, T_SAB, 1, 0}, {O_LOC, 1, 1, T_SBC, 1, 0}, {O_LOC, 2, 1, T_SAB, 1, 1},
                             {O_ZP, 7, 1, T_SAB, 1, -1}, {O_LOC, 3, 2, T_T6, 0, -1}};
            for (const H &g : hs) {
                st = gemm_rs(ctx, false, true, rows, g.kb * C, C, dO + g.ocol * C, ldo, 0, d.Wst + g.wpos * CC, C, 0, dT + g.tcol * C, ldt, 0,
                             1, g.acc, d.rowscale, 2, g.scol);
                if (st != GF_OK) return st;
            }
        }
    }
    if (s->n_extra) {   // the extra products of SMP_2D_ver7 (see the forward): dX = T-block^T L, dT-blocks += L X^T
        if (!s->extra_w || !s->extra_g) return fail(ctx, GF_ERR_INVALID, "fused level %d: the extra products' weights are not bound", l);
        if (!x_wgrad_done) {   // (the GEMM that reads T does not mask the absent S_ab blocks)
            st = smp_fused_ensure_zero_fill(s, l);
            if (st != GF_OK) return st;
        }
        const float *X = s->extra_w + (size_t)(l - 1) * 3 * CC;
        float *dX = s->extra_g + (size_t)(l - 1) * 3 * CC;
        const int ldO = ocols * C;
        const float *Lg = dO + O_LOC * C;
        // ([X_a; X_b] is one [2 C][C] matrix and [S_ab | S_bc] one [rows][2 C] operand: two products per direction instead of three)
        const GemmSpec none = {nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, nullptr, 0, {-1, -1, -1, -1}};
        GemmSpec wg[2] = {none, none};   // d[X_a; X_b] = [S_ab | S_bc]^T L and dX_c = (tr S_bc)^T L: one split launch over the rows, ordered fold
        wg[0].A = T + T_SAB * C, wg[0].B = Lg, wg[0].M = 2 * C, wg[0].N = C, wg[0].K = rows, wg[0].lda = ldt, wg[0].ldb = ldO, wg[0].ldc = C;
        wg[1] = wg[0];
        wg[1].A = T + T_SBC * C, wg[1].M = C, wg[1].rs = d.rowscale, wg[1].rs_ld = 2, wg[1].scol[0] = 1;
        GemmSpec bg[2] = {none, none};   // dT[S_ab] += L X_a^T; dT[S_bc] += L X_b^T + tr L X_c^T (two K pieces): one launch
        bg[0].A = Lg, bg[0].B = X, bg[0].C = dT + T_SAB * C, bg[0].M = rows, bg[0].N = C, bg[0].K = C, bg[0].lda = ldO, bg[0].ldb = C, bg[0].ldc = ldt;
        bg[1] = bg[0];
        bg[1].C = dT + T_SBC * C, bg[1].K = 2 * C, bg[1].nseg = 2, bg[1].b_off[0] = (long long)CC, bg[1].b_off[1] = 2 * (long long)CC;
        bg[1].klen[0] = bg[1].klen[1] = C, bg[1].rs = d.rowscale, bg[1].rs_ld = 2, bg[1].scol[0] = -1, bg[1].scol[1] = 1;
        if (gemm_grouped_supported(wg, 2, true, false) && gemm_grouped_supported(bg, 2, false, true) && C <= 64) {
            st = x_wgrad_done ? GF_OK : gemm_grouped_splitk(ctx, wg, 2, rows, dX, 0);
            // (the dS_ab rows of structural zeros were not written by the product kernel: what accumulates there is never read either)
            if (st == GF_OK && !x_bwd_done) st = gemm_grouped_rows(ctx, false, true, bg, 2, rows, 1);
        } else {
            st = x_wgrad_done ? GF_OK : gemm_rs(ctx, true, false, 2 * C, C, rows, T + T_SAB * C, ldt, 0, Lg, ldO, 0, dX, C, 0, 1, 0, nullptr, 0, -1);
            if (st == GF_OK && !x_wgrad_done) st = gemm_rs(ctx, true, false, C, C, rows, T + T_SBC * C, ldt, 0, Lg, ldO, 0, dX + 2 * CC, C, 0, 1, 0, d.rowscale, 2, 1);
            if (st == GF_OK && !x_bwd_done) st = gemm_rs(ctx, false, true, rows, 2 * C, C, Lg, ldO, 0, X, C, 0, dT + T_SAB * C, ldt, 0, 1, 1, nullptr, 0, -1);
            if (st == GF_OK && !x_bwd_done) st = gemm_rs(ctx, false, true, rows, C, C, Lg, ldO, 0, X + 2 * CC, C, 0, dT + T_SBC * C, ldt, 0, 1, 1, d.rowscale, 2, 1);
        }
        if (st != GF_OK) return st;
    }
    if (smp_fused_gather_enabled(s, l)) return GF_OK;  // dP is evaluated inside the consumer gather (smp_fused_gather_backward)
    st = ensure_P(s);
    if (st != GF_OK) return st;
    const std::vector<SizeClass> cls = classes_of(h, 4);
    for (const SizeClass &c : cls) {
        switch (c.ni) {
            case 1: st = launch_tables_bwd<1>(s, l, c, dT); break;
            case 2: st = launch_tables_bwd<2>(s, l, c, dT); break;
            case 4: st = launch_tables_bwd<4>(s, l, c, dT); break;
            default: st = launch_tables_bwd<8>(s, l, c, dT); break;
        }
        if (st != GF_OK) return st;
    }
    return GF_OK;
}

// df_l is given in d.df; produces dP in s->P, accumulates dK_l and db_l; the caller then runs the promotion backward.
gf_status smp_fused_backward_level(gf_smp *s, int l, const float *Kl, float *dKl, float *dbl, const float *node_df, bool rows_too) {
    gf_ctx *ctx = s->ctx;
    const gfsmp::LevelLayout &h = s->lay.level[l];
    const gf_smp::DevLevel &d = s->lv[l];
    // combine-backward reads dF rows, a per-node vector (the readout's broadcast), or -- a tower's level below the top -- both
    const float *dfrows = (node_df && !rows_too) ? nullptr : d.df;
    const int C = s->cfg.nChanels, nwin = (C + 63) / 64;
    float *dO = d.Q + (size_t)h.rows * T_COLS * C;
    gf_status st;
    // Slice dropout in TEST mode scales the forward values by nKept / 18 (RisiContraction_18_dropout.h:465-471) and its backward() does
    // not (:480-): a reverse sweep there is neither a derivative nor something the reference's drivers ever run (Predict is forward only).
    // The fused level carries one factor table per pass; the op-by-op levels reproduce that sweep (GF_SMP_FUSED_DROPOUT=0).
    if (s->drop_on && s->drop_scale != 1.f)
        return fail(ctx, GF_ERR_UNSUPPORTED, "gf_smp_backward: fused level %d under slice dropout in test mode (scale %.4f): set GF_SMP_FUSED_DROPOUT=0 "
                                             "for the reference's unscaled test-mode sweep", l, (double)s->drop_scale);
    // The forward pass wrote O in the layout of ITS product kernels; at C = 32 those exist on the split path only, so an option flipped
    // between the two passes would make this sweep read dO in the other layout: refused instead of differentiated wrongly.
    if (d.fwd_c64 != smp_c64_kernels(s))
        return fail(ctx, GF_ERR_INVALID, "gf_smp_backward: the product kernels' option (GF_OPT_SMP_FP32_PRODUCTS / GF_SMP_SPLIT / GF_SMP_ROWPANEL) "
                                         "changed since the forward pass of level %d", l);
    float *dzmax = (s->wbound && d.dzmax && d.fwd_c64) ? d.dzmax : (float *)nullptr;
    // round 5: on the forward's row panels where they exist (one wave per panel, every request up front; GF_SMP_COMBINE_BWD_PANELS=0: the
    // workgroup-per-(node, four x) kernel below, which also serves every other channel count)
    if (d.fwd_c64 && d.fwd_pan && smp_panel_channels(C) && (long long)h.rows * 512 < 0x3fffffffll && !env_is("GF_SMP_COMBINE_BWD_PANELS", '0')) {
        st = smp_combine_bwd_panels_c64(s, l, dfrows, node_df, dO, dzmax);
        if (st != GF_OK) return st;
        s->lv[l].dz_rows = d.fwd_npanels;
        s->lv[l].dz_ld = C;
        s->lv[l].dz_rows2 = 0;
        const BigPart big = big_part(h);
        if (big.nodes > 0) {   // the nodes above 32 positions: the workgroup kernel, its column maxima behind the panels' (one row of ITS window's
            // width per workgroup: at C = 64 the panels' width -- one table; at 32 / 16 channels a second set, dz_rows2)
            float *dz2 = dzmax ? dzmax + (size_t)d.fwd_npanels * C : (float *)nullptr;
            if (smp_half_window(C)) {
                const size_t lds = std::max(combine_lds<8>(big.smax), sizeof(float) * ((size_t)adj_lds_floats(big.smax) + 1024));
                st = opt_in_lds(ctx, smp_combine_bwd<8>, lds);
                if (st != GF_OK) return st;
                GF_LAUNCH(ctx, "smpf_combine_bwd_big", (smp_combine_bwd<8>), dim3((unsigned)(big.quads * (C / 32))), dim3(kThreads), lds, d.f, dfrows, node_df, d.adj,
                          dO, d.dVout, d.dSpart, d.dbpart, d.quad_node + big.quad0, d.quad_b0 + big.quad0, d.node_s, d.node_row, d.node_pair, C, C / 32, d.rsum, 2, dz2);
            } else {
                const size_t lds = std::max(combine_lds<16>(big.smax), sizeof(float) * ((size_t)adj_lds_floats(big.smax) + 1024));
                st = opt_in_lds(ctx, smp_combine_bwd<16>, lds);
                if (st != GF_OK) return st;
                GF_LAUNCH(ctx, "smpf_combine_bwd_big", (smp_combine_bwd<16>), dim3((unsigned)(big.quads * nwin)), dim3(kThreads), lds, d.f, dfrows, node_df, d.adj,
                          dO, d.dVout, d.dSpart, d.dbpart, d.quad_node + big.quad0, d.quad_b0 + big.quad0, d.node_s, d.node_row, d.node_pair, C, nwin, d.rsum, 2, dz2);
            }
            if (C == 64) {
                s->lv[l].dz_rows = d.fwd_npanels + big.quads;
            } else {
                s->lv[l].dz_rows2 = big.quads * (smp_half_window(C) ? C / 32 : nwin);
                s->lv[l].dz_off2 = (long long)d.fwd_npanels * C;
                s->lv[l].dz_ld2 = smp_half_window(C) ? 32 : 64;
            }
        }
        (void)Kl;
        return smp_fused_backward_level_grouped(s, l, dKl, dbl);
    }
    s->lv[l].dz_rows = (long long)h.quad_node.size();
    s->lv[l].dz_rows2 = 0;
    s->lv[l].dz_ld = smp_half_window(C) ? 32 : 64;   // (the workgroup kernels write one row of their window's width)
    if (smp_half_window(C)) {   // eight lanes per row (32-channel windows): at C = 32 every lane has channels
        const int nw8 = C / 32, N = h.buckets.back().s;
        // (the column maxima go through kThreads / 8 x 32 floats of the dz image: room for them whatever the field size)
        const size_t lds = std::max(combine_lds<8>(N), sizeof(float) * ((size_t)adj_lds_floats(N) + 1024));
        st = opt_in_lds(ctx, smp_combine_bwd<8>, lds);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smpf_combine_bwd", (smp_combine_bwd<8>), dim3((unsigned)(h.quad_node.size() * nw8)), dim3(kThreads), lds, d.f,
                  dfrows, node_df, d.adj, dO, d.dVout, d.dSpart, d.dbpart, d.quad_node, d.quad_b0, d.node_s, d.node_row, d.node_pair, C,
                  nw8, d.rsum, d.fwd_c64 ? 2 : O_COLS, dzmax);
    } else {
        const size_t lds = std::max(combine_lds<16>(h.buckets.back().s), sizeof(float) * ((size_t)adj_lds_floats(h.buckets.back().s) + 1024));
        st = opt_in_lds(ctx, smp_combine_bwd<16>, lds);
        if (st != GF_OK) return st;
        GF_LAUNCH(ctx, "smpf_combine_bwd", (smp_combine_bwd<16>), dim3((unsigned)(h.quad_node.size() * nwin)), dim3(kThreads), lds, d.f,
                  dfrows, node_df, d.adj, dO, d.dVout, d.dSpart, d.dbpart, d.quad_node, d.quad_b0, d.node_s, d.node_row, d.node_pair, C,
                  nwin, d.rsum, d.fwd_c64 ? 2 : O_COLS, dzmax);
    }
    (void)Kl;
    return smp_fused_backward_level_grouped(s, l, dKl, dbl);
}

// per-batch tables of smp_bwd_gather_v2, built on the device at prepare time on the handle's upload stream (behind the uploads
// of the consumer lists they are derived from)
// records of tables-forward: two int4 per node, in the level's (size class, molecule) order (mol_order):
//   {node, s, 0, 0}  {first row lo, hi, first pair lo, hi}
__global__ void build_tf_records(const int *__restrict__ mol_order, const int *__restrict__ node_s, const long long *__restrict__ node_row,
                                 const long long *__restrict__ node_pair, int4 *__restrict__ recs, int nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes) return;
    const int n = mol_order[i];
    const long long r = node_row[n], p = node_pair[n];
    recs[2 * i] = make_int4(n, node_s[n], 0, 0);
    recs[2 * i + 1] = make_int4((int)(unsigned)(r & 0xffffffffll), (int)(r >> 32), (int)(unsigned)(p & 0xffffffffll), (int)(p >> 32));
}

gf_status smp_build_tf_records(gf_smp *s, int l, hipStream_t stream) {
    gf_smp::DevLevel &d = s->lv[l];
    const int nodes = s->lay.level[l].nNodes;
    if (!d.tf_recs || nodes == 0) return GF_OK;
    hipLaunchKernelGGL(build_tf_records, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, stream, d.mol_order, d.node_s, d.node_row,
                       d.node_pair, d.tf_recs, nodes);
    GF_LAUNCH_CHECK(s->ctx, "build_tf_records");
    return GF_OK;
}

gf_status smp_build_gather_records(gf_smp *s, int l, hipStream_t stream) {
    gf_smp::DevLevel &d = s->lv[l];
    const gf_smp::DevLevel &pv = s->lv[l - 1];
    const gfsmp::LevelLayout &h = s->lay.level[l];
    if (!d.cons_hdr || h.pairs == 0) return GF_OK;
    const int C = s->cfg.nChanels;
    hipLaunchKernelGGL(build_gather_records, dim3((unsigned)s->lay.level[l - 1].nNodes), dim3(64), 0, stream, pv.node_s, d.cons_ptr,
                       d.cons_qbase, d.cons_s, d.cons_a, d.cons_row, d.cons_pair, d.pair_node, d.cons_inv_off, d.inv, d.rsum,
                       T_COLS * C * 4, d.cons_hdr, d.cons_qrec);
    GF_LAUNCH_CHECK(s->ctx, "build_gather_records");
    return GF_OK;
}

// the folded gather keeps a source node's row of accumulators in registers: receptive fields of level l-1 up to 32
bool smp_fused_gather_enabled(const gf_smp *s, int l) {
    const gfsmp::LevelLayout &hp = s->lay.level[l - 1];
    return s->bwd_gather && !hp.buckets.empty() && hp.buckets.back().s <= kGatherMaxS;
}

// df_{l-1} from the table gradients of level l without materialising dP (see smp_bwd_gather)
gf_status smp_fused_gather_backward(gf_smp *s, int l) {
    const gfsmp::LevelLayout &h = s->lay.level[l], &hp = s->lay.level[l - 1];
    const gf_smp::DevLevel &d = s->lv[l];
    const int C = s->cfg.nChanels;
    const float *dT = d.Q + (size_t)h.rows * T_COLS * C + (size_t)h.rows * O_COLS * C;
    if (!d.cons_hdr) return fail(s->ctx, GF_ERR_INVALID, "smp_fused_gather_backward: level %d has no gather records", l);
    if (!hp.buckets.empty() && hp.buckets.back().s > kGatherMaxS) return fail(s->ctx, GF_ERR_UNSUPPORTED, "smp_fused_gather_backward: receptive field > 64");
    const gf_smp::DevLevel &pv = s->lv[l - 1];
    gf_ctx *ctx = s->ctx;
    const GatherTables G = {d.cons_hdr, d.cons_qrec};
    const int n_items = (int)(hp.gather_items.size() / 2);
    if (n_items > 0)
        GF_LAUNCH(ctx, "smpf_bwd_gather", smp_bwd_gather_all, dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, dT, d.dVt, d.dSt, pv.df,
                  d.dFdc, pv.node_s, pv.node_row, pv.node_pair, pv.node_center, d.cons_ptr, d.cons_inv_off, d.cons_qbase, d.inv, G, C,
                  reinterpret_cast<const int2 *>(pv.gather_items), n_items);
    return GF_OK;
}

}  // namespace gf
