// Useful read bandwidth when only k of the 50 128-byte lines of every 6400-byte record are touched (the access pattern of the
// RisiContraction_50 backward kernels on G [x][z][50][32]): a wave owns 24 consecutive records and reads line j of records 2t, 2t+1
// with one dword load (lanes 0..31 / 32..63), as fam50_bwd_tables_mfma does.  Lines: the first k (contiguous) or k spread evenly.
// hipcc --offload-arch=gfx950 -O3 tools/micro/sparse_probe.hip -o tools/micro/sparse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K, int SPREAD>
__global__ __launch_bounds__(256) void rd(const float *__restrict__ G, long long nrows, float *out) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, m = lane & 31;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nrows) return;
    const float *g = G + w * 24 * 1600 + hi * 1600 + m;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int line = SPREAD ? (j * 50) / K : j;
        float v[12];
#pragma unroll
        for (int t = 0; t < 12; ++t) v[t] = g[(2 * t) * 1600 + line * 32];
#pragma unroll
        for (int t = 0; t < 12; ++t) s += v[t];
    }
    if (s == 123.456f) *out = s;
}
template <int K, int SPREAD>
void run(const float *buf, long long nrows, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<K, SPREAD>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, 0, buf, nrows, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("k = %2d of 50 lines, %s: %.3f ms  %.0f GB/s useful\n", K, SPREAD ? "spread    " : "contiguous", best, nrows * 24.0 * K * 128 / best / 1e6);
}
int main() {
    const long long nrows = 256 * 24;   // cfg5: 256 graphs x 24 rows x 24 records x 6400 B = 0.94 GB
    float *buf, *out;
    hipMalloc(&buf, nrows * 24 * 6400);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, nrows * 24 * 6400);
    run<5, 1>(buf, nrows, out);  run<9, 1>(buf, nrows, out);  run<15, 1>(buf, nrows, out); run<26, 1>(buf, nrows, out);
    run<41, 1>(buf, nrows, out); run<50, 1>(buf, nrows, out);
    run<9, 0>(buf, nrows, out);  run<15, 0>(buf, nrows, out);  run<26, 0>(buf, nrows, out);
    return 0;
}
