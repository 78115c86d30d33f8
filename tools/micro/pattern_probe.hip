// Read bandwidth of a [rows][256] fp32 matrix walked in 16-row tiles by 8-wave workgroups (tile = blockIdx + n gridDim), two ways:
//   strips: wave w reads the 128-byte column strip w of the tile (a wave instruction = 8 rows x 128 B, 2 KB apart)
//   rows:   wave w reads rows 2w, 2w+1 of the tile           (a wave instruction = 1 KB contiguous)
// hipcc --offload-arch=gfx950 -O3 tools/micro/pattern_probe.hip -o tools/micro/pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void rd(const float4 *__restrict__ T, long long rows, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (long long t = blockIdx.x; t * 16 < rows; t += gridDim.x) {
        const float4 *tile = T + t * 16 * 64;  // 64 float4 per row
        float4 a, b;
        if (MODE == 0) {
            const int q = lane & 7, pair = lane >> 3;
            a = tile[(2 * pair) * 64 + wave * 8 + q];
            b = tile[(2 * pair + 1) * 64 + wave * 8 + q];
        } else {
            a = tile[(2 * wave) * 64 + lane];
            b = tile[(2 * wave + 1) * 64 + lane];
        }
        s += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    }
    if (s == 123.456f) *out = s;
}
int main() {
    const long long rows = 3300000;
    float4 *buf;
    float *out;
    hipMalloc(&buf, rows * 1024);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, rows * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048})
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(512), 0, 0, buf, rows, out);
                else hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(512), 0, 0, buf, rows, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("grid %4d %s: %.3f ms  %.0f GB/s\n", grid, mode ? "rows  " : "strips", best, rows * 1024.0 / best / 1e6);
        }
    return 0;
}
