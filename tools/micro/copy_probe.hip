// copy_probe.hip -- which copy-kernel shape gets the box's best read+write rate (the yardstick of bench.py's roofline.hbm_copy_kernel_GBps).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/copy_probe.hip -o tools/micro/copy_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// A: grid-stride, U loads in flight, loads of a thread `gridDim * 256` float4 apart (what gf_hbm_copy_probe_f32 first did)
template <int U>
__global__ __launch_bounds__(256) void copy_strided(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) d[i + k * stride] = v[k];
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
// B: a block owns contiguous tiles of U * 256 float4 (U * 4 KiB); the U loads of a thread are 4 KiB apart
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tiled(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) {
                const float *p = reinterpret_cast<const float *>(s + i + k * 256);
                v[k] = make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3));
            } else v[k] = s[i + k * 256];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) {
                float *q = reinterpret_cast<float *>(d + i + k * 256);
                __builtin_nontemporal_store(v[k].x, q); __builtin_nontemporal_store(v[k].y, q + 1);
                __builtin_nontemporal_store(v[k].z, q + 2); __builtin_nontemporal_store(v[k].w, q + 3);
            } else d[i + k * 256] = v[k];
        }
    }
}
// C: one tile per block, no loop (n4 / (U*256) blocks)
template <int U>
__global__ __launch_bounds__(256) void copy_once(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n4) {
    const size_t i = (size_t)blockIdx.x * U * 256 + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = s[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) d[i + k * 256] = v[k];
}
// R / W: read-only (sum into a sink) and write-only, tiled
template <int U>
__global__ __launch_bounds__(256) void read_only(const float4 *__restrict__ s, float *sink, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    float acc = 0.f;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = s[i + k * 256];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 123.456f) *sink = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_only(float4 *__restrict__ d, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
#pragma unroll
        for (int k = 0; k < U; ++k) d[i + k * 256] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

// D: which half of the non-temporal copy matters: NTL (loads), NTS (stores)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_tiled_split(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NTL) {
                const float *p = reinterpret_cast<const float *>(s + i + k * 256);
                v[k] = make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3));
            } else v[k] = s[i + k * 256];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NTS) {
                float *q = reinterpret_cast<float *>(d + i + k * 256);
                __builtin_nontemporal_store(v[k].x, q); __builtin_nontemporal_store(v[k].y, q + 1);
                __builtin_nontemporal_store(v[k].z, q + 2); __builtin_nontemporal_store(v[k].w, q + 3);
            } else d[i + k * 256] = v[k];
        }
    }
}
// E: the same through raw buffer instructions with an explicit cache-policy immediate (what the library's kernels use): AUX bit 1 = nt
using u4 = __attribute__((ext_vector_type(4))) unsigned int;
template <int U, int LAUX, int SAUX>
__global__ __launch_bounds__(256) void copy_buffer(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(s + t * tile), 0, (unsigned)(tile * 16), 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(d + t * tile, 0, (unsigned)(tile * 16), 0x00020000);
        u4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)threadIdx.x * 16 + k * 4096, 0, LAUX);
#pragma unroll
        for (int k = 0; k < U; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], rd, (int)threadIdx.x * 16 + k * 4096, 0, SAUX);
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void write_only_nt(float4 *__restrict__ d, size_t n4) {
    const size_t tile = (size_t)U * 256, ntiles = n4 / tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t i = t * tile + threadIdx.x;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            float *q = reinterpret_cast<float *>(d + i + k * 256);
            __builtin_nontemporal_store(1.f, q); __builtin_nontemporal_store(2.f, q + 1);
            __builtin_nontemporal_store(3.f, q + 2); __builtin_nontemporal_store(4.f, q + 3);
        }
    }
}

template <typename F>
static double time_ms(F launch, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const size_t n = (size_t)1 << 28, n4 = n / 4;   // 1 GiB each way
    float4 *s, *d; float *sink;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4); hipMalloc(&sink, 4);
    hipMemset(s, 1, n * 4); hipMemset(d, 0, n * 4);
    const double gb = 2.0 * n * 4 / 1e9;
    auto rep = [&](const char *name, double ms, double bytes_gb) { std::printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes_gb / (ms * 1e-3)); };
    rep("hipMemcpyDtoD", time_ms([&] { hipMemcpyAsync(d, s, n * 4, hipMemcpyDeviceToDevice, 0); }, 5), gb);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "strided U=4 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(copy_strided<4>, dim3(blocks), dim3(256), 0, 0, s, d, n4); }, 5), gb);
        std::snprintf(nm, sizeof nm, "tiled U=4 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_tiled<4, false>), dim3(blocks), dim3(256), 0, 0, s, d, n4); }, 5), gb);
        std::snprintf(nm, sizeof nm, "tiled U=8 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_tiled<8, false>), dim3(blocks), dim3(256), 0, 0, s, d, n4); }, 5), gb);
        std::snprintf(nm, sizeof nm, "tiled U=4 NT blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_tiled<4, true>), dim3(blocks), dim3(256), 0, 0, s, d, n4); }, 5), gb);
        std::snprintf(nm, sizeof nm, "tiled U=2 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_tiled<2, false>), dim3(blocks), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    }
    rep("once U=1 (one float4 per thread)", time_ms([&] { hipLaunchKernelGGL(copy_once<1>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("once U=2", time_ms([&] { hipLaunchKernelGGL(copy_once<2>, dim3((unsigned)(n4 / 512)), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("once U=4", time_ms([&] { hipLaunchKernelGGL(copy_once<4>, dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("once U=8", time_ms([&] { hipLaunchKernelGGL(copy_once<8>, dim3((unsigned)(n4 / 2048)), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    for (int blocks : {2048, 8192}) {
        char nm[96];
        std::snprintf(nm, sizeof nm, "read-only U=4 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(read_only<4>, dim3(blocks), dim3(256), 0, 0, s, sink, n4); }, 5), gb / 2);
        std::snprintf(nm, sizeof nm, "write-only U=4 blocks=%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(write_only<4>, dim3(blocks), dim3(256), 0, 0, d, n4); }, 5), gb / 2);
    }
    rep("tiled U=4 blocks=4096 NT loads only", time_ms([&] { hipLaunchKernelGGL((copy_tiled_split<4, true, false>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("tiled U=4 blocks=4096 NT stores only", time_ms([&] { hipLaunchKernelGGL((copy_tiled_split<4, false, true>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("tiled U=4 blocks=4096 NT both", time_ms([&] { hipLaunchKernelGGL((copy_tiled_split<4, true, true>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 0/0", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 0, 0>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 0/2 (nt stores)", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 0, 2>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 2/0 (nt loads)", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 2, 0>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 2/2", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 2, 2>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 1/1 (sc0)", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 1, 1>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("buffer U=4 blocks=4096 aux 3/3 (sc0 nt)", time_ms([&] { hipLaunchKernelGGL((copy_buffer<4, 3, 3>), dim3(4096), dim3(256), 0, 0, s, d, n4); }, 5), gb);
    rep("write-only NT U=4 blocks=4096", time_ms([&] { hipLaunchKernelGGL((write_only_nt<4, true>), dim3(4096), dim3(256), 0, 0, d, n4); }, 5), gb / 2);
    rep("write-only    U=4 blocks=4096", time_ms([&] { hipLaunchKernelGGL(write_only<4>, dim3(4096), dim3(256), 0, 0, d, n4); }, 5), gb / 2);
    return 0;
}
