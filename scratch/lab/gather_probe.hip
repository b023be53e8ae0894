// gather_probe.hip -- how fast can the chip gather random 8-KB rows, from a table that fits the Infinity Cache (82 MB: the 10 000 x 2048 query block)
// and from one that does not (1 GB: the 125 000 x 2048 gallery shard)?  Decides whether re-scoring "by gallery row" (queries streamed from the
// Infinity Cache) could beat re-scoring "by query" (gallery rows gathered from HBM: 7.1 TB/s today).
//   hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave per gathered row (2048 floats = 8 x 16 B per lane), RPW rows per wave, 4 waves per workgroup
template <int RPW>
__global__ __launch_bounds__(256) void gather_rows(const float4* __restrict__ tab, const int* __restrict__ idx, int64_t n, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (w + r >= n) break;
        const float4* row = tab + (int64_t)idx[w + r] * 512;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = row[j * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 12345.678f) out[0] = acc;          // never true: keeps the loads alive
}

static void run(int64_t rows, int64_t n) {
    float4* tab; int* idx; float* out;
    CK(hipMalloc(&tab, rows * 8192)); CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(tab, 0, rows * 8192));
    std::vector<int> h(n);
    unsigned s = 12345;
    for (int64_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 8) % rows); }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_rows<4>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, 0, tab, idx, n, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("table %6.0f MB, %lld gathered rows of 8 KB (%.1f GB): %.3f ms = %.2f TB/s\n", rows * 8192 / 1e6, (long long)n, n * 8192 / 1e9, ms, n * 8192 / ms / 1e9);
    hipFree(tab); hipFree(idx); hipFree(out);
}

int main() {
    run(10000, 1400000);       // queries: fits the 256 MB Infinity Cache
    run(30000, 1400000);       // 246 MB
    run(125000, 1400000);      // gallery shard: 1 GB, HBM
    return 0;
}
