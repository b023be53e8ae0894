// Probe: HBM write rate of a (M x 256) fp32 matrix by store pattern.
//  A: MFMA C-layout stores: a wave-instruction writes 2 rows x 32 consecutive floats (4 B per lane), 16 per 32x32 tile; wave w owns column block w
//  B: row stores: a wave-instruction writes one whole 1-KB row (16 B per lane)
//  C: transposed-MFMA-layout stores: 16 B per lane = 4 consecutive floats of one row, 32 rows x 32 B per instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* y, long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const long ntiles = M / 64;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long p0 = t * 64;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long p = p0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    y[p * 256 + wave * 32 + l31] = (float)e;
                }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const long p = p0 + wave * 8 + r;
                *reinterpret_cast<float4*>(y + p * 256 + lane * 4) = make_float4(1.f, 2.f, 3.f, (float)r);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long p = p0 + i * 32 + l31;
                    *reinterpret_cast<float4*>(y + p * 256 + wave * 32 + 8 * g + 4 * half) = make_float4(1.f, 2.f, 3.f, (float)g);
                }
        }
    }
}
template <int MODE> static void run(const char* name, float* y, long M) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, y, M); CK(hipDeviceSynchronize());
    for (int grid : {256, 512, 1024}) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, y, M);
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-44s grid %4d: %.3f ms  %.0f GB/s\n", name, grid, ms, M * 1024.0 / ms / 1e6);
    }
}
int main() {
    const long M = 1024L * 56 * 56;
    float* y; CK(hipMalloc(&y, M * 1024));
    run<0>("A: MFMA C layout, 4 B/lane, 2 x 128 B", y, M);
    run<1>("B: whole rows, 16 B/lane, 1 KB contiguous", y, M);
    run<2>("C: transposed layout, 16 B/lane, 32 x 32 B", y, M);
    return 0;
}
