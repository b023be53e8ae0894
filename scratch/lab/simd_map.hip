// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: WAVE_ID[3:0], SIMD_ID[5:4], CU_ID[11:8] ...)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    __shared__ char lds[131072];
    lds[threadIdx.x] = 1;
    __syncthreads();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    // keep the CU busy a little so that consecutive workgroups do not simply reuse an idle CU
    float x = lds[threadIdx.x];
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 123.f) out[0] = 0;
}
int main() {
    unsigned* d; const int nb = 1024;
    hipMalloc(&d, nb * 8 * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
    int paired = 0, hist[4][8] = {};
    for (int b = 0; b < nb; ++b) {
        bool ok = true;
        for (int w = 0; w < 4; ++w) ok &= (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
        paired += ok;
        for (int w = 0; w < 8; ++w) hist[(h[b * 8 + w] >> 4) & 3][w]++;
        if (b < 6 || b == 300 || b == 700) { printf("block %4d simd of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
    }
    printf("blocks with wave w and w+4 on one SIMD: %d of %d\n", paired, nb);
    for (int s = 0; s < 4; ++s) { printf("SIMD %d:", s); for (int w = 0; w < 8; ++w) printf(" %4d", hist[s][w]); printf("\n"); }
    return 0;
}
