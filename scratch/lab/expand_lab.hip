// expand_lab.hip -- where does a workgroup of conv3x3_expand_kernel spend its life?  The product kernel (csrc/expand_kernel.hpp) instantiated with
// STAMPS: wave 0 records the shader clock at the phase boundaries; the host prints the average length of each phase over the steady-state workgroups
// and the kernel time.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../instance-search_amd/csrc expand_lab.hip -o expand_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "expand_kernel.hpp"
void isx_set_error(const char*, ...) {}
using namespace isx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool DUAL>
static void run(int B, bool with_res) {
    const int H = 56, Cin = 64;
    Conv3x3Geom g; g.H = H; g.W = H; g.Cin = Cin; g.stride = 1; g.Ho = H; g.Wo = H;
    const int64_t M = (int64_t)B * H * H;
    float *x, *w2, *b2, *w3t, *b3, *res, *y; unsigned long long* st;
    CK(hipMalloc(&x, M * Cin * 4)); CK(hipMalloc(&w2, 64 * 9 * Cin * 4)); CK(hipMalloc(&b2, 256)); CK(hipMalloc(&w3t, 128 * 256 * 4)); CK(hipMalloc(&b3, 1024));
    CK(hipMalloc(&res, M * 256 * 4)); CK(hipMalloc(&y, M * 256 * 4));
    const int64_t nwg = (M + 63) / 64;
    CK(hipMalloc(&st, nwg * 64));
    CK(hipMemset(x, 0, M * Cin * 4)); CK(hipMemset(w2, 0, 64 * 9 * Cin * 4)); CK(hipMemset(b2, 0, 256)); CK(hipMemset(w3t, 0, 128 * 256 * 4)); CK(hipMemset(b3, 0, 1024));
    CK(hipMemset(res, 0, M * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float* rp = DUAL ? res /* x2: (M, 64) */ : (with_res ? res : nullptr);
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((conv3x3_expand_kernel<4, DUAL, true>), dim3((unsigned)nwg), dim3(256), 0, 0, x, M, w2, g, b2, w3t, b3, rp, 1, y, st);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), st, nwg * 64, hipMemcpyDeviceToHost));
    double ph[4] = {0, 0, 0, 0}; int64_t n = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int64_t w = 2048; w < nwg - 2048; ++w) { for (int p = 0; p < 4; ++p) ph[p] += (double)(h[w * 8 + p + 1] - h[w * 8 + p]); ++n; }
    for (int64_t w = 0; w < nwg; ++w) { if (h[w * 8] < tmin) tmin = h[w * 8]; if (h[w * 8 + 4] > tmax) tmax = h[w * 8 + 4]; }
    const double flop = 2.0 * M * (9.0 * Cin * 64 + (DUAL ? 128 : 64) * 256.0);
    printf("%s%s B=%d: %.3f ms = %.1f TFLOP/s | kernel span %.0f kcycles (=> %.2f GHz) | per workgroup, cycles: 3x3 loop %.0f, mid tile -> LDS %.0f, expansion loop %.0f, epilogue %.0f (sum %.0f)\n",
           DUAL ? "dual" : "plain", DUAL ? "" : (with_res ? " +res" : " no res"), B, ms, flop / ms / 1e9, (tmax - tmin) / 1e3, (tmax - tmin) / (ms * 1e6),
           ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, (ph[0] + ph[1] + ph[2] + ph[3]) / n);
    hipFree(x); hipFree(w2); hipFree(b2); hipFree(w3t); hipFree(b3); hipFree(res); hipFree(y); hipFree(st);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;
    run<false>(B, true);
    run<false>(B, false);
    run<true>(B, true);
    return 0;
}
