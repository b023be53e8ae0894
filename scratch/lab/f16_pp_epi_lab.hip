// Lab: where does a workgroup of the shipped filter GEMM (gemm_f16_pp_kernel<true>, fast.hip) spend its life?  The product source is compiled
// into this program with ISX_PP_STAMP defined: shader-clock stamps of waves 0 and 4 at the phase boundaries, one record per workgroup.
// build (objects of the library first: make -C instance-search_amd/csrc):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinstance-search_amd/csrc -Iinclude -o scratch/lab/f16_pp_epi_lab scratch/lab/f16_pp_epi_lab.hip \
//         instance-search_amd/csrc/{pool,region,cosine,select,rank,train,conv,expand,stream1x1,stem,api,comm}.o -ldl
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_pp_stamps = nullptr;
#define ISX_PP_STAMP(i) do { if (g_pp_stamps && (threadIdx.x & 255) == 0) \
    g_pp_stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#include "../../instance-search_amd/csrc/fast.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int64_t M = 10000, N = 24576;
    const int D = 2048;
    const double z = argc > 1 ? atof(argv[1]) : 2.57;            // threshold in standard deviations of a score: 2.57 flags ~15 % of the 32-column groups
    std::vector<_Float16> hq((size_t)M * D), hg((size_t)N * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = (_Float16)(rnd() * 16000.0f);
    for (auto& v : hg) v = (_Float16)(rnd() * 16000.0f);
    _Float16 *dq, *dg; float *c, *thr; uint8_t* gflag;
    const int ngrp = (int)((N + 31) / 32);
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2));
    CK(hipMalloc(&c, (size_t)M * N * 4)); CK(hipMalloc(&thr, M * 4)); CK(hipMalloc(&gflag, (size_t)M * ngrp));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    const float sigma = sqrtf((float)D) * 16000.0f * 16000.0f / 12.0f;
    std::vector<float> ht(M, (float)(z * sigma));
    CK(hipMemcpy(thr, ht.data(), M * 4, hipMemcpyHostToDevice));
    const int64_t btm = (M + 255) / 256, btn = (N + 255) / 256, nwg = btm * btn;
    unsigned long long* st;
    CK(hipMalloc(&st, nwg * 2 * 8 * 8)); CK(hipMemset(st, 0, nwg * 2 * 8 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {                          // untimed stamps off, then on
        unsigned long long* p = rep == 2 ? st : nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), &p, sizeof(p)));
        CK(hipEventRecord(a));
        launch_gemm_f16(dq, M, dg, N, D, c, N, thr, gflag, nullptr);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("filter GEMM %lld x %lld x %d (%lld tiles = %.1f rounds): %.3f ms  %.0f TF%s\n", (long long)M, (long long)N, D, (long long)nwg, nwg / 256.0, ms,
               2.0 * M * N * D / ms * 1e-9, rep == 2 ? "  (stamps on)" : "");
    }
    std::vector<uint8_t> hf((size_t)M * ngrp);
    CK(hipMemcpy(hf.data(), gflag, hf.size(), hipMemcpyDeviceToHost));
    size_t fl = 0; for (auto v : hf) fl += v != 0;
    printf("flagged groups: %.1f %%\n", 100.0 * fl / hf.size());
    std::vector<unsigned long long> h(nwg * 16);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[5] = {"start -> k loop (launch, descriptors, first DMAs landed)", "k loop", "k loop end -> thresholds in LDS", "filter + stores", "flags out"};
    for (int w = 0; w < 2; ++w) {
        double sum[5] = {0, 0, 0, 0, 0}, life = 0; size_t cnt = 0;
        for (int64_t g = 0; g < nwg; ++g) {
            const unsigned long long* r = &h[(g * 2 + w) * 8];
            if (!r[0] || !r[5]) continue;
            for (int i = 0; i < 5; ++i) sum[i] += (double)(r[i + 1] - r[i]);
            life += (double)(r[5] - r[0]); ++cnt;
        }
        printf("wave %d, mean over %zu workgroups (memtime ticks, 100 MHz = 10 ns): life %.0f\n", w * 4, cnt, life / cnt);
        for (int i = 0; i < 5; ++i) printf("   %-60s %8.0f  %5.1f %%\n", names[i], sum[i] / cnt, 100.0 * sum[i] / life);
    }
    // gap between consecutive workgroups on one CU slot is not visible here; the launch time / rounds gives it
    return 0;
}
