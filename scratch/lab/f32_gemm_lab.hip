// Lab: fp32 MFMA GEMM with a 256x256 tile (C = Q . G^T, k-ordered fma chain), against libisx's 128x128 kernel.
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/f32_gemm_lab scratch/lab/f32_gemm_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,'$ORIGIN/../../instance-search_amd/csrc'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

extern "C" int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, void* stream);
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 16, LDA = BM + 1, LDB = BN + 1;
constexpr int STAGE_F = BK * (LDA + LDB);        // floats per stage

template <int STAGES>
__global__ __launch_bounds__(512) void gemm256(const float* __restrict__ Q, int64_t M, const float* __restrict__ G, int64_t N, int D,
                                               float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ float lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging: 256 rows x 4 chunks (16 floats) per operand = 1024 chunks / 512 threads = 2 per operand
    const float* asrc[2]; const float* bsrc[2]; int srow[2];
    const int sc = tid & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = j * 128 + (tid >> 2);
        srow[j] = row;
        int64_t ra = m0 + row; ra = ra < M ? ra : M - 1;
        int64_t rb = n0 + row; rb = rb < N ? rb : N - 1;
        asrc[j] = Q + ra * D + sc * 4;
        bsrc[j] = G + rb * D + sc * 4;
    }
    float4 ra4[2], rb4[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { ra4[j] = *reinterpret_cast<const float4*>(asrc[j] + k0); rb4[j] = *reinterpret_cast<const float4*>(bsrc[j] + k0); }
    };
    auto lstore = [&](float* st) {
        float* As = st; float* Bs = st + BK * LDA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = srow[j], k = sc * 4;
            As[(k + 0) * LDA + r] = ra4[j].x; As[(k + 1) * LDA + r] = ra4[j].y; As[(k + 2) * LDA + r] = ra4[j].z; As[(k + 3) * LDA + r] = ra4[j].w;
            Bs[(k + 0) * LDB + r] = rb4[j].x; Bs[(k + 1) * LDB + r] = rb4[j].y; Bs[(k + 2) * LDB + r] = rb4[j].z; Bs[(k + 3) * LDB + r] = rb4[j].w;
        }
    };
    auto compute = [&](const float* st) {
        const float* a_base = st + half * LDA + wm * 128 + l31;
        const float* b_base = st + BK * LDA + half * LDB + wn * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = a_base[(2 * kk) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = b_base[(2 * kk) * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = D / BK;                      // D % 16 == 0
    gload(0);
    lstore(lds);
    __syncthreads();
    if (nk > 1) gload(BK);
    for (int kt = 0; kt < nk; ++kt) {
        float* cur = lds + (STAGES == 2 ? (kt & 1) * STAGE_F : 0);
        float* nxt = lds + (STAGES == 2 ? ((kt + 1) & 1) * STAGE_F : 0);
        if (STAGES == 2) {
            if (kt + 1 < nk) { lstore(nxt); if (kt + 2 < nk) gload((kt + 2) * BK); }
            compute(cur);
            __syncthreads();
        } else {
            compute(cur);
            __syncthreads();
            if (kt + 1 < nk) { lstore(nxt); if (kt + 2 < nk) gload((kt + 2) * BK); __syncthreads(); }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M && n < N) C[m * ldc + n] = acc[i][j][e];
            }
        }
}

template <int STAGES>
static void launch256(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    const size_t sh = (size_t)STAGES * STAGE_F * 4;
    hipFuncSetAttribute((const void*)gemm256<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(gemm256<STAGES>, dim3(tm * tn), dim3(512), sh, 0, Q, M, G, N, D, C, N, tm, tn);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class F>
static float time_ms(F f, int it = 3) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); CK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 10000, N = argc > 2 ? atoll(argv[2]) : 32768;
    const int D = argc > 3 ? atoi(argv[3]) : 2048;
    std::vector<float> hq((size_t)M * D), hg((size_t)N * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = rnd() * 0.05f;
    for (auto& v : hg) v = rnd() * 0.05f;
    float *dq, *dg, *c0, *c1;
    CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * D;
    float t = time_ms([&] { isx_cosine_sim(dq, M, dg, N, D, c0, nullptr); });
    printf("libisx 128x128 BK16    : %.3f ms  %.1f TF\n", t, fl / t * 1e-9);
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
    for (int v = 0; v < 2; ++v) {
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        t = time_ms([&] { if (v) launch256<2>(dq, M, dg, N, D, c1); else launch256<1>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("256x256 %d stage(s)      : %.3f ms  %.1f TF\n", v + 1, t, fl / t * 1e-9);
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += (h0[i] != h1[i]);
        printf("   mismatches vs libisx: %zu\n", bad);
    }
    return 0;
}
