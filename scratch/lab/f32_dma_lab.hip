// Lab: fp32 MFMA GEMM (C = Q . G^T, k-ordered fma chain), 128x128 tile / 256 threads / BK = 16 / two LDS stages filled by LDS-DMA.
// Against the shipped kernel (register staging + transposed ds_write_b32 + ds_read_b32: ~3 non-MFMA instructions per MFMA) this
// loop has no staging registers, no ds_write and no address arithmetic: operands come from a row-major [row][4 x 16 B] XOR-swizzled
// image with ds_read_b128 (lanes 0-31: chunk 2p, lanes 32-63: chunk 2p + 1 of the same rows) and two v_permlane32_swap per read
// turn the four registers into the operands of four consecutive v_mfma_f32_32x32x2_f32 steps (k order untouched: bit-exact).
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/f32_dma_lab scratch/lab/f32_dma_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,$PWD/instance-search_amd/csrc
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#include <cmath>

extern "C" int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, void* stream);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int ROW_B = BK * 4;                 // 64 B per image row
template <int TM, int TN> struct Cfg {
    static constexpr int BM = 64 * TM, BN = 64 * TN, ROWS = BM + BN, STAGE_B = ROWS * ROW_B, NDMA = ROWS / 16 / 4;   // DMA instructions per wave and k-tile
};

template <int TM, int TN>
__global__ __launch_bounds__(256, TM * TN == 4 ? 4 : 5) void gemm32_dma(const float* __restrict__ Q, int64_t M, const float* __restrict__ G, int64_t N, int D,
                                                  float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    using K = Cfg<TM, TN>;
    __shared__ __attribute__((aligned(1024))) char lds[2 * K::STAGE_B];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 16;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * K::BM, n0 = (int64_t)(first_n + within % gsz) * K::BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // DMA: instruction x of this wave fills image rows (wave * NDMA + x) * 16 + lane / 4; slot lane % 4 of a row holds chunk slot ^ swz(row)
    const float* gsrc[K::NDMA];
#pragma unroll
    for (int x = 0; x < K::NDMA; ++x) {
        const int r = (wave * K::NDMA + x) * 16 + (lane >> 2);           // image row: A rows then B rows
        const bool isb = r >= K::BM;
        int64_t gr = isb ? n0 + (r - K::BM) : m0 + r;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        gsrc[x] = (isb ? G : Q) + gr * D + (((lane & 3) ^ ((r >> 2) & 3)) << 2);
    }
    const int T = D / BK;
    auto issue = [&](int t, int stage) {
#pragma unroll
        for (int x = 0; x < K::NDMA; ++x)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[x] + t * BK),
                                             (__attribute__((address_space(3))) void*)(lds + stage * K::STAGE_B + (wave * K::NDMA + x) * 1024), 16, 0, 0);
    };
    // operand reads: rows l31 of a 32-row block; lanes 0-31 take chunk 2p, lanes 32-63 chunk 2p + 1
    const int sw = (l31 >> 2) & 3;
    const char* a_ad[2];
    const char* b_ad[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        a_ad[p] = lds + (wm * 32 * TM + l31) * ROW_B + (((2 * p + half) ^ sw) << 4);
        b_ad[p] = lds + (K::BM + wn * 32 * TN + l31) * ROW_B + (((2 * p + half) ^ sw) << 4);
    }
    auto swap2 = [&](f32x4& v) {
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        // now: x = step 4p, z = step 4p + 1, y = step 4p + 2, w = step 4p + 3   (steps of two k each; chunk pair = 8 k)
    };
    auto body = [&](auto SC, int t) {
        constexpr int so = decltype(SC)::value * K::STAGE_B;
        if (t + 1 < T) issue(t + 1, decltype(SC)::value ^ 1);
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[p][i] = *reinterpret_cast<const f32x4*>(a_ad[p] + so + i * 32 * ROW_B);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[p][j] = *reinterpret_cast<const f32x4*>(b_ad[p] + so + j * 32 * ROW_B);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) swap2(af[p][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) swap2(bf[p][j]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = s == 0 ? af[p][i].x : s == 1 ? af[p][i].z : s == 2 ? af[p][i].y : af[p][i].w;
                        const float bb = s == 0 ? bf[p][j].x : s == 1 ? bf[p][j].z : s == 2 ? bf[p][j].y : bf[p][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[i][j], 0, 0, 0);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int t = 0;
    for (; t + 1 < T; t += 2) {
        body(std::integral_constant<int, 0>{}, t);
        body(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < T) body(std::integral_constant<int, 0>{}, t);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n0 + wn * 32 * TN + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M && n < N) C[m * ldc + n] = acc[i][j][e];
            }
        }
}

template <int TM, int TN>
static void launch(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + 64 * TM - 1) / (64 * TM)), tn = (int)((N + 64 * TN - 1) / (64 * TN));
    hipLaunchKernelGGL((gemm32_dma<TM, TN>), dim3(tm * tn), dim3(256), 0, 0, Q, M, G, N, D, C, N, tm, tn);
}

template <class F>
static float time_ms(F f, int it) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < it; ++i) f();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = fminf(best, ms / it);
    }
    return best;
}

static void one(int64_t M, int64_t N, int D, bool check) {
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
    const int it = fl > 5e11 ? 3 : 20;
    printf("---- %lld x %lld x %d\n", (long long)M, (long long)N, D);
    for (int rep = 0; rep < 2; ++rep) {
        float t = time_ms([&] { isx_cosine_sim(dq, M, dg, N, D, c0, nullptr); }, it);
        printf("shipped (auto tile)  : %.3f ms  %.1f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch<2, 2>(dq, M, dg, N, D, c1); }, it);
        printf("DMA 128x128          : %.3f ms  %.1f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch<2, 1>(dq, M, dg, N, D, c1); }, it);
        printf("DMA 128x64           : %.3f ms  %.1f TF\n", t, fl / t * 1e-9);
        CK(hipGetLastError());
        fflush(stdout);
    }
    if (check) {
        std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
        CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
            if (v == 0) launch<2, 2>(dq, M, dg, N, D, c1); else launch<2, 1>(dq, M, dg, N, D, c1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < h0.size(); ++i) if (memcmp(&h0[i], &h1[i], 4) != 0) ++bad;
            printf("variant %d vs shipped: %zu of %zu scores differ (bitwise)\n", v, bad, h0.size());
        }
    }
    hipFree(dq); hipFree(dg); hipFree(c0); hipFree(c1);
}

int main(int argc, char** argv) {
    one(1000, 3000, 512, true);
    one(10000, 32768, 2048, false);
    one(1024, 10000, 2048, false);
    one(200704, 256, 1024, false);     // 1x1 convolution shapes of the trunk (pixels x Cout x Cin)
    one(200704, 512, 1024, false);
    one(802816, 128, 512, false);
    one(50176, 512, 2048, false);
    return 0;
}
