// Lab (option NOT built into libisx): fused split-precision GEMM  C = A . W^T  with fp32 A split on the fly into (hi, lo) fp16,
// W pre-split, three fp16 MFMAs per fragment pair (hh into one accumulator, hl + lh into a second one scaled by 2^-11 at the end).
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/split_gemm_lab scratch/lab/split_gemm_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,'$ORIGIN/../../instance-search_amd/csrc'
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

extern "C" int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, void* stream);
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 64;                        // k per tile: 128-B fp16 rows = 8 chunks of 16 B
constexpr int PART = 128 * BK * 2;            // 16 KB per operand part
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(256) void split_gemm(const float* __restrict__ A, int64_t M, const _Float16* __restrict__ Wh,
                                                  const _Float16* __restrict__ Wl, int64_t N, int K, float* __restrict__ C, int tiles_m,
                                                  int tiles_n) {
    __shared__ __attribute__((aligned(16))) char lds[4 * PART];      // A_hi | A_lo | B_hi | B_lo
    char* Ahi = lds; char* Alo = lds + PART; char* Bhi = lds + 2 * PART; char* Blo = lds + 3 * PART;
    const int wg = blockIdx.x;
    const int tile_n = wg % tiles_n, tile_m = wg / tiles_n;
    const int64_t m0 = (int64_t)tile_m * 128, n0 = (int64_t)tile_n * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[2][2], acx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.0f; acx[i][j][e] = 0.0f; }

    // staging roles: 1024 (row, chunk) slots per operand, 4 per thread
    const float* asrc[4]; const _Float16* bhsrc[4]; const _Float16* blsrc[4]; int dst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 256 + tid, row = idx >> 3, c = idx & 7;
        int64_t ra = m0 + row; ra = ra < M ? ra : M - 1;
        int64_t rb = n0 + row; rb = rb < N ? rb : N - 1;
        asrc[j] = A + ra * K + c * 8;
        bhsrc[j] = Wh + rb * K + c * 8;
        blsrc[j] = Wl + rb * K + c * 8;
        dst[j] = row * 128 + ((c ^ swz(row)) << 4);
    }
    float4 a0[4], a1[4], bh[4], bl[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] = *reinterpret_cast<const float4*>(asrc[j] + k0);
            a1[j] = *reinterpret_cast<const float4*>(asrc[j] + k0 + 4);
            bh[j] = *reinterpret_cast<const float4*>(bhsrc[j] + k0);
            bl[j] = *reinterpret_cast<const float4*>(blsrc[j] + k0);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v[8] = {a0[j].x, a0[j].y, a0[j].z, a0[j].w, a1[j].x, a1[j].y, a1[j].z, a1[j].w};
            half8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                hi[q] = (_Float16)v[q];
                lo[q] = (_Float16)((v[q] - (float)hi[q]) * 2048.0f);
            }
            *reinterpret_cast<half8*>(Ahi + dst[j]) = hi;
            *reinterpret_cast<half8*>(Alo + dst[j]) = lo;
            *reinterpret_cast<float4*>(Bhi + dst[j]) = bh[j];
            *reinterpret_cast<float4*>(Blo + dst[j]) = bl[j];
        }
    };
    int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        a_off[i] = ra * 128; a_sw[i] = swz(ra);
        b_off[i] = rb * 128; b_sw[i] = swz(rb);
    }
    const int nk = K / BK;
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload((kt + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int c = 2 * s + half;
            half8 ah[2], al[2], bhh[2], bll[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(Ahi + a_off[i] + ((c ^ a_sw[i]) << 4));
                al[i] = *reinterpret_cast<const half8*>(Alo + a_off[i] + ((c ^ a_sw[i]) << 4));
                bhh[i] = *reinterpret_cast<const half8*>(Bhi + b_off[i] + ((c ^ b_sw[i]) << 4));
                bll[i] = *reinterpret_cast<const half8*>(Blo + b_off[i] + ((c ^ b_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bhh[j], acc[i][j], 0, 0, 0);
                    acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bll[j], acx[i][j], 0, 0, 0);
                    acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bhh[j], acx[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) { lstore(); __syncthreads(); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M && n < N) C[m * N + n] = acc[i][j][e] + acx[i][j][e] * (1.0f / 2048.0f);
            }
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class F>
static float time_ms(F f, int it = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); CK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 200704, N = argc > 2 ? atoll(argv[2]) : 512;
    const int K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : ha) { float x = rnd() * 4.0f; v = x > 0 ? x : 0.0f; }             // post-ReLU like
    for (auto& v : hw) v = rnd() * 0.1f;
    std::vector<_Float16> hwh(hw.size()), hwl(hw.size());
    for (size_t i = 0; i < hw.size(); ++i) { hwh[i] = (_Float16)hw[i]; hwl[i] = (_Float16)((hw[i] - (float)hwh[i]) * 2048.0f); }
    float *da, *dw, *c0, *c1; _Float16 *dwh, *dwl;
    CK(hipMalloc(&da, ha.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dwh, hw.size() * 2)); CK(hipMalloc(&dwl, hw.size() * 2));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwh, hwh.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwl, hwl.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * K;
    const int tm = (int)((M + 127) / 128), tn = (int)((N + 127) / 128);
    float t = time_ms([&] { isx_cosine_sim(da, M, dw, N, K, c0, nullptr); });
    printf("fp32 MFMA (libisx)        : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    t = time_ms([&] { hipLaunchKernelGGL(split_gemm, dim3(tm * tn), dim3(256), 0, 0, da, M, dwh, dwl, N, K, c1, tm, tn); });
    CK(hipGetLastError());
    printf("fused split fp16 x3       : %.3f ms  %.0f TF fp32-equivalent\n", t, fl / t * 1e-9);
    std::vector<float> h0((size_t)1024 * N), h1((size_t)1024 * N);
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, mean = 0;
    for (int m = 0; m < 1024; m += 37)
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)m * K + k] * (double)hw[(size_t)n * K + k];
            e0 = fmax(e0, fabs(h0[(size_t)m * N + n] - ref)); e1 = fmax(e1, fabs(h1[(size_t)m * N + n] - ref)); mean += fabs(ref);
        }
    mean /= (double)((1024 + 36) / 37) * N;
    printf("max |err| / mean|C| vs fp64: fp32 chain %.2e   split %.2e\n", e0 / mean, e1 / mean);
    return 0;
}
