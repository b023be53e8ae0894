// Lab: fp16 MFMA GEMM variants for the filter pass (C = Qh . Gh^T, fp32 accumulate).
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/f16_gemm_lab scratch/lab/f16_gemm_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,$PWD/instance-search_amd/csrc
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

extern "C" int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, void* stream);

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ---------------- variant B: 256x256 tile, 512 threads (8 waves 2x4, wave tile 128x64), BK=64, 2 LDS stages ----
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_B = (BM + BN) * BK * 2;      // 64 KB
__device__ __forceinline__ int hswz(int row) { return (row >> 1) & 7; }

template <int ABL>
__global__ __launch_bounds__(512) void gemm_b(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging: 512 rows x 8 chunks = 4096 chunks / 512 threads = 8 per thread: rows r = j*64 + tid/8 (j<4: A rows, j>=4: B rows)
    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j & 3) * 64 + sr;                  // 0..255 within A or B
        const bool isb = j >= 4;
        int64_t gr = (isb ? n0 : m0) + row;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        gsrc[j] = (isb ? G : Q) + gr * D + sc * 8;
        ldst[j] = (isb ? BM * 128 : 0) + row * 128 + ((sc ^ hswz(row)) << 4);
    }
    float4 reg[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) reg[j] = (k0 + sc * 8 < D) ? *reinterpret_cast<const float4*>(gsrc[j] + k0) : make_float4(0, 0, 0, 0);
    };
    auto lstore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(st + ldst[j]) = reg[j];
    };
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = BM * 128 + rb * 128; b_sw[j] = hswz(rb); }

    const int nk = (D + BK - 1) / BK;
    gload(0);
    lstore(lds);
    __syncthreads();
    if (nk > 1) gload(BK);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = lds + (kt & 1) * STAGE_B;
        char* nxt = lds + ((kt + 1) & 1) * STAGE_B;
        if (kt + 1 < nk) {
            if (ABL < 2) lstore(nxt);
            if (kt + 2 < nk && ABL < 1) gload((kt + 2) * BK);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int c = 2 * s + half;
            half8 a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const half8*>(cur + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
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

template <int MODE>
__global__ __launch_bounds__(512) void gemm_g(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging: 512 rows x 8 chunks = 4096 chunks / 512 threads = 8 per thread: rows r = j*64 + tid/8 (j<4: A rows, j>=4: B rows)
    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j & 3) * 64 + sr;                  // 0..255 within A or B
        const bool isb = j >= 4;
        int64_t gr = (isb ? n0 : m0) + row;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        gsrc[j] = (isb ? G : Q) + gr * D + sc * 8;
        ldst[j] = (isb ? BM * 128 : 0) + row * 128 + ((sc ^ hswz(row)) << 4);
    }
    float4 reg[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) reg[j] = (k0 + sc * 8 < D) ? *reinterpret_cast<const float4*>(gsrc[j] + k0) : make_float4(0, 0, 0, 0);
    };
    auto lstore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(st + ldst[j]) = reg[j];
    };
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = BM * 128 + rb * 128; b_sw[j] = hswz(rb); }

    const int nk = (D + BK - 1) / BK;
    gload(0);
    lstore(lds);
    __syncthreads();
    if (nk > 1) gload(BK);
    const bool late = (MODE == 1) && (wave >= 4);          // second wave of each SIMD: stage in mid-tile
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = lds + (kt & 1) * STAGE_B;
        char* nxt = lds + ((kt + 1) & 1) * STAGE_B;
        if (!late && kt + 1 < nk) {
            lstore(nxt);
            if (kt + 2 < nk) gload((kt + 2) * BK);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            if (s == 2 && late && kt + 1 < nk) {
                lstore(nxt);
                if (kt + 2 < nk) gload((kt + 2) * BK);
            }
            const int c = 2 * s + half;
            half8 a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const half8*>(cur + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
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

template <int MODE>
static void launch_g(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipFuncSetAttribute((const void*)gemm_g<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B);
    hipLaunchKernelGGL(gemm_g<MODE>, dim3(tm * tn), dim3(512), 2 * STAGE_B, 0, Q, M, G, N, D, C, N, tm, tn);
}

template <int ABL>
static void launch_b(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm_b<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B); }
    hipLaunchKernelGGL(gemm_b<ABL>, dim3(tm * tn), dim3(512), 2 * STAGE_B, 0, Q, M, G, N, D, C, N, tm, tn);
}


// ---------------- variant C: as B, tiles brought in by LDS-DMA (global_load_lds_dwordx4), two static stages ----
template <int PRE>
__global__ __launch_bounds__(512) void gemm_c(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char S0[STAGE_B];
    __shared__ __attribute__((aligned(1024))) char S1[STAGE_B];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // DMA pieces: wave w fills image rows [w*64, w*64+64) (8 pieces of 8 rows); lane -> row = piece*8 + lane/8, slot = lane%8
    const _Float16* gsrc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = wave * 64 + p * 8 + (lane >> 3);    // 0..511: A rows then B rows
        const bool isb = row >= BM;
        const int rr = isb ? row - BM : row;
        int64_t gr = (isb ? n0 : m0) + rr;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        const int c = (lane & 7) ^ hswz(rr);
        gsrc[p] = (isb ? G : Q) + gr * D + c * 8;
    }
    auto dma = [&](char* st, int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[p] + k0),
                                             (__attribute__((address_space(3))) void*)(st + (wave * 64 + p * 8) * 128), 16, 0, 0);
    };
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = BM * 128 + rb * 128; b_sw[j] = hswz(rb); }

    auto compute = [&](const char* cur) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int c = 2 * s + half;
            half8 a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const half8*>(cur + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = D / BK;                      // D % 64 == 0 required
    dma(S0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 1 < nk) dma(S1, (kt + 1) * BK);
        compute(S0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) dma(S0, (kt + 2) * BK);
            compute(S1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
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

static void launch_c(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_c<0>, dim3(tm * tn), dim3(512), 0, 0, Q, M, G, N, D, C, N, tm, tn);
}

// ---------------- variant E: as D with asm loads and explicit vmcnt waits; as B, but two staging register sets, unconditional loads (D % 64 == 0), LDS stores
// interleaved with the MFMAs of the running k-tile ----
template <int SCHED>
__global__ __launch_bounds__(512) void gemm_e(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char S0[STAGE_B];
    __shared__ __attribute__((aligned(16))) char S1[STAGE_B];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j & 3) * 64 + sr;
        const bool isb = j >= 4;
        int64_t gr = (isb ? n0 : m0) + row;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        gsrc[j] = (isb ? G : Q) + gr * D + sc * 8;
        ldst[j] = (isb ? BM * 128 : 0) + row * 128 + ((sc ^ hswz(row)) << 4);
    }
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = BM * 128 + rb * 128; b_sw[j] = hswz(rb); }

    float4 rA[8], rB[8];
    // loads by inline asm: the compiler's waitcnt pass cannot count them, the waits below are explicit
    auto gload = [&](float4 (&reg)[8], int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(reg[j]) : "v"(gsrc[j] + k0) : "memory");
    };
    // compute one k-tile from cur while the staged registers go to nxt, two stores per k16 step
    auto step = [&](const char* cur, char* nxt, const float4 (&reg)[8]) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int c = 2 * s + half;
            half8 a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const half8*>(cur + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
                if (i == 1) *reinterpret_cast<float4*>(nxt + ldst[2 * s]) = reg[2 * s];
                if (i == 3) *reinterpret_cast<float4*>(nxt + ldst[2 * s + 1]) = reg[2 * s + 1];
            }
            if (SCHED) {
                // 6 ds_read, then (2 MFMA, ...) with the two ds_write after MFMA 4 and 8
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
    };

    const int nk = D / BK;
    gload(rA, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(S0 + ldst[j]) = rA[j];
    __syncthreads();
    gload(rA, BK);                                   // nk even and >= 2 (D % 128 == 0)
    for (int kt = 0; kt < nk; kt += 2) {
        // loads past the last tile re-read it (their stores land in a stage nobody reads again): straight-line
        // code, so the compiler can count outstanding loads exactly (vmcnt(8), not vmcnt(0))
        gload(rB, min(kt + 2, nk - 1) * BK);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // rA (the 8 older loads) has landed
        step(S0, S1, rA);
        __syncthreads();
        gload(rA, min(kt + 3, nk - 1) * BK);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        step(S1, S0, rB);
        __syncthreads();
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

template <int SCHED>
static void launch_e(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_e<SCHED>, dim3(tm * tn), dim3(512), 0, 0, Q, M, G, N, D, C, N, tm, tn);
}

// ---------------- variant D: as B, but two staging register sets, unconditional loads (D % 64 == 0), LDS stores
// interleaved with the MFMAs of the running k-tile ----
template <int SCHED>
__global__ __launch_bounds__(512) void gemm_d(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char S0[STAGE_B];
    __shared__ __attribute__((aligned(16))) char S1[STAGE_B];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j & 3) * 64 + sr;
        const bool isb = j >= 4;
        int64_t gr = (isb ? n0 : m0) + row;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;
        gsrc[j] = (isb ? G : Q) + gr * D + sc * 8;
        ldst[j] = (isb ? BM * 128 : 0) + row * 128 + ((sc ^ hswz(row)) << 4);
    }
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = BM * 128 + rb * 128; b_sw[j] = hswz(rb); }

    float4 rA[8], rB[8];
    auto gload = [&](float4 (&reg)[8], int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) reg[j] = *reinterpret_cast<const float4*>(gsrc[j] + k0);
    };
    // compute one k-tile from cur while the staged registers go to nxt, two stores per k16 step
    auto step = [&](const char* cur, char* nxt, const float4 (&reg)[8]) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int c = 2 * s + half;
            half8 a[4], bb[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const half8*>(cur + b_off[j] + ((c ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
                if (i == 1) *reinterpret_cast<float4*>(nxt + ldst[2 * s]) = reg[2 * s];
                if (i == 3) *reinterpret_cast<float4*>(nxt + ldst[2 * s + 1]) = reg[2 * s + 1];
            }
            if (SCHED) {
                // 6 ds_read, then (2 MFMA, ...) with the two ds_write after MFMA 4 and 8
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
    };

    const int nk = D / BK;
    gload(rA, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(S0 + ldst[j]) = rA[j];
    __syncthreads();
    gload(rA, BK);                                   // nk even and >= 2 (D % 128 == 0)
    for (int kt = 0; kt < nk; kt += 2) {
        // loads past the last tile re-read it (their stores land in a stage nobody reads again): straight-line
        // code, so the compiler can count outstanding loads exactly (vmcnt(8), not vmcnt(0))
        gload(rB, min(kt + 2, nk - 1) * BK);
        step(S0, S1, rA);
        __syncthreads();
        gload(rA, min(kt + 3, nk - 1) * BK);
        step(S1, S0, rB);
        __syncthreads();
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

template <int SCHED>
static void launch_d(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_d<SCHED>, dim3(tm * tn), dim3(512), 0, 0, Q, M, G, N, D, C, N, tm, tn);
}

// ---------------- variant F: A (queries) through LDS as in B; B (gallery) fragments straight from a PRE-TILED
// global image Gt[row group of 32][k16 step][lane][8 halfs] (one coalesced 1 KiB load per fragment), no LDS for B ----
constexpr int STAGE_A = BM * BK * 2;     // 32 KB
__global__ __launch_bounds__(512) void gemm_f(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ Gt, int64_t N, int D,
                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char S0[STAGE_A];
    __shared__ __attribute__((aligned(16))) char S1[STAGE_A];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[4];
    int ldst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 64 + sr;
        int64_t gr = m0 + row;
        gr = gr < M ? gr : M - 1;
        gsrc[j] = Q + gr * D + sc * 8;
        ldst[j] = row * 128 + ((sc ^ hswz(row)) << 4);
    }
    int a_off[4], a_sw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
    // B fragment source: row group (n0/32 + wn*2 + j), k16 step ks -> + ks * 512 halfs; lane * 8 halfs
    const int64_t ngroups = (N + 31) / 32;
    const _Float16* bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int64_t g = n0 / 32 + wn * 2 + j;
        g = g < ngroups ? g : ngroups - 1;
        bsrc[j] = Gt + g * (int64_t)32 * D + lane * 8;
    }
    float4 aS[4];
    half8 b0[4][2], b1[4][2];
    auto aload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(aS[j]) : "v"(gsrc[j] + k0) : "memory");
    };
    auto bload = [&](half8 (&bf)[4][2], int k0) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bf[s2][j]) : "v"(bsrc[j] + (int64_t)(k0 / 16 + s2) * 512) : "memory");
    };
    auto astore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(st + ldst[j]) = aS[j];
    };
    auto compute = [&](const char* cur, const half8 (&bf)[4][2]) {
#pragma unroll
        for (int s2 = 0; s2 < BK / 16; ++s2) {
            const int c = 2 * s2 + half;
            half8 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(cur + a_off[i] + ((c ^ a_sw[i]) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bf[s2][j], acc[i][j], 0, 0, 0);
        }
    };
    const int nk = D / BK;                       // even, >= 2
    aload(0);
    bload(b0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    astore(S0);
    __syncthreads();
    aload(BK);
    bload(b1, BK);
    for (int kt = 0; kt < nk; kt += 2) {
        // even: compute tile kt (S0, b0); A tile kt+1 (in aS) -> S1; fetch A tile kt+2 and B frags kt+2
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // aS and b0 landed (the 8 b1 loads may still fly)
        astore(S1);
        aload(min(kt + 2, nk - 1) * BK);
        compute(S0, b0);
        bload(b0, min(kt + 2, nk - 1) * BK);
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // aS landed (the 8 b0 loads may still fly)
        astore(S0);
        aload(min(kt + 3, nk - 1) * BK);
        compute(S1, b1);
        bload(b1, min(kt + 3, nk - 1) * BK);
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

static void launch_f(const _Float16* Q, int64_t M, const _Float16* Gt, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_f, dim3(tm * tn), dim3(512), 0, 0, Q, M, Gt, N, D, C, N, tm, tn);
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
    const int64_t M = argc > 1 ? atoll(argv[1]) : 10000, N = argc > 2 ? atoll(argv[2]) : 32768;
    const int D = argc > 3 ? atoi(argv[3]) : 2048;
    std::vector<_Float16> hq((size_t)M * D), hg((size_t)N * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = (_Float16)(rnd() * 0.05f);
    for (auto& v : hg) v = (_Float16)(rnd() * 0.05f);
    _Float16 *dq, *dg; float *c0, *c1;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * D;
    float t = time_ms([&] { isx_cosine_sim_f16(dq, M, dg, N, D, c0, nullptr); });
    printf("baseline 128x128      : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    t = time_ms([&] { launch_b<0>(dq, M, dg, N, D, c1); });
    CK(hipGetLastError());
    printf("B 256x256 2-stage     : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    // compare
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double mx = 0;
    for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
    printf("B vs baseline: %zu mismatches (max diff %.3g)\n", bad, mx);
    CK(hipMemset(c1, 0, (size_t)M * N * 4));
    t = time_ms([&] { launch_c(dq, M, dg, N, D, c1); });
    CK(hipGetLastError());
    printf("C 256x256 LDS-DMA     : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
    bad = 0; mx = 0;
    for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
    printf("C vs baseline: %zu mismatches (max diff %.3g)\n", bad, mx);
    for (int v = 0; v < 2; ++v) {
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        t = time_ms([&] { if (v) launch_g<1>(dq, M, dg, N, D, c1); else launch_g<0>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("G%d 256x256 staggered stores: %.3f ms  %.0f TF\n", v, t, fl / t * 1e-9);
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; mx = 0;
        for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
        printf("G%d vs baseline: %zu mismatches (max diff %.3g)\n", v, bad, mx);
    }
    if (0) {
        // pre-tiled gallery image
        const int64_t ng = (N + 31) / 32;
        std::vector<_Float16> ht((size_t)ng * 32 * D);
        for (int64_t g = 0; g < ng; ++g)
            for (int ks = 0; ks < D / 16; ++ks)
                for (int ln = 0; ln < 64; ++ln) {
                    int64_t row = g * 32 + (ln & 31); if (row >= N) row = N - 1;
                    const int k = ks * 16 + (ln >> 5) * 8;
                    for (int x = 0; x < 8; ++x) ht[((size_t)(g * (D / 16) + ks) * 64 + ln) * 8 + x] = hg[(size_t)row * D + k + x];
                }
        _Float16* dt; CK(hipMalloc(&dt, ht.size() * 2));
        CK(hipMemcpy(dt, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        t = time_ms([&] { launch_f(dq, M, dt, N, D, c1); });
        CK(hipGetLastError());
        printf("F 256x256 B direct     : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; mx = 0;
        for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
        printf("F vs baseline: %zu mismatches (max diff %.3g)\n", bad, mx);
    }
    for (int v = 1; v < 1; ++v) {
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        t = time_ms([&] { if (v) launch_e<1>(dq, M, dg, N, D, c1); else launch_e<0>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("E%d 256x256 asm loads  : %.3f ms  %.0f TF\n", v, t, fl / t * 1e-9);
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; mx = 0;
        for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
        printf("E%d vs baseline: %zu mismatches (max diff %.3g)\n", v, bad, mx);
    }
    for (int v = 0; v < 0; ++v) {
        CK(hipMemset(c1, 0, (size_t)M * N * 4));
        t = time_ms([&] { if (v) launch_d<1>(dq, M, dg, N, D, c1); else launch_d<0>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("D%d 256x256 interleaved: %.3f ms  %.0f TF\n", v, t, fl / t * 1e-9);
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        bad = 0; mx = 0;
        for (size_t i = 0; i < h0.size(); ++i) { if (h0[i] != h1[i]) { ++bad; mx = fmax(mx, fabs((double)h0[i] - h1[i])); } }
        printf("D%d vs baseline: %zu mismatches (max diff %.3g)\n", v, bad, mx);
    }
    t = time_ms([&] { launch_b<1>(dq, M, dg, N, D, c1); });
    printf("B abl1 (no global loads in loop): %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    t = time_ms([&] { launch_b<2>(dq, M, dg, N, D, c1); });
    printf("B abl2 (no loads, no LDS stores): %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
    return 0;
}
