// fast.hip -- EXACT top-k retrieval with a half-precision MFMA filter.
//
// isx_cosine_topk (cosine.hip) computes every score with fp32 MFMAs: exact by construction, bounded
// by the 157 TFLOP/s fp32 matrix peak.  The fp16 matrix cores are 16x faster, and a top-k search does
// not need every score exactly -- only the scores of the candidates that can make the list:
//
//   1. operands rounded to fp16 (RNE), row norms kept                         rows_to_f16_kernel
//   2. approximate scores  S' = Qh . Gh^T  on v_mfma_f32_32x32x16_f16 (fp32 accumulate) with the
//      same fused filter epilogue / group select as the fp32 path -> the KL best APPROXIMATE
//      candidates of every query (KL = 256 >= k)                             cosine_gemm_f16_kernel
//   3. |S' - S| <= eps_i for every pair (eps_i = c * |q_i| * max_j |g_j|, bound below), hence every
//      member of the exact top-k has S' >= a_k - 2 eps_i where a_k is the k-th best approximate
//      score.  Candidates inside that window are re-scored EXACTLY (the k-ordered fp32 fma chain of
//      the oracle) and sorted by the canonical key                            rescore_kernel
//   4. a row whose window is not fully covered by its KL candidates (dense clusters of near-equal
//      scores) is recomputed by exhaustive exact search                      exhaustive_rows_kernel
//
// The result is bit-identical to isx_cosine_topk for ANY input (tests: random, clustered galleries
// that force step 4, adversarial orderings).  Reference call sites: the same as isx_cosine_topk
// (test/classif_finetune_test.py:82 + utils/metrics.py:10-13,33).
//
// Error bound (unit roundoff u16 = 2^-11 for fp16 RNE, u32 = 2^-24): each product carries relative
// error <= 2 u16 + u16^2, fp16 underflow adds <= 2^-24 absolute per element (2^-25 rounding of a
// subnormal times |other| <= 1 after scaling... bounded by D * 2^-24 * |q|_inf |g|_inf), the fp32
// accumulations of both S' and S add <= 2 * D * u32 * sum|q g|.  With sum |q_k g_k| <= |q| |g|:
//   eps_i = (2^-10 + 2^-20 + 2 D 2^-24) |q_i| gmax + D 2^-23 qinf_i ginf      (computed per query)
// Inputs must satisfy |x| < 6e4 (fp16 range); descriptors are unit vectors.
#include <hip/hip_fp16.h>

#include "isx_internal.hpp"

namespace isx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// ---- 1. fp32 -> fp16 rows (+ squared norm and max |x| per row) ---------------------------------
__global__ __launch_bounds__(256) void rows_to_f16_kernel(const float* __restrict__ x, int64_t B, int D, _Float16* __restrict__ h,
                                                          float* __restrict__ norm2, float* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* r = x + row * D;
    _Float16* o = h + row * D;
    float ss = 0.0f, mx = 0.0f;
    for (int j = lane; j < D; j += 64) {
        const float v = r[j];
        o[j] = (_Float16)v;                      // RNE
        ss += v * v;
        mx = fmaxf(mx, fabsf(v));
    }
    ss = wave_sum(ss);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    if (lane == 0) { norm2[row] = ss * 1.000001f; amax[row] = mx; }     // slight inflation: norm2 is an upper bound
}

// ---- 2. fp16 MFMA GEMM, 128x128 tile, BK = 64, same filter epilogue as the fp32 kernel ------------
constexpr int HBK = 64;                                  // halfs per k-tile: 128 B per row = 8 chunks of 16 B
constexpr int HTILE_B = 128 * HBK * 2;                   // 16 KB per operand tile

__device__ __forceinline__ int hswz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void h_load_tile(const _Float16* __restrict__ P, int64_t rows, int D, int64_t row0, int k0, float4 (&reg)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 256 + threadIdx.x;           // 1024 chunks: row = idx / 8, chunk = idx % 8
        int64_t r = row0 + (idx >> 3);
        r = r < rows ? r : rows - 1;
        const int k = k0 + ((idx & 7) << 3);
        reg[j] = (k < D) ? *reinterpret_cast<const float4*>(P + r * D + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ void h_store_tile(char* __restrict__ T, const float4 (&reg)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 256 + threadIdx.x;
        const int r = idx >> 3, c = idx & 7;
        *reinterpret_cast<float4*>(T + r * 128 + ((c ^ hswz(r)) << 4)) = reg[j];
    }
}

template <bool FILTER>
__global__ __launch_bounds__(256) void cosine_gemm_f16_kernel(const _Float16* __restrict__ Q, int64_t M,
                                                              const _Float16* __restrict__ G, int64_t N, int D,
                                                              float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n,
                                                              const float* __restrict__ thr, uint8_t* __restrict__ gflag, int ngrp) {
    __shared__ __attribute__((aligned(16))) char lds[2 * HTILE_B];
    char* As = lds;
    char* Bs = lds + HTILE_B;
    // XCD-aware bijective remap + 16-wide n groups (as the fp32 kernel)
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    const int per_group = 16 * tiles_m, gid = wg / per_group, first_n = gid * 16;
    const int gsz = min(16, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * 128, n0 = (int64_t)(first_n + within % gsz) * 128;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // per-lane operand rows and their swizzle terms
    int a_off[2], b_off[2], a_sw[2], b_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        a_off[i] = ra * 128; a_sw[i] = hswz(ra);
        b_off[i] = rb * 128; b_sw[i] = hswz(rb);
    }

    float4 ra4[4], rb4[4];
    const int nk = (D + HBK - 1) / HBK;
    h_load_tile(Q, M, D, m0, 0, ra4);
    h_load_tile(G, N, D, n0, 0, rb4);
    h_store_tile(As, ra4);
    h_store_tile(Bs, rb4);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            h_load_tile(Q, M, D, m0, (kt + 1) * HBK, ra4);
            h_load_tile(G, N, D, n0, (kt + 1) * HBK, rb4);
        }
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            const int c = 2 * s + half;                          // this lane's 16-B chunk (8 consecutive k)
            half8 a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const half8*>(As + a_off[i] + ((c ^ a_sw[i]) << 4));
                bb[i] = *reinterpret_cast<const half8*>(Bs + b_off[i] + ((c ^ b_sw[i]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            h_store_tile(As, ra4);
            h_store_tile(Bs, rb4);
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t ng = n0 + wn * 64 + j * 32;
            const int64_t n = ng + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const float v = acc[i][j][e];
                if (FILTER) {
                    const bool row_ok = (m < M);
                    const float t = row_ok ? thr[m] : INFINITY;
                    const unsigned long long qm = __ballot(n < N && v >= t);
                    const bool qq = ((half ? (qm >> 32) : qm) & 0xFFFFFFFFull) != 0ull;
                    if (row_ok && ng < N) {
                        if (l31 == 0) gflag[m * ngrp + (ng >> 5)] = qq ? 1 : 0;
                        if (qq && n < N) C[m * ldc + n] = v;
                    }
                } else {
                    if (m < M && n < N) C[m * ldc + n] = v;
                }
            }
        }
    }
}

static int launch_gemm_f16(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C, int64_t ldc, const float* thr,
                           uint8_t* gflag, hipStream_t st) {
    if (M == 0 || N == 0) return ISX_OK;
    const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128;
    if (tm * tn >= (1ll << 31)) { isx_set_error("f16 gemm: too many tiles"); return ISX_ERR_ARG; }
    const dim3 grid((unsigned)(tm * tn)), block(256);
    const int ngrp = (int)((N + 31) / 32);
    if (gflag) hipLaunchKernelGGL(cosine_gemm_f16_kernel<true>, grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)tm, (int)tn, thr, gflag, ngrp);
    else hipLaunchKernelGGL(cosine_gemm_f16_kernel<false>, grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)tm, (int)tn, thr, gflag, ngrp);
    ISX_CHECK_LAUNCH("cosine_gemm_f16");
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

// fp32 rows -> fp16 rows (RNE) + per-row squared norm (upper bound) and max |x|.  h: (B,D) fp16 (2 B/elem).
ISX_API int isx_rows_to_f16(const float* x, int64_t B, int D, void* h, float* norm2, float* amax, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D > 0 && B < (1ll << 33), "isx_rows_to_f16: bad shape B=%lld D=%d", (long long)B, D);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(x && h && norm2 && amax, "isx_rows_to_f16: null pointer");
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, B, D, (_Float16*)h, norm2, amax);
    ISX_CHECK_LAUNCH("isx_rows_to_f16");
    return ISX_OK;
}

// Approximate similarity matrix from fp16 operands (fp32 accumulate): building block / diagnostic of the
// fast path.  Qh: (M,D) fp16, Gh: (N,D) fp16, D % 8 == 0, 16-B aligned.
ISX_API int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0 && D % 8 == 0, "isx_cosine_sim_f16: bad shape M=%lld N=%lld D=%d (D %% 8 == 0 required)", (long long)M, (long long)N, D);
    ISX_REQUIRE((Qh && Gh && sim) || M * N == 0, "isx_cosine_sim_f16: null pointer");
    ISX_REQUIRE((((uintptr_t)Qh | (uintptr_t)Gh) % 16) == 0, "isx_cosine_sim_f16: operands must be 16-B aligned");
    return launch_gemm_f16((const _Float16*)Qh, M, (const _Float16*)Gh, N, D, sim, N, nullptr, nullptr, (hipStream_t)stream);
}
