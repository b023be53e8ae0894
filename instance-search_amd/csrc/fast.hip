// fast.hip -- EXACT top-k retrieval with a half-precision MFMA filter.
//
// isx_cosine_topk (cosine.hip) computes every score with fp32 MFMAs: exact by construction, bounded
// by the 157 TFLOP/s fp32 matrix peak.  The fp16 matrix cores are 16x faster, and a top-k search does
// not need every score exactly -- only the scores of the candidates that can make the list:
//
//   1. operands scaled by a power of two (exact) and rounded to fp16 (RNE), row norms kept
//                                                                   amax_kernel, rows_to_f16_kernel
//   2. approximate scores  S' = Qh . Gh^T  on v_mfma_f32_32x32x16_f16 (fp32 accumulate) with the
//      same fused filter epilogue / group select as the fp32 path (run_topk_chunks, cosine.hip) ->
//      the KL = min(256, 2k + 32) best APPROXIMATE candidates of every query
//                                                 cosine_gemm_f16_big_kernel / cosine_gemm_f16_kernel
//   3. |S' - S| <= eps_i for every pair (bound below), hence every
//      member of the exact top-k has S' >= a_k - 2 eps_i where a_k is the k-th best approximate
//      score.  Candidates inside that window are re-scored EXACTLY (the k-ordered fp32 fma chain of
//      the oracle) and sorted by the canonical key                            rescore_kernel
//   4. rows whose window is not fully covered by their KL candidates (dense clusters of near-equal
//      scores) are compacted on the device and searched exactly by the fp32 pipeline (the same
//      run_topk_chunks with m_active / row_map: launch sizes fixed, idle tiles exit at once)
//                                                          compact_rows_kernel, gather_rows_kernel
//
// The result is bit-identical to isx_cosine_topk for ANY input (tests/test_gpu_fast.py: random, clustered
// galleries that force step 4, adversarial orderings, magnitudes, non-finite values, 40 random shapes).  Reference call sites: the same as isx_cosine_topk
// (test/classif_finetune_test.py:82 + utils/metrics.py:10-13,33).
//
// Error bound (round 3: from what the conversion actually lost, not from the worst case of the format).  u32 = 2^-24.  Operands are
// first multiplied by a power of two (exact) that brings the largest |x| of the matrix into [2^13, 2^14): scores scale by the exact
// factor sq*sg.  The conversion stores h = RNE_fp16(x), or 0 where that would be an fp16 subnormal, and records per row the norm of the
// loss r = |x - h| (differences exact in fp32).  With q~, g~ the stored rows:  q~.g~ - q.g = (q~ - q).g~ + q.(g~ - g), hence
//   operand rounding                |q~.g~ - q.g| <= rq (|g| + rg) + |q| rg          (Cauchy-Schwarz; rq, rg = the recorded losses)
//   fp32 accumulation, exact chain  (D-1) u32 sum|q g|,  MFMA chain (any order, truncation-safe) 2 D u32 sum|q~ g~|
//   eps_i = rq_i (|g|max + rg_max) + |q_i| rg_max + 3 D 2^-24 (|q_i| + rq_i)(|g|max + rg_max)        (inflated by 1 %)
// For unit rows at D = 2048 this is ~7.6e-4 against 1.34e-3 for the format's worst case (2^-10 |q||g|): the window of candidates that
// must be re-scored shrinks from ~38 to ~21 beyond k per query.
// Matrices whose largest |x| is outside [2^-20, 2^15], or not finite, take the exact fp32 path.
#include <hip/hip_fp16.h>
#include <type_traits>

#include "isx_internal.hpp"

namespace isx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#ifndef ISX_PP_STAMP            // scratch/lab/f16_pp_epi_lab.hip defines it: shader-clock stamps at the phase boundaries of gemm_f16_pp_kernel
#define ISX_PP_STAMP(i)
#endif

// ---- 1. fp32 -> fp16 rows ------------------------------------------------------------------------
// stats words are bit patterns of non-negative floats (unsigned order == float order) accumulated with
// atomicMax: order independent, deterministic.  stats[0] = max squared row norm (upper bound), stats[1] = max |x|.
__device__ __forceinline__ float nanmax(float a, float b) { return (b > a || b != b) ? b : a; }     // NaN-propagating

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ stats) {
    float mx = 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if ((((uintptr_t)x) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            mx = nanmax(nanmax(mx, fabsf(v.x)), nanmax(fabsf(v.y), nanmax(fabsf(v.z), fabsf(v.w))));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth) mx = nanmax(mx, fabsf(x[i]));
    } else {
        for (int64_t i = tid; i < n; i += nth) mx = nanmax(mx, fabsf(x[i]));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mx = nanmax(mx, __shfl_xor(mx, s, 64));
    if (!(mx >= 0.0f)) mx = INFINITY;                     // NaN -> unusable
    // one atomic per workgroup (16 k same-address atomics from one per wave cost 0.2 ms on their own)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&stats[1], __float_as_uint(nanmax(nanmax(wmax[0], wmax[1]), nanmax(wmax[2], wmax[3]))));
}

__device__ __forceinline__ bool f16_usable(float amax) { return amax >= 9.5367431640625e-7f && amax <= 32768.0f; }   // [2^-20, 2^15]
__device__ __forceinline__ float f16_scale(float amax) {
    return f16_usable(amax) ? ldexpf(1.0f, 13 - ilogbf(amax)) : 1.0f;
}

// stats (optional): scale from stats[1] (must be final: amax_kernel ran before), max norm2 into stats[0].
// In the scaled conversion of the search (stats given) fp16 results below the normal range are stored as ZERO (whether the matrix cores
// round fp16 subnormals gradually or flush them then does not matter), and res2 (optional) receives the squared norm of what the row LOST: sum (x - h / scale)^2, the difference taken in
// fp32 without rounding (x * scale and h lie within half an fp16 ulp of each other).  stats[2] = largest res2 (with want_norm_max).
__device__ __forceinline__ _Float16 to_f16_normal(float vs, float& lost, bool flush) {
    asm volatile("" : "+v"(vs));                         // the product as ONE fp32 value: left to itself hipcc converts with v_fma_mixlo_f16(x, scale, +0), and -0 + +0 = +0
    _Float16 hv = (_Float16)vs;                          // conversion RNE
    float hf = (float)hv;
    if (flush && fabsf(hf) < 6.103515625e-05f) { hv = (_Float16)0.0f; hf = 0.0f; }
    lost = vs - hf;
    return hv;
}

__global__ __launch_bounds__(256) void rows_to_f16_kernel(const float* __restrict__ x, int64_t B, int D, _Float16* __restrict__ h,
                                                          float* __restrict__ norm2, float* __restrict__ amax,
                                                          unsigned* __restrict__ stats, int want_norm_max, float* __restrict__ res2) {
    const int lane = threadIdx.x & 63;
    __shared__ float wn2[4], wr2[4];
    const float scale = stats ? f16_scale(__uint_as_float(stats[1])) : 1.0f;
    const float inv_s2 = 1.0f / (scale * scale);          // power of two: exact
    const bool flush = stats != nullptr;                  // the plain building block (isx_rows_to_f16) keeps the IEEE image, subnormals included
    float run_res = 0.0f;
    float run_max = 0.0f;                                // this wave's largest row norm (grid-stride over row groups:
    const bool vec = (D & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)h)) & 15) == 0;      // one atomic per workgroup at the end)
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < B; row += (int64_t)gridDim.x * 4) {
        const float* r = x + row * D;
        _Float16* o = h + row * D;
        float ss = 0.0f, mx = 0.0f, rs = 0.0f;
        if (vec) {
            // 8 elements per lane and step: two 16-B loads, one 16-B store
            for (int j = lane * 8; j < D; j += 512) {
                const float4 v0 = *reinterpret_cast<const float4*>(r + j), v1 = *reinterpret_cast<const float4*>(r + j + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                half8 hv;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float lost;
                    hv[q] = to_f16_normal(v[q] * scale, lost, flush);       // power-of-two scaling is exact
                    rs += lost * lost;
                    ss += v[q] * v[q];
                    mx = nanmax(mx, fabsf(v[q]));
                }
                *reinterpret_cast<half8*>(o + j) = hv;
            }
        } else {
            for (int j = lane; j < D; j += 64) {
                const float v = r[j];
                float lost;
                o[j] = to_f16_normal(v * scale, lost, flush);
                rs += lost * lost;
                ss += v * v;
                mx = nanmax(mx, fabsf(v));
            }
        }
        ss = wave_sum(ss);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) mx = nanmax(mx, __shfl_xor(mx, s, 64));
        ss *= 1.000001f;                                                    // slight inflation: norm2 is an upper bound
        rs = wave_sum(rs) * inv_s2 * 1.00001f;                              // unscaled, slightly inflated (the caller adds 1 % to eps)
        if (!(ss >= 0.0f)) ss = INFINITY;
        if (!(mx >= 0.0f)) mx = INFINITY;
        if (!(rs >= 0.0f)) rs = INFINITY;
        if (lane == 0) {
            if (norm2) norm2[row] = ss;
            if (amax) amax[row] = mx;
            if (res2) res2[row] = rs;
        }
        run_max = fmaxf(run_max, ss);
        run_res = fmaxf(run_res, rs);
    }
    if (want_norm_max) {                                   // uniform per launch
        if (lane == 0) { wn2[threadIdx.x >> 6] = run_max; wr2[threadIdx.x >> 6] = run_res; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMax(&stats[0], __float_as_uint(fmaxf(fmaxf(wn2[0], wn2[1]), fmaxf(wn2[2], wn2[3]))));
            atomicMax(&stats[2], __float_as_uint(fmaxf(fmaxf(wr2[0], wr2[1]), fmaxf(wr2[2], wr2[3]))));
        }
    }
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


// ---- 2b. large-problem variant: 256x256 tile, 512 threads (8 waves as 2x4, wave tile 128x64), two LDS
// stages (128 KB, one barrier per k-tile).  Half the L2->LDS traffic and 3/4 of the LDS reads per MFMA
// of the 128x128 kernel: 870 vs 620-730 TFLOP/s on 10k x 32k x 2048 (same box, same data).  Same k
// order, bit-identical scores.  Lab history: scratch/lab/f16_gemm_lab.hip (LDS-DMA staging was slower:
// 780; the MFMA + ds_read loop alone runs at 1150).
constexpr int GBM = 256, GBN = 256;
constexpr int GSTAGE_B = (GBM + GBN) * HBK * 2;          // 64 KB

template <bool FILTER, bool FULLK>
__global__ __launch_bounds__(512) void cosine_gemm_f16_big_kernel(const _Float16* __restrict__ Q, int64_t M,
                                                                  const _Float16* __restrict__ G, int64_t N, int D,
                                                                  float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n,
                                                                  const float* __restrict__ thr, uint8_t* __restrict__ gflag, int ngrp) {
    __shared__ __attribute__((aligned(16))) char S0[GSTAGE_B];
    __shared__ __attribute__((aligned(16))) char S1[GSTAGE_B];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * GBM, n0 = (int64_t)(first_n + within % gsz) * GBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging: 512 image rows (A then B) x 8 chunks of 16 B = 8 chunks per thread
    const int sr = tid >> 3, sc = tid & 7;
    const _Float16* gsrc[8];
    int ldst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (j & 3) * 64 + sr;
        const bool isb = j >= 4;
        int64_t gr = (isb ? n0 : m0) + row;
        const int64_t lim = isb ? N : M;
        gr = gr < lim ? gr : lim - 1;                         // clamp: rows past the edge are never stored
        gsrc[j] = (isb ? G : Q) + gr * D + sc * 8;
        ldst[j] = (isb ? GBM * 128 : 0) + row * 128 + ((sc ^ hswz(row)) << 4);
    }
    float4 reg[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)      // FULLK (D % 64 == 0): unconditional loads, no exec-masked branch in the k loop
            reg[j] = (FULLK || k0 + sc * 8 < D) ? *reinterpret_cast<const float4*>(gsrc[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lstore = [&](char* st) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(st + ldst[j]) = reg[j];
    };
    int a_off[4], b_off[2], a_sw[4], b_sw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int ra = wm * 128 + i * 32 + l31; a_off[i] = ra * 128; a_sw[i] = hswz(ra); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int rb = wn * 64 + j * 32 + l31; b_off[j] = GBM * 128 + rb * 128; b_sw[j] = hswz(rb); }
    auto compute = [&](const char* cur) {
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
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

    const int nk = (D + HBK - 1) / HBK;
    gload(0);
    lstore(S0);
    __syncthreads();
    if (nk > 1) gload(HBK);
    for (int kt = 0; kt < nk; kt += 2) {
        // even k-tile: compute from S0 while tile kt+1 goes to S1 and tile kt+2 is fetched
        if (kt + 1 < nk) {
            lstore(S1);
            if (kt + 2 < nk) gload((kt + 2) * HBK);
        }
        compute(S0);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) {
                lstore(S0);
                if (kt + 3 < nk) gload((kt + 3) * HBK);
            }
            compute(S1);
            __syncthreads();
        }
    }

    // epilogue.  The main loop of this kernel is short (fp16 rate), so the epilogue is a visible share of a
    // tile: thresholds come from LDS (one coalesced load per tile, then broadcast reads), addresses are
    // advanced by adds.
    float* thr_s = reinterpret_cast<float*>(S0);               // all waves are past their last S0 / S1 reads
    if (FILTER) {
        if (tid < GBM) thr_s[tid] = (m0 + tid < M) ? thr[m0 + tid] : INFINITY;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = wm * 128 + i * 32 + 4 * half;          // tile row of e = 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t ng = n0 + wn * 64 + j * 32;
            const int64_t n = ng + l31;
            const bool n_ok = n < N;
            float* cp = C + (m0 + rbase) * ldc + n;
            uint8_t* fp = FILTER ? gflag + (m0 + rbase) * (int64_t)ngrp + (ng >> 5) : nullptr;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);             // row offset inside the 32x32 tile
                const bool row_ok = (m0 + rbase + ro < M);
                const float v = acc[i][j][e];
                if (FILTER) {
                    const float t = thr_s[rbase + ro];
                    const unsigned long long qm = __ballot(n_ok && v >= t);
                    const bool qq = ((half ? (qm >> 32) : qm) & 0xFFFFFFFFull) != 0ull;
                    if (row_ok && ng < N) {
                        if (l31 == 0) fp[(int64_t)ro * ngrp] = qq ? 1 : 0;
                        if (qq && n_ok) cp[(int64_t)ro * ldc] = v;
                    }
                } else {
                    if (row_ok && n_ok) cp[(int64_t)ro * ldc] = v;
                }
            }
        }
    }
}

// ---- 2c. ping-pong variant of the 256x256 tile for D % 64 == 0 (the shipped path of the filter pass) ------------
// Same block tile, BK and swizzled [row][8 x 16 B] LDS image as 2b, different schedule (lab: scratch/lab/f16_pp_lab.hip):
//  * operand tiles arrive by LDS-DMA (global_load_lds_dwordx4; the XOR swizzle is applied to the per-lane SOURCE address, the
//    LDS side of a DMA is lane-linear) as 16-KB half-tiles (128 rows x 64 k) issued 1.5 k-tiles ahead and retired by ONE counted
//    s_waitcnt vmcnt(4) per k-tile: no staging registers, no ds_write (the LDS store path was what held 2b at 870 TFLOP/s);
//  * every 128x128 quadrant of the block tile is split 2 (M) x 4 (N) over the 8 waves: a wave owns a 64x32 piece of each quadrant
//    = 4 x 2 tiles of v_mfma_f32_16x16x32_f16 (same k order per output as the 32x32x16 form: scores bit-identical to 2b);
//  * a k-tile is two phases -- A: quadrants (0,0) + (0,1) [reads A0, B0, B1], B: quadrants (1,1) + (1,0) [reads A1] -- and a phase is
//    [operand ds_read_b128s + DMA issue + lgkmcnt(0)] s_barrier [32 MFMAs] s_barrier.  Waves 4-7 run ONE barrier behind waves 0-3,
//    so on every SIMD one wave is in its MFMA segment while its partner (wave w + 4) is in its load segment;
//  * hazards: a half-tile is read one phase or more after the barrier that follows every wave's vmcnt wait for it (RAW), and is
//    re-staged only after a barrier that follows the lgkmcnt(0) of both wave groups' reads of it (WAR).
// Measured (10 240 x 16 384, plain store): 1 204 TFLOP/s at D = 2048 and 1 370 at D = 4096 against 861 / 1 020 for 2b; the k loop
// itself runs at ~1.3 us per k-tile (1 650 TFLOP/s), the rest is the output burst of the plain-store epilogue.
// raw buffer descriptor from wave-uniform inputs (see gemm_tile.hpp uniform_rsrc; repeated here: fast.hip does not include it)
__device__ __forceinline__ auto uniform_rsrc16(const void* base, int64_t nbytes) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)base);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
    const unsigned nb = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nbytes < 0xFFFFFFFFll ? (nbytes > 0 ? nbytes : 0) : 0xFFFFFFFFll));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | (uintptr_t)lo), 0, (int)nb, 0x00020000);
}

constexpr int PP_HT_B = 128 * HBK * 2;                   // one half-tile: 16 KB
constexpr int PP_BUF_B = 4 * PP_HT_B;                    // [A0][A1][B0][B1] = 64 KB per k-tile
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// half-tile kinds in issue order within a k-tile: A0, B0 (first read in phase A), B1 (phase A), A1 (phase B) -> slot in the buffer
__device__ __forceinline__ constexpr int pp_slot(int j) { return j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 3 : 1; }

template <bool FILTER>
__global__ __launch_bounds__(512) void gemm_f16_pp_kernel(const _Float16* __restrict__ Q, int64_t M,
                                                          const _Float16* __restrict__ G, int64_t N, int D,
                                                          float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n,
                                                          const float* __restrict__ thr, uint8_t* __restrict__ gflag, int ngrp) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * PP_BUF_B];       // the ONLY LDS object (a second one makes hipcc drain vmcnt before ds_reads)
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * GBM, n0 = (int64_t)(first_n + within % gsz) * GBN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    ISX_PP_STAMP(0);

    f32x4_t acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][bb][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // DMA sources: half-tile j, instruction i fills image rows wave*16 + i*8 + lane/8; slot lane%8 of a row holds chunk slot ^ hswz(row).
    // buffer_load ... lds: a wave-uniform descriptor of the tile's 256 rows per operand, one constant 32-bit lane offset per
    // (half-tile, instruction), the k-tile offset as SGPR -- no vector address arithmetic in the load segments, where every
    // instruction beside the partner's MFMA stream is expensive.  Rows past the edge fall outside the descriptor (zeros, never stored).
    const auto rq = uniform_rsrc16(Q + m0 * D, ((M - m0) < GBM ? (M - m0) : GBM) * (int64_t)D * 2);
    const auto rg = uniform_rsrc16(G + n0 * D, ((N - n0) < GBN ? (N - n0) : GBN) * (int64_t)D * 2);
    unsigned gvo[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + (lane >> 3);
            gvo[j][i] = (unsigned)((pp_slot(j) & 1) * 128 + r) * (unsigned)D * 2u + (unsigned)(((lane & 7) ^ hswz(r)) << 4);
        }
    }
    const int T = D / HBK;                                      // D % 64 == 0 (launcher)
    auto issue = [&](int j, int t) {                            // half-tile j of k-tile t -> buffer t & 1
        char* dst = lds + (t & 1) * PP_BUF_B + pp_slot(j) * PP_HT_B + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (pp_slot(j) >= 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, gvo[j][i], (unsigned)t * (HBK * 2), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, gvo[j][i], (unsigned)t * (HBK * 2), 0, 0);
        }
    };
    // operand fragments of the 16x16x32 MFMA: lane l holds row (l & 15), chunk 4 s + l / 16 of its 16-row block
    const int x0 = (lane >> 4) ^ ((lane >> 1) & 7);               // chunk ^ hswz(row) for s = 0 (block bases are multiples of 16 rows)
    const int lrow = (lane & 15) * 128;
    int a_ad[2], b_ad[2];
    a_ad[0] = wm * 64 * 128 + lrow + (x0 << 4);
    a_ad[1] = wm * 64 * 128 + lrow + ((x0 ^ 4) << 4);
    b_ad[0] = 2 * PP_HT_B + wn * 32 * 128 + lrow + (x0 << 4);
    b_ad[1] = 2 * PP_HT_B + wn * 32 * 128 + lrow + ((x0 ^ 4) << 4);
    half8 af[4][2], bf[2][2][2];
    auto read_a = [&](const char* buf, int ah) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const half8*>(buf + ah * PP_HT_B + i * 2048 + a_ad[s]);
    };
    auto read_b = [&](const char* buf, int bh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[bh][j][s] = *reinterpret_cast<const half8*>(buf + bh * PP_HT_B + j * 2048 + b_ad[s]);
    };
    auto mfmas = [&](int ah) {
        __builtin_amdgcn_s_setprio(1);                          // also keeps hipcc from moving MFMAs across the barriers
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah][bh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][s], bf[bh][j][s], acc[ah][bh][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: k-tile 0 and half of k-tile 1 in flight; k-tile 0 landed and visible
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h / 4 < T) issue(h % 4, h / 4);
    if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                  // waves 4-7 run one barrier behind (wave-uniform branch)
    ISX_PP_STAMP(1);

    for (int t = 0; t < T; ++t) {
        const char* buf = lds + (t & 1) * PP_BUF_B;
        // ---- phase A
        read_a(buf, 0);
        read_b(buf, 0);
        read_b(buf, 1);
        if (t + 1 < T) { issue(2, t + 1); issue(3, t + 1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B: k-tile t + 1 retired before the barrier that precedes its first read
        read_a(buf, 1);
        if (t + 2 < T) { issue(0, t + 2); issue(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                  // pairs the extra barrier of waves 4-7: every LDS read is behind us now
    ISX_PP_STAMP(2);

    // epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r.  The two 16-column tiles j = 0, 1 of a
    // piece form one 32-column group of the filter.  The k loop is short (fp16 rate), so the epilogue is kept lean: thresholds
    // come from LDS (one 16-B read per four rows), the "any score of the group >= thr" test is two compares + wave ballots per row,
    // group flags are collected in LDS ([group][row] bytes, four rows per ds_write_b32) and leave as eight byte stores per row
    // at the end; scores are stored only for flagged groups.
    float* thr_s = reinterpret_cast<float*>(lds);                       // 256 floats
    uint8_t* flg_s = reinterpret_cast<uint8_t*>(lds) + 1024;            // [8 groups][256 rows]
    if (FILTER) {
        if (tid < GBM) thr_s[tid] = (m0 + tid < M) ? thr[m0 + tid] : INFINITY;
        __syncthreads();
    }
    ISX_PP_STAMP(3);
    const int l15 = lane & 15, lq = lane >> 4;
    const bool interior = (m0 + GBM <= M) && (n0 + GBN <= N);           // uniform: the common case carries no edge tests at all
    // the thresholds of this lane's 32 rows: eight 16-B LDS reads in flight together (one exposed latency instead of sixteen)
    f32x4_t t4s[2][4];
    if (FILTER) {
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
            for (int i = 0; i < 4; ++i) t4s[ah][i] = *reinterpret_cast<const f32x4_t*>(thr_s + ah * 128 + wm * 64 + i * 16 + lq * 4);
    }
    auto store_tile = [&](auto IN) {
        constexpr bool IN_ = decltype(IN)::value;
#pragma unroll
        for (int ah = 0; ah < 2; ++ah) {
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                const int64_t ng = n0 + bh * 128 + wn * 32;              // first column of this wave's 32-column group (uniform)
                const int64_t na = ng + l15, nb = na + 16;
                const bool a_ok = IN_ || na < N, b_ok = IN_ || nb < N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rbase = ah * 128 + wm * 64 + i * 16 + lq * 4;   // tile row of r = 0
                    float* cp = C + (m0 + rbase) * ldc + na;
                    if (FILTER) {
                        const f32x4_t t4 = t4s[ah][i];
                        if (IN_) {
                            // block reject first: the epilogue is VALU-issue-bound (two waves per SIMD, ~1 k instructions each: 10 % of a
                            // workgroup's life by shader-clock stamps, scratch/lab/f16_pp_epi_lab.hip) and with tight thresholds most 16 x 32
                            // blocks hold no candidate at all.  fmaxf drops a NaN operand, as the two compares below do.
                            const bool any = fmaxf(acc[ah][bh][i][0][0], acc[ah][bh][i][1][0]) >= t4[0] || fmaxf(acc[ah][bh][i][0][1], acc[ah][bh][i][1][1]) >= t4[1] ||
                                             fmaxf(acc[ah][bh][i][0][2], acc[ah][bh][i][1][2]) >= t4[2] || fmaxf(acc[ah][bh][i][0][3], acc[ah][bh][i][1][3]) >= t4[3];
                            if (__ballot(any) == 0ull) {
                                if (l15 == 0) *reinterpret_cast<unsigned*>(flg_s + (bh * 4 + wn) * 256 + rbase) = 0u;
                                continue;
                            }
                        }
                        unsigned fl = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float va = acc[ah][bh][i][0][r], vb = acc[ah][bh][i][1][r];
                            const unsigned long long qm = IN_ ? (__ballot(va >= t4[r]) | __ballot(vb >= t4[r]))
                                                              : (__ballot(a_ok && va >= t4[r]) | __ballot(b_ok && vb >= t4[r]));
                            const bool qq = ((unsigned)(qm >> (lane & 48)) & 0xFFFFu) != 0u;      // any of this row's 32 columns
                            fl |= (qq ? 1u : 0u) << (8 * r);
                            if (qq && (IN_ || m0 + rbase + r < M)) {
                                if (a_ok) cp[(int64_t)r * ldc] = va;
                                if (b_ok) cp[(int64_t)r * ldc + 16] = vb;
                            }
                        }
                        if (l15 == 0) *reinterpret_cast<unsigned*>(flg_s + (bh * 4 + wn) * 256 + rbase) = fl;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (IN_ || m0 + rbase + r < M) {
                                if (a_ok) cp[(int64_t)r * ldc] = acc[ah][bh][i][0][r];
                                if (b_ok) cp[(int64_t)r * ldc + 16] = acc[ah][bh][i][1][r];
                            }
                        }
                    }
                }
            }
        }
    };
    if (interior) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    ISX_PP_STAMP(4);
    if (FILTER) {
        __syncthreads();
        if (tid < GBM && m0 + tid < M) {
            uint8_t* fp = gflag + (m0 + tid) * (int64_t)ngrp + (n0 >> 5);
            if (interior && (ngrp & 7) == 0 && (((uintptr_t)gflag) & 7) == 0) {      // the eight flag bytes of a row as ONE 8-byte store (n0 / 32 is a multiple of 8)
                unsigned long long v = 0ull;
#pragma unroll
                for (int g = 0; g < 8; ++g) v |= (unsigned long long)flg_s[g * 256 + tid] << (8 * g);
                *reinterpret_cast<unsigned long long*>(fp) = v;
            } else {
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    if (interior || n0 + g * 32 < N) fp[g] = flg_s[g * 256 + tid];
            }
        }
    }
    ISX_PP_STAMP(5);
}

static std::atomic<int> g_force_f16_tile{-1};        // debug / A-B hook: 0 = 128x128, 1 = 256x256, 2 = 256x256 register-staged (2b), -1 = automatic

int launch_gemm_f16(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C, int64_t ldc, const float* thr,
                           uint8_t* gflag, hipStream_t st) {
    if (M == 0 || N == 0) return ISX_OK;
    const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128;
    if (tm * tn >= (1ll << 31)) { isx_set_error("f16 gemm: too many tiles"); return ISX_ERR_ARG; }
    const int ngrp = (int)((N + 31) / 32);
    // 256x256 tiles (one 512-thread workgroup per CU) once they fill the chip for >= 2 rounds
    const int64_t btm = (M + GBM - 1) / GBM, btn = (N + GBN - 1) / GBN;
    const bool big = g_force_f16_tile >= 0 ? g_force_f16_tile >= 1 : (btm * btn >= 512);
    if (big) {
        const dim3 grid((unsigned)(btm * btn)), block(512);
        const bool fullk = (D % HBK == 0);
        const bool pp = fullk && g_force_f16_tile != 2;            // 2 = the register-staged 2b kernel on full-k shapes too (A/B)
        if (gflag && pp) hipLaunchKernelGGL((gemm_f16_pp_kernel<true>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        else if (pp) hipLaunchKernelGGL((gemm_f16_pp_kernel<false>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        else if (gflag && fullk) hipLaunchKernelGGL((cosine_gemm_f16_big_kernel<true, true>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        else if (gflag) hipLaunchKernelGGL((cosine_gemm_f16_big_kernel<true, false>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        else if (fullk) hipLaunchKernelGGL((cosine_gemm_f16_big_kernel<false, true>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        else hipLaunchKernelGGL((cosine_gemm_f16_big_kernel<false, false>), grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)btm, (int)btn, thr, gflag, ngrp);
        ISX_CHECK_LAUNCH("cosine_gemm_f16_big");
        return ISX_OK;
    }
    const dim3 grid((unsigned)(tm * tn)), block(256);
    if (gflag) hipLaunchKernelGGL(cosine_gemm_f16_kernel<true>, grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)tm, (int)tn, thr, gflag, ngrp);
    else hipLaunchKernelGGL(cosine_gemm_f16_kernel<false>, grid, block, 0, st, Q, M, G, N, D, C, ldc, (int)tm, (int)tn, thr, gflag, ngrp);
    ISX_CHECK_LAUNCH("cosine_gemm_f16");
    return ISX_OK;
}


// ---- 3. exact re-scoring of the candidates inside the error window -------------------------------
// One 256-thread workgroup per query row.  cand: the row's KL best APPROXIMATE keys (canonical order,
// 0 = empty).  Window: approx >= a_k - 2 eps (a_k = k-th best approximate score).  Sufficient iff the
// window ends inside the list (or the list holds the whole gallery).  Candidate t is re-scored by
// thread t with the canonical fma chain; the candidate rows are staged through LDS 32 k-values at a
// time (8 lanes fetch one 128-B row segment: coalesced) and read back conflict-free ([cand][36] floats).
constexpr int RS_T = 256;
constexpr int RS_BK = 32;
constexpr int RS_LD = 36;

__device__ __forceinline__ float fast_eps(float qn2, float qres2, float gn2, float gres2, int D) {
    // eps = rq (|g|max + rg) + |q| rg + 3 D 2^-24 (|q| + rq)(|g|max + rg), evaluated in fp32 and inflated by 1 %
    const float qn = sqrtf(qn2), rq = sqrtf(qres2), gn = sqrtf(gn2), rg = sqrtf(gres2);
    const float c1 = 3.0f * (float)D * 5.9604644775390625e-8f;
    return 1.01f * (rq * (gn + rg) + qn * rg + c1 * (qn + rq) * (gn + rg));
}

// Width of the window of interest per query row, in the (scaled) score domain of the approximate pass: 2 eps; +inf where the fp16 pass
// is not usable (the row then takes the exact fallback).  Read by the group selects of the approximate pass and by rescore_kernel.
__global__ __launch_bounds__(256) void window_kernel(const float* __restrict__ qnorm2, const float* __restrict__ qres2, const float* __restrict__ qstats,
                                                     const float* __restrict__ gstats, int D, int64_t M, float* __restrict__ win) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const float qmax = qstats[1], gmax = gstats[1];
    const float sc = f16_scale(qmax) * f16_scale(gmax);
    const float eps = fast_eps(qnorm2[row], qres2[row], gstats[0], gstats[2], D) * sc;
    const bool usable = f16_usable(qmax) && f16_usable(gmax) && (eps < 1e30f);
    win[row] = usable ? 2.0f * eps : INFINITY;
}

__global__ __launch_bounds__(RS_T) void rescore_kernel(const float* __restrict__ Q, const float* __restrict__ G, int D, int64_t N, int KL, int k,
                                                       const uint64_t* __restrict__ cand, const float* __restrict__ win,
                                                       int64_t idx_base, float* __restrict__ top_score, int64_t* __restrict__ top_idx,
                                                       int* __restrict__ ok) {
    __shared__ __attribute__((aligned(16))) float tile[RS_T * RS_LD];     // 36 KB
    __shared__ uint64_t keys[RS_T];
    __shared__ int cnt_s[2];
    const int t = threadIdx.x;
    const int64_t row = blockIdx.x;
    const uint64_t key = (t < KL) ? cand[row * KL + t] : 0ull;
    keys[t] = key;
    if (t < 2) cnt_s[t] = 0;
    __syncthreads();
    // approximate scores live in the scaled domain: S' ~ sq * sg * S; so does the window width 2 eps (window_kernel)
    const float w2 = win[row];
    const int kk = (k < KL) ? k : KL;
    const uint64_t kth = keys[kk - 1];                                    // 0 when the list holds fewer than k
    const bool usable = (w2 < INFINITY);
    float floor_v = -INFINITY;
    if (kth) floor_v = key_score(kth) - w2 - 2e-7f * fabsf(key_score(kth));   // (fp32 rounding of this line covered)
    const bool valid = key != 0ull;
    const bool in_win = valid && (key_score(key) >= floor_v);
    if (valid) atomicAdd(&cnt_s[0], 1);
    if (in_win) atomicAdd(&cnt_s[1], 1);
    __syncthreads();
    const int nvalid = cnt_s[0], c = cnt_s[1];                            // the window is a prefix of the sorted list
    const bool sufficient = usable && (N <= KL || c < KL) && (nvalid >= (N < KL ? (int)N : KL));
    if (!sufficient) {
        if (t == 0) ok[row] = 0;
        return;
    }
    if (t == 0) ok[row] = 1;
    // ---- exact scores of candidates 0..c-1 ----
    const float* q = Q + row * D;
    const int lr = t >> 3, lc = t & 7;                                    // staging role: row-in-pass, 16-B chunk
    const int npass = (c + 31) >> 5;
    const float* gp[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ci = p * 32 + lr;
        const uint64_t kc = keys[ci < c ? ci : 0];
        gp[p] = G + (int64_t)key_idx(kc) * D + lc * 4;
    }
    float acc = 0.0f;
    float4 reg[8];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < npass) reg[p] = (k0 + lc * 4 < D) ? *reinterpret_cast<const float4*>(gp[p] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < npass) *reinterpret_cast<float4*>(&tile[(p * 32 + lr) * RS_LD + lc * 4]) = reg[p];
    };
    load(0);
    store();
    __syncthreads();
    for (int k0 = 0; k0 < D; k0 += RS_BK) {
        const bool more = (k0 + RS_BK < D);
        if (more) load(k0 + RS_BK);
        if (t < c) {
            const int kn = (D - k0 < RS_BK) ? D - k0 : RS_BK;
            if (kn == RS_BK) {
#pragma unroll
                for (int u = 0; u < RS_BK; u += 4) {
                    const float4 g4 = *reinterpret_cast<const float4*>(&tile[t * RS_LD + u]);
                    acc = fmaf(q[k0 + u], g4.x, acc);
                    acc = fmaf(q[k0 + u + 1], g4.y, acc);
                    acc = fmaf(q[k0 + u + 2], g4.z, acc);
                    acc = fmaf(q[k0 + u + 3], g4.w, acc);
                }
            } else {
                for (int u = 0; u < kn; ++u) acc = fmaf(q[k0 + u], tile[t * RS_LD + u], acc);
            }
        }
        __syncthreads();
        if (more) {
            store();
            __syncthreads();
        }
    }
    __syncthreads();
    keys[t] = (t < c) ? rank_key(acc, key_idx(key)) : 0ull;
    __syncthreads();
    bitonic_sort_desc<RS_T>(keys, RS_T);
    for (int i = t; i < k; i += RS_T) {
        const uint64_t kx = (i < RS_T) ? keys[i] : 0ull;
        top_score[row * k + i] = kx ? key_score(kx) : -INFINITY;
        top_idx[row * k + i] = kx ? (idx_base + (int64_t)key_idx(kx)) : -1;
    }
}

// ---- 4. rows whose window was not covered: compact, gather, exact fp32 search ---------------------
__global__ __launch_bounds__(1024) void compact_rows_kernel(const int* __restrict__ ok, int M, int* __restrict__ list, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 1024) {
        const int i = i0 + t;
        const bool bad = (i < M) && (ok[i] == 0);
        const unsigned long long m = __ballot(bad);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (bad) list[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (t == 0) { int s2 = 0; for (int w = 0; w < 16; ++w) s2 += wsum[w]; base_s += s2; }
        __syncthreads();
    }
    if (t == 0) *count = base_s;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ Q, int D, const int* __restrict__ list,
                                                          const int* __restrict__ count, float* __restrict__ out) {
    const int r = blockIdx.x;
    if (r >= *count) return;
    const float* src = Q + (int64_t)list[r] * D;
    float* dst = out + (int64_t)r * D;
    for (int j = threadIdx.x; j < D; j += 256) dst[j] = src[j];
}

static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
static int fast_kl(int k) {
    static const int pct = [] { const char* e = getenv("ISX_FAST_KL_PCT"); return e ? atoi(e) : 200; }();      // A/B: candidates kept per query, in % of k (+ 32)
    int kl = (pct * k / 100 + 32 + 31) / 32 * 32;
    if (kl < 64) kl = 64;
    if (kl > kGroupSelectMaxK) kl = kGroupSelectMaxK;
    return kl;
}
constexpr int kFastMaxK = 128;

struct FastLayout {
    size_t qh, qn2, qstats, gh, gstats, ok, list, count, qbad, fixed_approx, fixed_exact, chunk, total;
    int64_t nc;
};
static FastLayout fast_layout(int64_t M, int64_t N, int D, int k, bool own_gallery) {
    FastLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += a256(bytes); return at; };
    L.qh = take((size_t)M * D * 2);
    L.qn2 = take((size_t)M * 4);
    L.qstats = take(16);
    L.gh = take(own_gallery ? (size_t)N * D * 2 : 0);
    L.gstats = take(16);
    L.ok = take((size_t)M * 4);
    L.list = take((size_t)M * 4);
    L.count = take(16);
    L.qbad = take((size_t)M * D * 4);
    // the two chunk pipelines run one after the other and share one area: [fixed | flags | chunk]
    const int kl = fast_kl(k);
    L.fixed_approx = topk_fixed_bytes(M, kl);
    L.fixed_exact = topk_fixed_bytes(M, k);
    L.chunk = o;
    L.nc = topk_recommended_chunk(M, N);
    L.total = o + L.fixed_approx + topk_chunk_bytes(M, L.nc);
    return L;
}

}  // namespace isx

using namespace isx;

// fp32 rows -> fp16 rows (RNE) + per-row squared norm (upper bound) and max |x|.  h: (B,D) fp16 (2 B/elem).
ISX_API int isx_rows_to_f16(const float* x, int64_t B, int D, void* h, float* norm2, float* amax, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D > 0 && B < (1ll << 33), "isx_rows_to_f16: bad shape B=%lld D=%d", (long long)B, D);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(x && h && norm2 && amax, "isx_rows_to_f16: null pointer");
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, B, D, (_Float16*)h, norm2, amax,
                       (unsigned*)nullptr, 0, (float*)nullptr);
    ISX_CHECK_LAUNCH("isx_rows_to_f16");
    return ISX_OK;
}

// stats must be zeroed (stream-ordered) before: pass 1 max |x|, pass 2 scaled conversion + max norm2.
static int convert_scaled(const float* x, int64_t B, int D, _Float16* h, float* norm2, float* res2, unsigned* stats, int want_norm_max, hipStream_t st) {
    const int64_t n = B * D;
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, stats);
    ISX_CHECK_LAUNCH("amax");
    hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((B + 3) / 4 < 4096 ? (B + 3) / 4 : 4096)), dim3(256), 0, st, x, B, D, h, norm2, (float*)nullptr, stats,
                       want_norm_max, res2);
    ISX_CHECK_LAUNCH("rows_to_f16");
    return ISX_OK;
}

// Gallery preparation for isx_cosine_topk_fast: power-of-two scaled fp16 image of the rows + gstats[4] =
// {max squared row norm (upper bound), max |x|, max squared norm of what a row lost in the conversion, 0}.  Done once per gallery
// shard and cached by the caller.
ISX_API int isx_gallery_to_f16(const float* G, int64_t N, int D, void* Gh, float* gstats, isx_stream_t stream) {
    ISX_REQUIRE(N >= 0 && D > 0 && N < (1ll << 33), "isx_gallery_to_f16: bad shape N=%lld D=%d", (long long)N, D);
    ISX_REQUIRE(gstats && ((G && Gh) || N == 0), "isx_gallery_to_f16: null pointer");
    if (hipMemsetAsync(gstats, 0, 16, (hipStream_t)stream) != hipSuccess) { isx_set_error("isx_gallery_to_f16: hipMemsetAsync failed"); return ISX_ERR_HIP; }
    if (N == 0) return ISX_OK;
    return convert_scaled(G, N, D, (_Float16*)Gh, nullptr, nullptr, (unsigned*)gstats, 1, (hipStream_t)stream);
}

// Approximate similarity matrix from fp16 operands (fp32 accumulate): building block / diagnostic of the
// fast path.  Qh: (M,D) fp16, Gh: (N,D) fp16, D % 8 == 0, 16-B aligned.
ISX_API int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0 && D % 8 == 0, "isx_cosine_sim_f16: bad shape M=%lld N=%lld D=%d (D %% 8 == 0 required)", (long long)M, (long long)N, D);
    ISX_REQUIRE((Qh && Gh && sim) || M * N == 0, "isx_cosine_sim_f16: null pointer");
    ISX_REQUIRE((((uintptr_t)Qh | (uintptr_t)Gh) % 16) == 0, "isx_cosine_sim_f16: operands must be 16-B aligned");
    return launch_gemm_f16((const _Float16*)Qh, M, (const _Float16*)Gh, N, D, sim, N, nullptr, nullptr, (hipStream_t)stream);
}

// Shapes the filter path does not cover run the fp32 path unchanged.
static bool fast_applicable(int64_t M, int64_t N, int D, int k) {
    return M > 0 && N > 0 && D % 8 == 0 && k <= kFastMaxK && M < (1ll << 31) && N > 4 * (int64_t)fast_kl(k);
}

ISX_API size_t isx_cosine_topk_fast_workspace(int64_t M, int64_t N, int D, int k, int have_gallery_f16) {
    if (M <= 0 || N <= 0 || k <= 0 || D <= 0) return 256;
    const size_t plain = isx_cosine_topk_workspace(M, N, D, k);
    if (!fast_applicable(M, N, D, k)) return plain;
    const size_t fast = fast_layout(M, N, D, k, !have_gallery_f16).total;
    return fast > plain ? fast : plain;
}

ISX_API int isx_cosine_topk_fast(const float* Q, int64_t M, const float* G, int64_t N, int D, int k, int64_t idx_base,
                                 const void* Gh, const float* gstats, float* top_score, int64_t* top_idx, void* ws,
                                 size_t ws_bytes, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0, "isx_cosine_topk_fast: bad shape M=%lld N=%lld D=%d", (long long)M, (long long)N, D);
    ISX_REQUIRE(k >= 1 && k <= kSelectMaxK, "isx_cosine_topk_fast: k=%d outside [1,%d]", k, kSelectMaxK);
    ISX_REQUIRE((Gh == nullptr) == (gstats == nullptr), "isx_cosine_topk_fast: Gh and gstats go together");
    const bool aligned = ((((uintptr_t)Q | (uintptr_t)G | (uintptr_t)Gh) % 16) == 0);
    if (!fast_applicable(M, N, D, k) || !aligned)
        return isx_cosine_topk(Q, M, G, N, D, k, idx_base, top_score, top_idx, ws, ws_bytes, stream);
    ISX_REQUIRE(idx_base >= 0 && idx_base + N <= 0xFFFFFFFFll && N <= 0x7FFFFFFFll, "isx_cosine_topk_fast: gallery indices must stay below 2^32");
    ISX_REQUIRE(Q && G && top_score && top_idx, "isx_cosine_topk_fast: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool own = (Gh == nullptr);
    const FastLayout L = fast_layout(M, N, D, k, own);
    const size_t need_min = L.chunk + L.fixed_approx + topk_chunk_bytes(M, 128);
    if (!ws || ((uintptr_t)ws % 256) != 0 || ws_bytes < need_min) {
        isx_set_error("isx_cosine_topk_fast: workspace of %zu bytes too small or misaligned (need >= %zu, 256-B aligned)", ws_bytes, need_min);
        return ISX_ERR_WORKSPACE;
    }
    char* w = (char*)ws;
    _Float16* qh = (_Float16*)(w + L.qh);
    float* qn2 = (float*)(w + L.qn2);
    float* qst = (float*)(w + L.qstats);
    int* ok = (int*)(w + L.ok);
    int* list = (int*)(w + L.list);
    int* count = (int*)(w + L.count);
    float* qbad = (float*)(w + L.qbad);
    const int kl = fast_kl(k);
    int rc;

    // 1. fp16 operands
    if (hipMemsetAsync(qst, 0, 16, st) != hipSuccess) { isx_set_error("isx_cosine_topk_fast: hipMemsetAsync failed"); return ISX_ERR_HIP; }
    float* qres2 = (float*)ok;                          // per-row conversion loss, read by window_kernel; rescore_kernel rewrites the slot as `ok`
    rc = convert_scaled(Q, M, D, qh, qn2, qres2, (unsigned*)qst, 0, st);
    if (rc) return rc;
    const _Float16* gh = (const _Float16*)Gh;
    const float* gs = gstats;
    if (own) {
        rc = isx_gallery_to_f16(G, N, D, w + L.gh, (float*)(w + L.gstats), stream);
        if (rc) return rc;
        gh = (const _Float16*)(w + L.gh);
        gs = (const float*)(w + L.gstats);
    }
    // 2. the KL best approximate candidates per row (keys stay in the workspace)
    TopkJob a{};
    a.who = "isx_cosine_topk_fast";
    a.Qh = qh; a.Gh = gh; a.M = M; a.N = N; a.D = D; a.k = kl;
    a.idx_base = 0; a.emit = false;
    a.ws = w + L.chunk; a.ws_bytes = ws_bytes - L.chunk; a.st = st;
    // only candidates within 2 eps of the k-th best approximate score are re-scored: the selects tighten the filter threshold accordingly
    // (the window widths live in the `list` slot, which compact_rows_kernel rewrites after the approximate pass)
    float* win = (float*)(w + L.list);
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, qn2, qres2, qst, gs, D, M, win);
    ISX_CHECK_LAUNCH("window");
    a.win = (k < kl) ? win : nullptr; a.k_win = k;
    rc = run_topk_chunks(a);
    if (rc) return rc;
    const uint64_t* cand = (const uint64_t*)(w + L.chunk);
    // 3. exact re-scoring inside the error window
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)M), dim3(RS_T), 0, st, Q, G, D, N, kl, k, cand, win, idx_base, top_score, top_idx, ok);
    ISX_CHECK_LAUNCH("rescore");
    // 4. exact fp32 search for the rows that were not covered (usually none: every launch below exits at once)
    hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, st, ok, (int)M, list, count);
    ISX_CHECK_LAUNCH("compact_rows");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)M), dim3(256), 0, st, Q, D, list, count, qbad);
    ISX_CHECK_LAUNCH("gather_rows");
    TopkJob e{};
    e.who = "isx_cosine_topk_fast";
    e.Q = qbad; e.G = G; e.M = M; e.N = N; e.D = D; e.k = k;
    e.idx_base = idx_base; e.top_score = top_score; e.top_idx = top_idx; e.emit = true;
    e.ws = w + L.chunk; e.ws_bytes = ws_bytes - L.chunk; e.st = st;
    e.m_active = count; e.row_map = list;
    return run_topk_chunks(e);
}

// Debug / A-B hook (not declared in include/isx.h): force the fp16 GEMM tile (0 = 128x128, 1 = 256x256, -1 = automatic).
ISX_API void isx_debug_set_f16_tile(int t) { g_force_f16_tile = t; }

// Byte offset, inside the workspace of isx_cosine_topk_fast called with the same arguments, of an int32 that holds -- once
// that call has completed on its stream -- the number of query rows that needed the exact fp32 fallback.  (size_t)-1 when such a
// call runs the fp32 search as a whole.  Lets a caller watch the fallback rate without a host synchronisation inside libisx.
ISX_API size_t isx_cosine_topk_fast_fallback_offset(int64_t M, int64_t N, int D, int k, int have_gallery_f16) {
    if (!fast_applicable(M, N, D, k)) return (size_t)-1;
    return fast_layout(M, N, D, k, !have_gallery_f16).count;
}

// Debug / test hook (not declared in include/isx.h): number of query rows of the LAST isx_cosine_topk_fast
// call on this workspace that took the exact fp32 fallback (-1: the call ran the fp32 path as a whole).
// Synchronises the device.
ISX_API int isx_debug_fast_fallback_rows(const void* ws, int64_t M, int64_t N, int D, int k, int have_gallery_f16) {
    if (!ws || !fast_applicable(M, N, D, k)) return -1;
    const FastLayout L = fast_layout(M, N, D, k, !have_gallery_f16);
    int c = -2;
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    if (hipMemcpy(&c, (const char*)ws + L.count, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return c;
}
