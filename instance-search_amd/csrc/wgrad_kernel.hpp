// wgrad_kernel.hpp -- the TN GEMM over a K-major pair of operands (shared by backward.hip: convolution weight gradients, and head.hip: the
// input gradient of the descriptor head's Linear).
#pragma once
#include "gemm_tile.hpp"

namespace isx {

// ---- weight gradient: TN GEMM over the pixels ---------------------------------------------------------------------------------
// pixel p of dZ -> row of X that meets it under filter tap (dh, dw): identity for a 1x1 convolution, (ho * s + dh, wo * s + dw) otherwise
struct WgradGeom { int ident, H, W, Ho, Wo, stride; };

__device__ __forceinline__ int64_t wgrad_src_row(const WgradGeom& g, int64_t p, int dh, int dw) {
    if (g.ident) return p;
    const int hw = g.Ho * g.Wo;
    const int b = (int)(p / hw), rem = (int)(p - (int64_t)b * hw);
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    const int h = ho * g.stride + dh, w = wo * g.stride + dw;
    if (h < 0 || h >= g.H || w < 0 || w >= g.W) return -1;
    return ((int64_t)b * g.H + h) * g.W + w;
}

// C[s][n1][tap * N2 + n2] = sum over the pixels p of split s of A[p][n1] * B[src(p, tap)][n2];  block tile (64 TM) x (64 TN), BK = 32
// pixels per k-tile.  grid = (tiles, taps, splits): split s owns the k-tiles [s * kt_per, (s + 1) * kt_per) and writes its own partial
// (the consumer, bn_fold_backward_kernel, adds the partials in split order: a fixed summation tree, no atomics).  The blocks of
// (tile_n = 0, tap = 0) also produce the partial COLUMN SUMS of A (the bias gradient) from the A tiles they stage anyway.
// N1 % (64 TM) == 0, N2 % (64 TN) == 0.
// FOLD > 0 (the descriptor head's input gradient, head.hip): the reduction is summed in two levels -- a chain per FOLD k-tiles, the chains added in
// order into a second accumulator (gemm_tile.hpp) -- so that a run whose reduction is SHARDED across ranks (every rank a few of the chains,
// isx_head_linear_dgrad_parts) lands on the same bits.  fold_kt: k-tiles per chain.
// Workgroups per CU (round 5): without a bound hipcc spent 224 registers on the 128x128 shape and 293 on the folding 192x64 one (ONE wave per SIMD)
#ifndef ISX_LB_WGRAD
#define ISX_LB_WGRAD 3
#endif
template <int TM, int TN, bool FOLD = false>
__global__ __launch_bounds__(256, TM * TN == 1 ? 4 : FOLD ? 2 : ISX_LB_WGRAD) void wgrad_gemm_kernel(const float* __restrict__ A, int64_t K, int N1, const float* __restrict__ Bm, int N2,
                                                         WgradGeom g, int taps, float* __restrict__ C, int64_t ldc, int tiles_n, int kt_per,
                                                         int splits, float* __restrict__ colsum, int fold_kt = 0) {
    constexpr int BK = 32, BM = 64 * TM, BN = 64 * TN, LDA = BM + 4, LDB = BN + 4;       // +4: rows stay 16-B aligned for the float4 stores
    constexpr int CA = BM / 4, CB = BN / 4, NA = BK * CA / 256, NB = BK * CB / 256;
    __shared__ float lds[BK * (LDA + LDB)];
    float* As = lds;
    float* Bs = lds + BK * LDA;
    const int tile_m = (int)blockIdx.x / tiles_n, tile_n = (int)blockIdx.x % tiles_n;
    // blockIdx.z = leaf * splits + sub: K is the pixel count of ONE leaf (micro-batch); leaf l owns the pixels [l * K, (l + 1) * K) of dz and its
    // k-tiles restart at its first pixel, so a leaf's partials are the same bits whether it is launched alone or with its siblings
    const int tap = (int)blockIdx.y, split = (int)blockIdx.z, sub = split % splits;
    const int64_t pix0 = (int64_t)(split / splits) * K;
    const int dh = taps == 9 ? tap / 3 - 1 : 0, dw = taps == 9 ? tap % 3 - 1 : 0;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const bool sums = colsum != nullptr && tile_n == 0 && tap == 0 && (int)threadIdx.x < BM;      // (uniform per wave: BM is 64 or 128)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[NA], rb[NB];
    auto load = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const int64_t p = k0 + idx / CA;
            ra[j] = p < K ? *reinterpret_cast<const float4*>(A + (pix0 + p) * N1 + m0 + ((idx % CA) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const int64_t p = k0 + idx / CB;
            const int64_t s = p < K ? wgrad_src_row(g, pix0 + p, dh, dw) : -1;
            rb[j] = s >= 0 ? *reinterpret_cast<const float4*>(Bm + s * N2 + n0 + ((idx % CB) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = j * 256 + threadIdx.x;
            *reinterpret_cast<float4*>(As + (idx / CA) * LDA + ((idx % CA) << 2)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = j * 256 + threadIdx.x;
            *reinterpret_cast<float4*>(Bs + (idx / CB) * LDB + ((idx % CB) << 2)) = rb[j];
        }
    };

    const int64_t nk_all = (K + BK - 1) / BK;
    const int64_t kt0 = (int64_t)sub * kt_per;
    const int64_t kt1 = kt0 + kt_per < nk_all ? kt0 + kt_per : nk_all;
    float csum = 0.0f;
    f32x16 tot[FOLD ? TM : 1][FOLD ? TN : 1];
    if constexpr (FOLD) zero_tiles(tot);
    if (kt0 < kt1) {
        load(kt0 * BK);
        store();
        __syncthreads();
        const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
        const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
        if constexpr (FOLD) {
            for (int64_t kt = kt0; kt < kt1;) {
                const int64_t kend = kt + fold_kt < kt1 ? kt + fold_kt : kt1;
                for (; kt < kend; ++kt) {
                    const bool more = kt + 1 < kt1;
                    if (more) load((kt + 1) * BK);
                    mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
                    __syncthreads();
                    if (more) {
                        store();
                        __syncthreads();
                    }
                }
                fold_chunk<TM, TN>(tot, acc);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
        } else
        for (int64_t kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) load((kt + 1) * BK);
            if (sums) {
#pragma unroll
                for (int k = 0; k < BK; ++k) csum += As[k * LDA + threadIdx.x];           // pixel order
            }
            mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
            __syncthreads();
            if (more) {
                store();
                __syncthreads();
            }
        }
    }
    if (sums) colsum[(int64_t)split * N1 + m0 + threadIdx.x] = csum;
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    float* Ct = C + (int64_t)split * N1 * ldc + (int64_t)tap * N2;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                Ct[(int64_t)row * ldc + col] = acc[i][j][e];
            }
        }
}

}  // namespace isx
