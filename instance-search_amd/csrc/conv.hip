// conv.hip -- convolutions of the inference trunk on channels-last activations (model/nn_utils.py fold_batch_norm).
//
//   isx_conv1x1_nhwc   the activation matrix (B*H*W, Cin) IS the row-major A operand of the fp32-MFMA GEMM of
//                      cosine.hip; this file only adds the entry point (epilogue mode 2 of cosine_gemm_kernel)
//   isx_conv3x3_nhwc        implicit GEMM below (conv3x3_nhwc_kernel)
//   isx_conv1x1_dual_nhwc   last 1x1 convolution of a bottleneck block + its projection shortcut as ONE GEMM over the
//                           concatenated K = [t ; x_strided] (conv1x1_dual_nhwc_kernel)
//   (expand.hip: isx_conv3x3_expand_nhwc / isx_conv3x3_expand_dual_nhwc, conv2 + conv3 (+ projection) of the 64-channel bottlenecks as ONE kernel,
//    on the main loop of conv3x3_tile.hpp)
// Launches in 128x128 tiles run the rows past their last whole round of resident workgroups as 64x64 tiles in the same grid
// (conv3x3_tail_kernel, conv1x1_dual_tail_kernel; conv1x1_tail_kernel in cosine.hip); isx_debug_set_conv_cfg(7) turns that off (A/B).
//
// Reference call sites: the torchvision ResNet `features` trunk built by model/ModelDefinition.py, split by
// model/nn_utils.py:56-71 and run from model/siamese.py:20,107,151.
#include <stdlib.h>

#include "conv3x3_tile.hpp"

namespace isx {

// one output tile: main loop + bias / residual / ReLU epilogue.  CHUNK: two-level sum of the inference trunk (gemm_tile.hpp), 0 for the gradient kernels
template <int TM, int TN, int BK, int CHUNK = kConvChunk>
__device__ __forceinline__ void conv3x3_tile(float* __restrict__ lds, const float* __restrict__ x, int64_t M, const float* __restrict__ Wt, int64_t N,
                                             const Conv3x3Geom& g, float* __restrict__ C, int64_t m0, int64_t n0,
                                             const float* __restrict__ bias, const float* __restrict__ res, int relu, const float* __restrict__ mask = nullptr) {
    f32x16 acc[TM][TN];
    conv3x3_mainloop<TM, TN, BK, false, 2, CHUNK>(lds, x, M, Wt, N, g, m0, n0, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm_u = __builtin_amdgcn_readfirstlane(wave >> 1), wn_u = __builtin_amdgcn_readfirstlane(wave & 1);
    // residual / mask look-ahead by register budget: gradient kernels (CHUNK == 0, 116 of 128 VGPRs) one tile at a time, 128x128 inference tiles two
    conv_epilogue_buffers<TM, TN, (CHUNK == 0 || TM * TN < 4) ? 1 : 2>(acc, C, res, bias, relu, m0, M, n0, N, N, 64 * TM, wm_u * (32 * TM), wn_u * (32 * TN), lane & 31, lane >> 5, mask);
}

// workgroups per CU: 64x64 tiles 6; 128x64 4 (two-level: 126 VGPRs); 128x128 4 with one accumulator set (gradients), 2 with two (64 + 64 accumulator VGPRs)
template <int TM, int TN, int BK, bool GRAD>
__global__ __launch_bounds__(256, TM * TN == 1 ? 6 : (TM * TN == 4 && !GRAD) ? ISX_WG_PER_CU_128 : 4) void conv3x3_nhwc_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ Wt, int64_t N,
                                                           Conv3x3Geom g, float* __restrict__ C, TileMap tm,
                                                           const float* __restrict__ bias, const float* __restrict__ res, int relu, const float* __restrict__ mask) {
    __shared__ float lds[BK * (64 * TM + 64 * TN + 2 * lds_pad(BK))];
    int tile_m, tile_n;
    tile_of_block(tm, tile_m, tile_n);
    conv3x3_tile<TM, TN, BK, GRAD ? 0 : kConvChunk>(lds, x, M, Wt, N, g, C, (int64_t)tile_m * (64 * TM), (int64_t)tile_n * (64 * TN), bias, res, relu, mask);
}

// 128x128 tiles with a 64x64 TAIL.  A launch whose tile count is a little above a whole number of rounds (1024 resident workgroups) ends
// with a few 128x128 tiles running alone on their CUs at half the matrix-pipe rate while the other CUs idle -- 256->256 at 14x14, B = 1024:
// 3136 tiles = 3 rounds + 64 tiles, 256 us of tail in a 1.79 ms launch.  Here the rows past the last whole round are cut into 64x64
// tiles (a quarter of the work each, four times as many): the same blocks of the grid, same arithmetic per output element.
__global__ __launch_bounds__(256, ISX_WG_PER_CU_128) void conv3x3_tail_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ Wt, int64_t N, Conv3x3Geom g,
                                                              float* __restrict__ C, TileMap tm_big, TileMap tm_small, int64_t m_split,
                                                              const float* __restrict__ bias, const float* __restrict__ res, int relu) {
    constexpr int kBig = 16 * (128 + 128 + 2 * lds_pad(16));
    __shared__ float lds[kBig > kTailLdsFloats ? kBig : kTailLdsFloats];
    const int nbig = tm_big.tiles_m * tm_big.tiles_n;                                // a multiple of 8: the XCD of a block is the same in both numberings
    int tile_m, tile_n;
    if ((int)blockIdx.x < nbig) {
        tile_of_block(tm_big, tile_m, tile_n, (int)blockIdx.x, nbig);
        conv3x3_tile<2, 2, 16>(lds, x, M, Wt, N, g, C, (int64_t)tile_m * 128, (int64_t)tile_n * 128, bias, res, relu);
    } else {
        tile_of_block(tm_small, tile_m, tile_n, (int)blockIdx.x - nbig, tm_small.tiles_m * tm_small.tiles_n);
        conv3x3_tile<ISX_TAIL_TM, 1, 32>(lds, x, M, Wt, N, g, C, m_split + (int64_t)tile_m * (64 * ISX_TAIL_TM), (int64_t)tile_n * 64, bias, res, relu);
    }
}

template <int TM, int TN, int BK>
static void launch_conv3x3(const float* x, int64_t M, const float* w, int64_t N, const Conv3x3Geom& g, float* y, const float* bias,
                           const float* res, int relu, hipStream_t st, const float* mask = nullptr) {
    TileMap tm;
    tm.m_active = nullptr;
    tm.tiles_m = (int)((M + 64 * TM - 1) / (64 * TM));
    tm.tiles_n = (int)((N + 64 * TN - 1) / (64 * TN));
    const int64_t split = (TM == 2 && TN == 2 && !mask) ? gemm_tail_split_rows(M, N, 256 * ISX_WG_PER_CU_128) : 0;
    if (split > 0) {
        TileMap small;
        small.m_active = nullptr;
        tm.tiles_m = (int)(split / 128);
        small.tiles_m = (int)((M - split + 64 * ISX_TAIL_TM - 1) / (64 * ISX_TAIL_TM));
        small.tiles_n = (int)((N + 63) / 64);
        hipLaunchKernelGGL(conv3x3_tail_kernel, dim3((unsigned)(tm.tiles_m * tm.tiles_n + small.tiles_m * small.tiles_n)), dim3(256), 0, st, x, M, w, N, g, y,
                           tm, small, split, bias, res, relu);
        return;
    }
    if (mask) hipLaunchKernelGGL((conv3x3_nhwc_kernel<TM, TN, BK, true>), dim3((unsigned)(tm.tiles_m * tm.tiles_n)), dim3(256), 0, st, x, M, w, N, g, y, tm, bias,
                                 res, relu, mask);
    else hipLaunchKernelGGL((conv3x3_nhwc_kernel<TM, TN, BK, false>), dim3((unsigned)(tm.tiles_m * tm.tiles_n)), dim3(256), 0, st, x, M, w, N, g, y, tm, bias,
                            res, relu, mask);
}


// ---- last convolution of a bottleneck block WITH its projection shortcut, as ONE GEMM -------------------------
//   y = relu( W3 . t  +  Wd . x_s  +  (b3 + bd) )  =  relu( [W3 | Wd] . [t ; x_s] + b )
// t: (M, K1) the block's 3x3 output, x: the block input (B,H,W,K2) sampled with the projection's stride (1x1, no
// padding), weights concatenated along K.  The shortcut tensor is never written or read back (a (M, Cout) round trip:
// 6.6 GB in layer 1 at B = 1024) and one launch disappears.  K order: all of t's channels, then all of x's -- one fp32
// fma chain per output, restated by the oracle.
struct DualGeom { int H, W, Ho, Wo, stride, K1, K2; };

template <int TM, int TN, int BK>
__device__ __forceinline__ void conv1x1_dual_tile(float* __restrict__ lds, const float* __restrict__ t, const float* __restrict__ x, int64_t M,
                                                  const float* __restrict__ Wt, int64_t N, const DualGeom& g, float* __restrict__ C, int64_t m0, int64_t n0,
                                                  const float* __restrict__ bias, int relu) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + lds_pad(BK), LDB = BN + lds_pad(BK);
    constexpr int CH = BK / 4, NA = BM * CH / 256;
    float* As = lds;
    float* Bs = lds + BK * LDA;
    const int D = g.K1 + g.K2;
    const int64_t ldc = N;

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
    f32x16 tot[kConvChunk ? TM : 1][kConvChunk ? TN : 1];     // second accumulator of the two-level sum (gemm_tile.hpp)
    zero_tiles(tot);

    const int c4 = (threadIdx.x % CH) << 2;
    // A rows through buffer loads (as load_tile / conv3x3): descriptor 1 = the tile's rows of t, descriptor 2 = the block input from the
    // first pixel the tile samples; one 32-bit offset per staged row and source, the channel offset inside the source as SGPR offset
    const int64_t mleft = M - m0;
    const auto r1 = uniform_rsrc(t + m0 * g.K1, (mleft < BM ? mleft : BM) * (int64_t)g.K1 * 4);
    const int hw_ = g.Ho * g.Wo;
    auto pixel_of = [&](int64_t m) {
        const int b = (int)(m / hw_), rem = (int)(m - (int64_t)b * hw_);
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        return (int64_t)(b * g.H + ho * g.stride) * g.W + wo * g.stride;
    };
    const int64_t base_pix = pixel_of(m0 < M ? m0 : M - 1);
    const auto r2 = uniform_rsrc(x + base_pix * g.K2, ((int64_t)(M / hw_) * g.H * g.W - base_pix) * g.K2 * 4);
    unsigned vo1[NA], vo2[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (j * 256 + threadIdx.x) / CH;
        int64_t m = m0 + row;
        m = m < M ? m : M - 1;                    // rows past the edge repeat the last one (never stored)
        vo1[j] = (unsigned)((m - m0) * g.K1 + c4) * 4u;
        vo2[j] = (unsigned)((pixel_of(m) - base_pix) * g.K2 + c4) * 4u;
    }
    int kdone = 0;                                // channels already issued (uniform)
    float4 ra[NA], rb[BN * BK / 1024];
    auto load_a = [&]() {
        if (kdone < g.K1) {                       // uniform
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r1, vo1[j], (unsigned)kdone * 4u, 0));
        } else {                                  // the block input, sampled with the projection's stride
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r2, vo2[j], (unsigned)(kdone - g.K1) * 4u, 0));
        }
        kdone += BK;
    };
    const int nk = D / BK;
    load_a();
    load_tile<true, BN, BK>(Wt, N, D, n0, 0, rb);
    store_tile<BM, BK>(As, ra);
    store_tile<BN, BK>(Bs, rb);
    __syncthreads();

    const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
    const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
    constexpr bool PINNED = kConvChunk != 0 && TM * TN == 4 && ISX_PIN_KTILE;
    KtilePtrs<BK> pins;
    if constexpr (PINNED) pins = pin_ktile_ptrs<BK, LDA, LDB>(a_base, b_base);
    // outer loop: chunks of the two-level sum over the flattened [t ; x] reduction; inner loop: the staged k-tiles of a chunk (gemm_tile.hpp); the
    // first k-tile of a chunk starts its chains with C = 0
    f32x16 (*totp)[TN] = nullptr;
    if constexpr (kConvChunk != 0) totp = tot;
    auto body = [&](int kt, auto zero_c) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_a();
            load_tile<true, BN, BK>(Wt, N, D, n0, (kt + 1) * BK, rb);
        }
        mfma_ktile_sel<TM, TN, BK, LDA, LDB, PINNED, decltype(zero_c)::value>(a_base, b_base, pins, acc, totp);
        __syncthreads();
        if (more) {
            store_tile<BM, BK>(As, ra);
            store_tile<BN, BK>(Bs, rb);
            __syncthreads();
        }
    };
    if constexpr (kConvChunk == 0) {
        for (int kt = 0; kt < nk; ++kt) body(kt, std::false_type());
    } else {
        for (int kt = 0; kt < nk;) {
            const int kend = kt + kConvChunk / BK < nk ? kt + kConvChunk / BK : nk;
            body(kt++, std::true_type());                      // (interleaved fold: adds the PREVIOUS chunk's chain in front of its C = 0 MFMAs)
            for (; kt < kend; ++kt) body(kt, std::false_type());
            if (!((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2)) add_chunk<TM, TN>(tot, acc);
        }
        if ((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2) add_chunk<TM, TN>(tot, acc);       // the last chunk
    }
    if constexpr (kConvChunk != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
    }

    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    conv_epilogue_buffers<TM, TN>(acc, C, nullptr, bias, relu, m0, M, n0, N, ldc, BM, wm_u * (32 * TM), wn_u * (32 * TN), l31, half);
}

template <int TM, int TN, int BK>
__global__ __launch_bounds__(256, TM * TN == 1 ? 6 : TM * TN == 4 ? ISX_WG_PER_CU_128 : 4) void conv1x1_dual_nhwc_kernel(const float* __restrict__ t, const float* __restrict__ x, int64_t M,
                                                                                      const float* __restrict__ Wt, int64_t N, DualGeom g,
                                                                                      float* __restrict__ C, TileMap tm,
                                                                                      const float* __restrict__ bias, int relu) {
    __shared__ float lds[BK * (64 * TM + 64 * TN + 2 * lds_pad(BK))];
    int tile_m, tile_n;
    tile_of_block(tm, tile_m, tile_n);
    conv1x1_dual_tile<TM, TN, BK>(lds, t, x, M, Wt, N, g, C, (int64_t)tile_m * (64 * TM), (int64_t)tile_n * (64 * TN), bias, relu);
}

// 128x128 tiles + 64x64 tail in one grid (see conv3x3_tail_kernel)
__global__ __launch_bounds__(256, ISX_WG_PER_CU_128) void conv1x1_dual_tail_kernel(const float* __restrict__ t, const float* __restrict__ x, int64_t M, const float* __restrict__ Wt,
                                                                   int64_t N, DualGeom g, float* __restrict__ C, TileMap tm_big, TileMap tm_small,
                                                                   int64_t m_split, const float* __restrict__ bias, int relu) {
    constexpr int kBig = 16 * (128 + 128 + 2 * lds_pad(16));
    __shared__ float lds[kBig > kTailLdsFloats ? kBig : kTailLdsFloats];
    const int nbig = tm_big.tiles_m * tm_big.tiles_n;
    int tile_m, tile_n;
    if ((int)blockIdx.x < nbig) {
        tile_of_block(tm_big, tile_m, tile_n, (int)blockIdx.x, nbig);
        conv1x1_dual_tile<2, 2, 16>(lds, t, x, M, Wt, N, g, C, (int64_t)tile_m * 128, (int64_t)tile_n * 128, bias, relu);
    } else {
        tile_of_block(tm_small, tile_m, tile_n, (int)blockIdx.x - nbig, tm_small.tiles_m * tm_small.tiles_n);
        conv1x1_dual_tile<ISX_TAIL_TM, 1, 32>(lds, t, x, M, Wt, N, g, C, m_split + (int64_t)tile_m * (64 * ISX_TAIL_TM), (int64_t)tile_n * 64, bias, relu);
    }
}

template <int TM, int TN, int BK>
static void launch_dual(const float* t, const float* x, int64_t M, const float* w, int64_t N, const DualGeom& g, float* y, const float* bias, int relu,
                        hipStream_t st) {
    TileMap tm;
    tm.m_active = nullptr;
    tm.tiles_m = (int)((M + 64 * TM - 1) / (64 * TM));
    tm.tiles_n = (int)((N + 64 * TN - 1) / (64 * TN));
    const int64_t split = (TM == 2 && TN == 2) ? gemm_tail_split_rows(M, N, 256 * ISX_WG_PER_CU_128) : 0;
    if (split > 0) {
        TileMap small;
        small.m_active = nullptr;
        tm.tiles_m = (int)(split / 128);
        small.tiles_m = (int)((M - split + 64 * ISX_TAIL_TM - 1) / (64 * ISX_TAIL_TM));
        small.tiles_n = (int)((N + 63) / 64);
        hipLaunchKernelGGL(conv1x1_dual_tail_kernel, dim3((unsigned)(tm.tiles_m * tm.tiles_n + small.tiles_m * small.tiles_n)), dim3(256), 0, st, t, x, M, w, N, g,
                           y, tm, small, split, bias, relu);
        return;
    }
    hipLaunchKernelGGL((conv1x1_dual_nhwc_kernel<TM, TN, BK>), dim3((unsigned)(tm.tiles_m * tm.tiles_n)), dim3(256), 0, st, t, x, M, w, N, g, y, tm,
                       bias, relu);
}

static std::atomic<int> g_force_conv_cfg{[] { const char* e = getenv("ISX_DEBUG_CONV_CFG"); return e ? atoi(e) : -1; }()};       // debug / A-B hook

}  // namespace isx

using namespace isx;

// 1x1 stride-1 convolution on NHWC activations: one GEMM over the pixels with the bias / residual / ReLU epilogue fused
// (the backbone layers of model/ModelDefinition.py's torchvision ResNets inside `features`, model/siamese.py:20,107,151).
ISX_API int isx_conv1x1_nhwc(const float* x, int64_t M, int Cin, const float* w, int Cout, const float* bias, const float* residual,
                             int relu, float* y, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && Cin > 0 && Cout > 0 && Cout <= (1 << 20), "isx_conv1x1_nhwc: bad shape M=%lld Cin=%d Cout=%d (Cout <= 2^20: 32-bit offsets inside a tile's rows)", (long long)M, Cin, Cout);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(x && w && bias && y, "isx_conv1x1_nhwc: null pointer");
    ISX_REQUIRE(y != x && y != residual, "isx_conv1x1_nhwc: y must not alias x or residual");
    // first-stage layers at 56x56 (Cin 64 / 256, HBM-bound): streaming kernel (stream1x1.hip), same arithmetic; debug cfg 9 = general path (A/B)
    if (g_force_conv_cfg != 9 && conv1x1_stream_applicable(M, Cin, Cout, x, residual))
        return launch_conv1x1_stream(x, M, w, Cin, Cout, bias, residual, relu ? 1 : 0, y, (hipStream_t)stream);
    return launch_conv1x1_gemm(x, M, w, Cout, Cin, y, bias, residual, relu ? 1 : 0, (hipStream_t)stream);
}

// 3x3 convolution, padding 1, stride 1 or 2, NHWC activations, weights (Cout,3,3,Cin), bias / residual / ReLU fused.
ISX_API int isx_conv3x3_nhwc(const float* x, int64_t B, int H, int W, int Cin, const float* w_ohwi, int Cout, int stride,
                             const float* bias, const float* residual, int relu, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (stride == 1 || stride == 2),
                "isx_conv3x3_nhwc: bad shape B=%lld H=%d W=%d Cin=%d Cout=%d stride=%d", (long long)B, H, W, Cin, Cout, stride);
    ISX_REQUIRE(Cin % 32 == 0, "isx_conv3x3_nhwc: Cin=%d must be a multiple of 32", Cin);
    ISX_REQUIRE(Cout <= (1 << 20), "isx_conv3x3_nhwc: Cout=%d above 2^20 (32-bit offsets inside a tile's rows)", Cout);
    ISX_REQUIRE(H < 32767 && W < 32767 && B * H * W < (1ll << 31), "isx_conv3x3_nhwc: input has too many pixels for 32-bit pixel indices");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(x && w_ohwi && bias && y, "isx_conv3x3_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)x | (uintptr_t)w_ohwi) % 16) == 0, "isx_conv3x3_nhwc: x and w must be 16-B aligned");
    ISX_REQUIRE(y != x && y != residual, "isx_conv3x3_nhwc: y must not alias x or residual");
    Conv3x3Geom g;
    g.H = H; g.W = W; g.Cin = Cin; g.stride = stride;
    g.Ho = (H - 1) / stride + 1;
    g.Wo = (W - 1) / stride + 1;
    const int64_t M = B * g.Ho * g.Wo, N = Cout;
    ISX_REQUIRE(((M + 63) / 64) * ((N + 63) / 64) < (1ll << 31), "isx_conv3x3_nhwc: too many tiles");
    // measured on the ResNet-50 shapes at B = 1024 (ms, 128x128 / 128x64 / 64x64): 64->64 @56 3.70 / 2.10 / 2.04,
    // 128->128 @28 1.89 / 1.98 / 1.97, 256->256 @14 1.93 / 1.98 / 1.91, 512->512 @7 2.08 / 2.04 / 1.94
    // the round / tail model of the GEMMs with this kernel's efficiencies (B = 1024, ms for 128x128 / 128x64 / 64x64: 128->128 at 28x28
    // 1.71 / 1.78 / 1.76; 64->64 at 56x56 - / 1.89 / 1.85): 128x128 (+ 64x64 tail) wherever the grid fills the chip, the shape with the
    // fewest idle CUs below that (512->512 at 14x14 with 64 images: 1568 tiles of 64x64 are 1.02 rounds, 784 of 128x64 are 0.77)
    static const float eff3x3[4] = {0.90f, 0.0f, 0.865f, 0.87f};
    int best = pick_tile_cfg(M, N, gemm_tail_split_rows(M, N, 256 * ISX_WG_PER_CU_128), eff3x3, 0xD, ISX_WG_PER_CU_128);
    { const int fc_ = g_force_conv_cfg; if (fc_ == 0 || fc_ == 2 || fc_ == 3) best = fc_; }
    hipStream_t st = (hipStream_t)stream;
    switch (best) {
        case 0: launch_conv3x3<2, 2, 16>(x, M, w_ohwi, N, g, y, bias, residual, relu ? 1 : 0, st); break;
        case 2: launch_conv3x3<2, 1, 32>(x, M, w_ohwi, N, g, y, bias, residual, relu ? 1 : 0, st); break;
        default: launch_conv3x3<1, 1, 32>(x, M, w_ohwi, N, g, y, bias, residual, relu ? 1 : 0, st); break;
    }
    ISX_CHECK_LAUNCH("isx_conv3x3_nhwc");
    return ISX_OK;
}

// Gradient of a 3x3 convolution (padding 1) wrt its input, as a stride-1 3x3 convolution of dz with the TRANSPOSED, FLIPPED weight
// wt[ci][kh][kw][co] = w'[co][2-kh][2-kw][ci] (a stride-2 layer hands in dz zero-upsampled to the input grid), with the ReLU of the layer
// below fused: dx = conv(dz, wt) . [mask > 0].  dz: (B,H,W,Cout), wt: (Cin,3,3,Cout), mask / dx: (B,H,W,Cin).  Cout % 32 == 0.
ISX_API int isx_conv3x3_dgrad_nhwc(const float* dz, int64_t B, int H, int W, int Cout, const float* wt, int Cin, const float* mask, float* dx,
                                   isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "isx_conv3x3_dgrad_nhwc: bad shape B=%lld H=%d W=%d Cout=%d Cin=%d", (long long)B, H, W, Cout, Cin);
    ISX_REQUIRE(Cout % 32 == 0, "isx_conv3x3_dgrad_nhwc: Cout=%d must be a multiple of 32", Cout);
    ISX_REQUIRE(Cin <= (1 << 20) && H < 32767 && W < 32767 && B * H * W < (1ll << 31), "isx_conv3x3_dgrad_nhwc: shape out of the 32-bit index range");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(dz && wt && dx, "isx_conv3x3_dgrad_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)dz | (uintptr_t)wt) % 16) == 0, "isx_conv3x3_dgrad_nhwc: dz and wt must be 16-B aligned");
    ISX_REQUIRE(dx != dz && dx != mask, "isx_conv3x3_dgrad_nhwc: dx must not alias an input");
    Conv3x3Geom g;
    g.H = H; g.W = W; g.Cin = Cout; g.stride = 1; g.Ho = H; g.Wo = W;
    const int64_t M = B * H * W, N = Cin;
    static const float eff3x3[4] = {0.90f, 0.0f, 0.865f, 0.87f};
    const int best = pick_tile_cfg(M, N, 0, eff3x3, 0xD);
    hipStream_t st = (hipStream_t)stream;
    switch (best) {
        case 0: launch_conv3x3<2, 2, 16>(dz, M, wt, N, g, dx, nullptr, nullptr, 0, st, mask); break;
        case 2: launch_conv3x3<2, 1, 32>(dz, M, wt, N, g, dx, nullptr, nullptr, 0, st, mask); break;
        default: launch_conv3x3<1, 1, 32>(dz, M, wt, N, g, dx, nullptr, nullptr, 0, st, mask); break;
    }
    ISX_CHECK_LAUNCH("isx_conv3x3_dgrad_nhwc");
    return ISX_OK;
}

// Last 1x1 convolution of a bottleneck block fused with its 1x1 projection shortcut (see conv1x1_dual_nhwc_kernel).
ISX_API int isx_conv1x1_dual_nhwc(const float* t, int K1, const float* x, int64_t B, int H, int W, int K2, int stride, const float* w_cat,
                                  int Cout, const float* bias, int relu, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && K1 > 0 && K2 > 0 && Cout > 0 && (stride == 1 || stride == 2),
                "isx_conv1x1_dual_nhwc: bad shape B=%lld H=%d W=%d K1=%d K2=%d Cout=%d stride=%d", (long long)B, H, W, K1, K2, Cout, stride);
    ISX_REQUIRE(K1 % 32 == 0 && K2 % 32 == 0, "isx_conv1x1_dual_nhwc: K1=%d and K2=%d must be multiples of 32", K1, K2);
    ISX_REQUIRE(Cout <= (1 << 20), "isx_conv1x1_dual_nhwc: Cout=%d above 2^20 (32-bit offsets inside a tile's rows)", Cout);
    ISX_REQUIRE(B * H * W < (1ll << 31), "isx_conv1x1_dual_nhwc: input has too many pixels for 32-bit pixel indices");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(t && x && w_cat && bias && y, "isx_conv1x1_dual_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)t | (uintptr_t)x | (uintptr_t)w_cat) % 16) == 0, "isx_conv1x1_dual_nhwc: t, x and w must be 16-B aligned");
    ISX_REQUIRE(y != t && y != x, "isx_conv1x1_dual_nhwc: y must not alias an input");
    DualGeom g;
    g.H = H; g.W = W; g.stride = stride; g.K1 = K1; g.K2 = K2;
    g.Ho = (H - 1) / stride + 1;
    g.Wo = (W - 1) / stride + 1;
    const int64_t M = B * g.Ho * g.Wo, N = Cout;
    ISX_REQUIRE(((M + 63) / 64) * ((N + 63) / 64) < (1ll << 31), "isx_conv1x1_dual_nhwc: too many tiles");
    // measured at B = 1024 (ms, 128x128 / 128x64 / 64x64): layer 1 2.10 / 2.22 / 2.21, layer 2 2.64 / 2.68 / 2.84,
    // layer 3 2.54 / 2.54 / 2.75, layer 4 (50 k pixels) 2.55 / 2.48 / 2.72
    static const float eff_dual[4] = {0.92f, 0.0f, 0.885f, 0.85f};                         // layer 1-4 at B = 1024: 128x128 best, then 128x64, then 64x64
    int best = pick_tile_cfg(M, N, gemm_tail_split_rows(M, N, 256 * ISX_WG_PER_CU_128), eff_dual, 0xD, ISX_WG_PER_CU_128);
    { const int fc_ = g_force_conv_cfg; if (fc_ == 0 || fc_ == 2 || fc_ == 3) best = fc_; }
    hipStream_t st = (hipStream_t)stream;
    if (best == 0) launch_dual<2, 2, 16>(t, x, M, w_cat, N, g, y, bias, relu ? 1 : 0, st);
    else if (best == 2) launch_dual<2, 1, 32>(t, x, M, w_cat, N, g, y, bias, relu ? 1 : 0, st);
    else launch_dual<1, 1, 32>(t, x, M, w_cat, N, g, y, bias, relu ? 1 : 0, st);
    ISX_CHECK_LAUNCH("isx_conv1x1_dual_nhwc");
    return ISX_OK;
}

// Debug / A-B hook (not declared in include/isx.h): force the conv3x3 tile shape (0, 2, 3), -1 = automatic.
ISX_API void isx_debug_set_conv_cfg(int c) {
    isx::g_tail_split = (c != 7);        // 7 = automatic tile choice WITHOUT the 64x64 tails (A/B)
    g_force_conv_cfg = (c == 7) ? -1 : c;
}

