// expand_kernel.hpp -- the fused conv2 + conv3 kernel of expand.hip (a header so that scratch/lab/expand_lab.hip can instantiate it with phase stamps).
#pragma once
#include "conv3x3_tile.hpp"

namespace isx {

// ---- 3x3 convolution to 64 channels + the 1x1 expansion behind it, as ONE kernel ------------------------------
//   y = act3( W3 . relu(conv3x3(x, W2) + b2) + b3 (+ residual) )        (conv2 + conv3 of a torchvision Bottleneck with 64 mid channels)
// A 64-pixel tile of the 3x3 convolution holds ALL 64 mid channels of its pixels = a complete A tile of the 1x1 expansion: the wave
// accumulators get bias + ReLU, go to the LDS (K-major, in place of the operand stages) and feed a second MFMA loop against W3 (given
// TRANSPOSED, (64, Cout): the B operands are coalesced buffer loads that hit the L2, offsets as SGPRs).  The mid activation (0.8 GB at
// 56x56, B = 1024) is neither written nor read back, and the HBM-bound expansion (7.4 GB for 105 GFLOP) runs inside an MFMA-bound kernel.
// Same arithmetic per element as isx_conv3x3_nhwc followed by isx_conv1x1_nhwc: the mid values are the fp32 numbers that path stores.
// DUAL: the first block of the stage, whose shortcut is a 1x1 projection of the block input x2 (64 channels, same pixels: stride 1):
//   y = act( [W3 | Wd] . [relu(conv3x3(x, W2) + b2) ; x2] + b ),  W3t = the concatenated weight transposed, (128, Cout);
// the x2 rows of the tile are fetched at kernel start, wait in registers during the 3x3 loop and go to a second LDS tile.
// Two-level sum (gemm_tile.hpp): the 3x3 loop folds its chain every 64 terms (conv3x3_mainloop); the expansion is ONE chunk (64 mid channels), with
// DUAL two -- the chain over the mid channels, then the chain over the x2 channels, added: that kernel holds two accumulator sets (2 x 64 VGPRs)
// and runs two workgroups per CU (one launch per trunk: the first block of stage 1).
template <int TN2, bool DUAL, bool STAMPS = false>
__global__ __launch_bounds__(256, DUAL ? 2 : 4) void conv3x3_expand_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ W2, Conv3x3Geom g,
                                                                const float* __restrict__ b2, const float* __restrict__ W3t, const float* __restrict__ b3,
                                                                const float* __restrict__ res, int relu, float* __restrict__ y,
                                                                unsigned long long* __restrict__ stamps = nullptr) {
    constexpr int COUT = 64 * TN2, LDY = 64 + 1, TILE_F = 32 * (64 + 64 + 2 * lds_pad(32));
    __shared__ float lds[(DUAL ? 2 : 1) * TILE_F];                          // 4160 floats: the 3x3 operand stages, then the 64 x 65 mid tile (+ the x2 tile)
    static_assert(TILE_F >= 64 * LDY, "mid tile must fit the operand stages");
    // XCD-aware order: XCD x gets a contiguous range of pixel tiles (neighbouring tiles share their halo rows in its L2)
    const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int64_t m0 = (int64_t)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3)) * 64;

    // STAMPS (scratch/lab/expand_lab.hip only): shader-clock stamps of wave 0 at the phase boundaries, 8 per workgroup
    auto stamp = [&](int i) { if (STAMPS && threadIdx.x == 0) stamps[(int64_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    float4 x2r[4];
    if (DUAL) {                                                             // rows m0 .. m0 + 63 of x2 (= res): 16 chunks of 16 B each; rows past M read zeros
        const int64_t left = M - m0;
        const auto xr2 = uniform_rsrc(res + m0 * 64, (left < 64 ? left : 64) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) x2r[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr2, (unsigned)((j * 256 + (int)threadIdx.x) * 16), 0, 0));
    }
    f32x16 acc[1][1];
    conv3x3_mainloop<1, 1, 32, true>(lds, x, M, W2, 64, g, m0, 0, acc);
    stamp(1);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    if (DUAL) {                                                             // x2 tile -> second LDS tile, K-major X[k][pixel]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 256 + (int)threadIdx.x, row = idx >> 4, k = (idx & 15) << 2;
            float* d = lds + TILE_F + k * LDY + row;
            d[0] = x2r[j].x; d[LDY] = x2r[j].y; d[2 * LDY] = x2r[j].z; d[3 * LDY] = x2r[j].w;
        }
    }
    {   // mid tile -> LDS, K-major: Y[k = mid channel][pixel]; lanes of a half-wave write consecutive k (stride 65: conflict-free)
        const float bv = b2[wn * 32 + l31];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            lds[(wn * 32 + l31) * LDY + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = fmaxf(acc[0][0][e] + bv, 0.0f);
    }
    __syncthreads();
    stamp(2);

    // expansion: wave w = all 64 pixels x output channels 16 TN2 w .. (TN2 / 2 column blocks): per k-step (mid channels 2s, 2s + 1) two A reads
    // from the LDS, TN2 / 2 coalesced B loads from the L2 and TN2 MFMAs; the four waves read disjoint quarters of W3
    constexpr int NJ = TN2 / 2;
    f32x16 acc2[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.0f;
    const float* a_base = lds + half * LDY + l31;
    const auto wr = uniform_rsrc(W3t, (int64_t)(DUAL ? 128 : 64) * COUT * 4);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned wvo = (unsigned)((half * COUT + l31) * 4);
    const unsigned wso = (unsigned)(wave_u * 32 * NJ * 4);
    f32x16 acc3[DUAL ? 2 : 1][DUAL ? NJ : 1];                                   // DUAL: chain over the x2 channels (second chunk)
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc3[i][j][e] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < (DUAL ? 64 : 32); ++s) {
        const int ao = s < 32 ? 2 * s * LDY : TILE_F + 2 * (s - 32) * LDY;      // mid channels, then the x2 channels
        const float a0 = a_base[ao], a1 = a_base[ao + 32];
        float bq[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            bq[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wvo, wso + (unsigned)((2 * s * COUT + 32 * j) * 4), 0));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (DUAL && s >= 32) {
                acc3[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[j], acc3[0][j], 0, 0, 0);
                acc3[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[j], acc3[1][j], 0, 0, 0);
            } else {
                acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[j], acc2[0][j], 0, 0, 0);
                acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[j], acc2[1][j], 0, 0, 0);
            }
        }
    }
    if (DUAL) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc2[i][j] = acc2[i][j] + acc3[i][j];       // tot = (0 + chain_0) + chain_1
    }
    stamp(3);
    conv_epilogue_buffers<2, NJ, 2>(acc2, y, DUAL ? nullptr : res, b3, relu, m0, M, 0, COUT, COUT, 64, 0, wave_u * (32 * NJ), l31, half);
    if (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(4); }
}

// ---- the same fusion on a 256-PIXEL tile (round 4) ------------------------------------------------------------------------------
// The 64-pixel kernel above gives every wave ONE accumulator in the 3x3 loop: all its MFMAs depend on each other and the matrix pipe is fed
// by wave switches (0.84 busy; the kernels with four independent accumulators per wave reach 0.90).  Here the four waves are stacked along the
// pixels: wave w owns pixels 64 w .. 64 w + 63 of the tile and ALL 64 mid channels = 2 x 2 independent accumulator tiles, the weight k-tile in
// the LDS is shared by the four waves (a quarter of the B traffic per MFMA).  A wave's mid tile is exactly the A operand of ITS OWN rows of the
// expansion, so it goes to a wave-private LDS region and the 1x1 expansion needs no workgroup barrier: four passes of 64 output channels, 2 x 2
// accumulators each, the epilogue of pass p in flight while other waves run their MFMAs.  Same fma chain per output element as the 64-pixel
// kernel (and as conv3x3 followed by conv1x1).  LDS: max(operand stages 41 KB, four mid tiles 66.6 KB): two workgroups per CU.
template <int TN2>
__global__ __launch_bounds__(256, 2) void conv3x3_expand256_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ W2, Conv3x3Geom g,
                                                                   const float* __restrict__ b2, const float* __restrict__ W3t, const float* __restrict__ b3,
                                                                   const float* __restrict__ res, int relu, float* __restrict__ y) {
    constexpr int COUT = 64 * TN2, LDY = 64 + 1, STAGES_F = 32 * (256 + 64 + 2 * lds_pad(32)), MID_F = 4 * 64 * LDY;
    __shared__ float lds[STAGES_F > MID_F ? STAGES_F : MID_F];
    const int nwg = (int)gridDim.x, b = (int)blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int64_t m0 = (int64_t)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3)) * 256;

    f32x16 acc[2][2];
    conv3x3_mainloop<2, 2, 32, false, 4>(lds, x, M, W2, 64, g, m0, 0, acc);          // every wave has left the operand stages when this returns

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float* mid = lds + wave_u * (64 * LDY);                                          // this wave's Y[k = mid channel][pixel], K-major
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bv = b2[j * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                mid[(j * 32 + l31) * LDY + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = fmaxf(acc[i][j][e] + bv, 0.0f);
    }
    __syncthreads();                                                                 // (only the wave's own region is read below; one cheap barrier keeps the LDS ordering explicit)

    const float* a_base = mid + half * LDY + l31;
    const auto wr = uniform_rsrc(W3t, (int64_t)64 * COUT * 4);
    const unsigned wvo = (unsigned)((half * COUT + l31) * 4);
#pragma unroll 1
    for (int p = 0; p < COUT / 64; ++p) {
        f32x16 acc2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float a0 = a_base[2 * s * LDY], a1 = a_base[2 * s * LDY + 32];
            const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wvo, (unsigned)((2 * s * COUT + p * 64) * 4), 0));
            const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, wvo, (unsigned)((2 * s * COUT + p * 64 + 32) * 4), 0));
            acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q0, acc2[0][0], 0, 0, 0);
            acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q1, acc2[0][1], 0, 0, 0);
            acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q0, acc2[1][0], 0, 0, 0);
            acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q1, acc2[1][1], 0, 0, 0);
        }
        conv_epilogue_buffers<2, 2, 1>(acc2, y, res, b3, relu, m0, M, 0, COUT, COUT, 256, wave_u * 64, p * 64, l31, half);
    }
}

}  // namespace isx
