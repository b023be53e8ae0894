// conv3x3_tile.hpp -- the implicit-GEMM main loop of the 3x3 convolution (shared by conv.hip and expand.hip).
#pragma once
#include "gemm_tile.hpp"

namespace isx {

// ---- 3x3 convolution (padding 1, stride 1 or 2) on NHWC activations as an IMPLICIT GEMM ----------------------
// Same tile machinery as cosine_gemm_kernel: M = B*Ho*Wo output pixels, N = Cout, K = 9*Cin ordered (kh, kw, ci),
// weights pre-arranged (Cout, 3, 3, Cin).  A k-tile lies inside one filter tap (Cin % BK == 0), so the A rows of a
// k-tile are the input pixels shifted by that tap: one base pixel per staged row, kept in registers, plus a
// bounds test per tap (padding rows load zeros -- fma(0, w, acc) leaves acc unchanged, as skipping the tap would).
// Epilogue: bias (+ residual) + ReLU fused, wave-uniform row pointers.  Replaces conv2 of the torchvision
// Bottleneck / both convolutions of BasicBlock inside the `features` trunk.
struct Conv3x3Geom { int H, W, Cin, Ho, Wo, stride; };

// accumulators of one (64 TM) x (64 TN) output tile at rows m0.., columns n0.. (every wave has left the LDS when this returns);
// lds: BK * (64 TM + 64 TN + 2 pads) floats
// AHEAD2 (64x64 tiles only): operands requested two k-tiles ahead instead of one (16 more VGPRs: the fused expand kernel has them, the plain
// 64x64 kernel at six workgroups per CU does not).
// WM: waves along M (2: the 2x2 arrangement of every other kernel; 4: four waves stacked along the pixels, each TM x TN tiles of the FULL width)
// CHUNK: terms per first-level chain of the two-level sum (gemm_tile.hpp; the inference trunk), 0 = one chain over all 9 Cin terms (gradients)
template <int TM, int TN, int BK, bool AHEAD2 = false, int WM = 2, int CHUNK = kConvChunk>
__device__ __forceinline__ void conv3x3_mainloop(float* __restrict__ lds, const float* __restrict__ x, int64_t M, const float* __restrict__ Wt, int64_t N,
                                                 const Conv3x3Geom& g, int64_t m0, int64_t n0, f32x16 (&acc)[TM][TN]) {
    constexpr int WN = 4 / WM, BM = 32 * TM * WM, BN = 32 * TN * WN, LDA = BM + lds_pad(BK), LDB = BN + lds_pad(BK);
    constexpr int CH = BK / 4, NA = BM * CH / 256;
    float* As = lds;
    float* Bs = lds + BK * LDA;
    const int D = 9 * g.Cin;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    f32x16 tot[CHUNK ? TM : 1][CHUNK ? TN : 1];
    zero_tiles(tot);

    // staged A rows of this thread: top-left input pixel of the 3x3 window (may be -1: padding)
    int pbase[NA], hw0[NA];                       // pixel index of (hi0, wi0); (hi0 + 1) << 16 | (wi0 + 1)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int idx = j * 256 + threadIdx.x;
        int64_t m = m0 + idx / CH;
        m = m < M ? m : M - 1;
        const int hw = g.Ho * g.Wo;
        const int b = (int)(m / hw), rem = (int)(m - (int64_t)b * hw);
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        const int hi0 = ho * g.stride - 1, wi0 = wo * g.stride - 1;
        pbase[j] = (b * g.H + hi0) * g.W + wi0;
        hw0[j] = ((hi0 + 1) << 16) | (wi0 + 1);
    }
    const int c4 = (threadIdx.x % CH) << 2;
    int kh = 0, kw = 0, ci0 = 0;                  // tap / channel offset of the NEXT k-tile to load (uniform)
    float4 ra[NA], rb[BN * BK / 1024];
    // A rows come through BUFFER loads: a wave-uniform descriptor that starts at the first input pixel this tile can touch, one 32-bit
    // byte offset per staged row (recomputed once per filter tap), the channel offset inside the tap as the SGPR offset.  A padding
    // tap gets an offset outside the descriptor and loads zeros: no per-k-tile address arithmetic, no select on the loaded values.
    int64_t mf = m0 < M ? m0 : M - 1;
    const int hw_ = g.Ho * g.Wo;
    const int bf = (int)(mf / hw_), remf = (int)(mf - (int64_t)bf * hw_);
    const int hof = remf / g.Wo, wof = remf - hof * g.Wo;
    int64_t base_pix = ((int64_t)bf * g.H + (hof * g.stride - 1)) * g.W + (wof * g.stride - 1);      // top-left tap of the tile's first row
    base_pix = base_pix > 0 ? base_pix : 0;
    const int64_t left = ((int64_t)(M / hw_) * g.H * g.W - base_pix) * g.Cin * 4;                     // bytes up to the end of the input
    const auto xr = uniform_rsrc(x + base_pix * g.Cin, left);
    unsigned voff[NA];                            // byte offset of the current tap's pixel of each staged row (0xFFFFFFFF: padding)
    auto load_a = [&]() {
        if (ci0 == 0) {                           // new tap (uniform branch, once per Cin / BK k-tiles)
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int hi = (hw0[j] >> 16) - 1 + kh, wi = (hw0[j] & 0xFFFF) - 1 + kw;
                const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                voff[j] = ok ? (unsigned)(((int64_t)pbase[j] + kh * g.W + kw - base_pix) * g.Cin + c4) * 4u : 0xFFFFFFFFu;
            }
        }
        const unsigned soff = (unsigned)ci0 * 4u;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            // (bit_cast of the whole vector: indexing the builtin's result through `auto` gave element 0 four times with hipcc 7.2)
            ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[j], soff, 0));
        }
        ci0 += BK;
        if (ci0 == g.Cin) { ci0 = 0; if (++kw == 3) { kw = 0; ++kh; } }
    };
    const int nk = D / BK;
    const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
    const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
    constexpr bool PINNED = CHUNK != 0 && TM * TN == 4 && ISX_PIN_KTILE;
    KtilePtrs<BK> pins;
    if constexpr (PINNED) pins = pin_ktile_ptrs<BK, LDA, LDB>(a_base, b_base);
    auto take_tot = [&]() {                       // the tile's value: the sum of the chunk sums
        if constexpr (CHUNK != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
        }
    };
    if (AHEAD2 && TM * TN == 1 && !(nk & 1)) {
        // 64x64 tiles: a k-tile is 16 MFMAs per wave -- 1024 matrix-pipe cycles, ~4000 when four waves share the SIMD -- while a loaded HBM / L2
        // round trip can take longer (scratch/lab/expand_lab.hip: 6200 -> 5800 cycles per k-tile, fused kernel 2.98 -> 2.95 ms): the operands are
        // requested TWO k-tiles ahead, in two staging register sets (+16 VGPRs; the LDS stays single-staged).  An even k-tile count only (Cin a
        // multiple of 64); one trip of the loop = two k-tiles = ONE chunk of the two-level sum.
        static_assert(!AHEAD2 || CHUNK % (2 * BK) == 0, "a chunk is a whole number of trips of the two-ahead loop");
        f32x16 (*totp2)[TN] = nullptr;
        if constexpr (CHUNK != 0) totp2 = tot;
        float4 ra1[NA], ra2[NA], rb2[BN * BK / 1024];
        auto stage = [&](float4 (&qa)[NA], float4 (&qb)[BN * BK / 1024], int kt) {
            load_a();
#pragma unroll
            for (int j = 0; j < NA; ++j) qa[j] = ra[j];
            load_tile<true, BN, BK>(Wt, N, D, n0, kt * BK, qb);
        };
        load_a();
        load_tile<true, BN, BK>(Wt, N, D, n0, 0, rb);
        store_tile<BM, BK>(As, ra);
        store_tile<BN, BK>(Bs, rb);
        __syncthreads();
        stage(ra1, rb, 1);
        // steady state, two k-tiles per trip: tile kt is in the LDS, tile kt + 1 on its way to (ra1, rb), tile kt + 2 is requested into (ra2, rb2).
        // The requests inside the loop are UNCONDITIONAL (hipcc's wait-count pass falls back to vmcnt(0) behind a conditional load, which would
        // undo the prefetch); those of the last trip point past the last tap / weight column -- range-checked buffer loads, values never used.
        for (int kt = 0; kt < nk; kt += 2) {
            stage(ra2, rb2, kt + 2);
            if (CHUNK != 0 && (kt * BK) % (CHUNK ? CHUNK : 1) == 0)      // chunk start: C = 0 (interleaved fold: the previous chunk's chain is added in front of it)
                mfma_ktile<TM, TN, BK, LDA, LDB, CHUNK != 0>(a_base, b_base, acc, ISX_FOLD_INTERLEAVE >= 2 ? totp2 : nullptr);
            else mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
            __syncthreads();
            store_tile<BM, BK>(As, ra1);
            store_tile<BN, BK>(Bs, rb);
            __syncthreads();
            stage(ra1, rb, kt + 3);
            mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
            __syncthreads();
            if (kt + 2 < nk) {
                store_tile<BM, BK>(As, ra2);
                store_tile<BN, BK>(Bs, rb2);
                __syncthreads();
            }
            if constexpr (ISX_FOLD_INTERLEAVE < 2) {
                if constexpr (CHUNK == 2 * BK) add_chunk<TM, TN>(tot, acc);
                else if constexpr (CHUNK != 0) { if (((kt + 2) * BK) % CHUNK == 0 || kt + 2 >= nk) add_chunk<TM, TN>(tot, acc); }
            }
        }
        if constexpr (ISX_FOLD_INTERLEAVE >= 2 && CHUNK != 0) add_chunk<TM, TN>(tot, acc);      // the last chunk
        take_tot();
        return;
    }
    load_a();
    load_tile<true, BN, BK>(Wt, N, D, n0, 0, rb);
    store_tile<BM, BK>(As, ra);
    store_tile<BN, BK>(Bs, rb);
    __syncthreads();

    // outer loop: chunks of the two-level sum (one pass when CHUNK == 0); inner loop: the staged k-tiles of a chunk (gemm_tile.hpp); the first
    // k-tile of a chunk starts its chains with C = 0
    f32x16 (*totp)[TN] = nullptr;
    if constexpr (CHUNK != 0) totp = tot;
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
    if constexpr (CHUNK == 0) {
        for (int kt = 0; kt < nk; ++kt) body(kt, std::false_type());
    } else {
        for (int kt = 0; kt < nk;) {
            const int kend = kt + CHUNK / BK < nk ? kt + CHUNK / BK : nk;
            body(kt++, std::true_type());                      // (interleaved fold: adds the PREVIOUS chunk's chain in front of its C = 0 MFMAs)
            for (; kt < kend; ++kt) body(kt, std::false_type());
            if (!((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2)) add_chunk<TM, TN>(tot, acc);
        }
        if ((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2) add_chunk<TM, TN>(tot, acc);       // the last chunk
    }
    take_tot();
}

}  // namespace isx
