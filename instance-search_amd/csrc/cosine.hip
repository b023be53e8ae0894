// cosine.hip -- query x gallery cosine similarity on the fp32 matrix cores.
//
// Replaces  sim = torch.mm(test_emb, ref_emb.t())  (test/classif_finetune_test.py:82,
// classif_regions_test.py:73, siamese_descriptor_test.py:77, siamese_regions_test.py:76,
// utils/train_siamese.py:53,70) and, fused with select.hip, the sort/topk/max that the
// reference runs on that matrix (utils/metrics.py:10-13,33).
//
// Numerics: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fma chain, so every
// score equals  acc = fmaf(q[k], g[k], acc), k = 0..D-1  exactly -- the oracle's
// definition.  K is never split across waves or blocks, so the order is preserved.
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs), BK = 16 (124 VGPRs, 16.5 KB LDS:
// 4 workgroups per CU; measured 133 TFLOP/s vs 130 at BK = 32 and 122 at BK = 8); smaller tiles
// (64x128, 128x64, 64x64, BK = 32) are picked for mid-size problems.  Q and G tiles
// are staged K-major in LDS ([k][row], row stride 129 floats): the staging loads are
// 16 B/lane with 8 lanes covering one 128-B row segment (coalesced), the transposed
// ds_write_b32 are bank-conflict-free by the odd stride, and the MFMA operand reads are
// 32 consecutive floats per half-wave (conflict-free ds_read_b32).  Global loads of
// tile t+1 are issued before the MFMAs of tile t (register prefetch).  Workgroups are
// remapped XCD-aware: each XCD walks a contiguous range of tiles, ordered so that
// concurrently resident tiles share Q / G panels in that XCD's L2.
//
// The same kernel, epilogue mode 2, is the 1x1 convolution of the inference trunk (entry point in conv.hip); the chunked
// running-top-k driver at the end of this file (run_topk_chunks) serves isx_cosine_topk and both phases of fast.hip.
#include <stdlib.h>

#include "gemm_tile.hpp"

namespace isx {

#ifndef ISX_A_NT
#define ISX_A_NT 0              // A/B (round 6): aux bits of the ACTIVATION operand loads of the convolution GEMM (2 = nt)
#endif
#ifndef ISX_ST_NT
#define ISX_ST_NT 0             // A/B (round 6): aux bits of the convolution GEMM's output stores (2 = nt)
#endif
#ifndef ISX_STAMPS
#define ISX_STAMPS 0            // lab builds only (tools/build_variant.sh stamps -DISX_STAMPS=1, tools/conv_phase_lab.py): wave 0 of every workgroup of the
#endif                          // convolution GEMM records the shader clock at its phase boundaries into the buffer set by isx_debug_set_stamps
#if ISX_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define ISX_STAMP(i) do { if (g_stamps && threadIdx.x == 0) g_stamps[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define ISX_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ISX_STAMP(i) do { } while (0)
#define ISX_STAMP_DRAIN() do { } while (0)
#endif

// Block tile (64*TM) x (64*TN): 4 waves as 2x2, each wave TM x TN MFMA tiles of 32x32.
// EPI: 0 = store scores, 1 = top-k filter (thr, gflag, ngrp), 2 = 1x1-convolution epilogue: thr = bias[n],
// gflag = residual (float, same layout as C) or null, ngrp = relu flag
// CHUNK (convolution mode only): terms per first-level chain of the two-level sum (gemm_tile.hpp), 0 = one chain over all of D (scores, gradients)
template <bool ALIGNED, int TM, int TN, int EPI, int BK, int CHUNK = (EPI == 2 ? kConvChunk : 0)>
__device__ __forceinline__ void cosine_gemm_tile(float* __restrict__ lds, const float* __restrict__ Q, int64_t M,
                                                 const float* __restrict__ G, int64_t N, int D,
                                                 float* __restrict__ C, int64_t ldc, int64_t m0, int64_t n0,
                                                 const float* __restrict__ thr, uint8_t* __restrict__ gflag,
                                                 int ngrp) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + lds_pad(BK), LDB = BN + lds_pad(BK);
    float* As = lds;
    float* Bs = lds + BK * LDA;

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
    f32x16 tot[CHUNK ? TM : 1][CHUNK ? TN : 1];
    zero_tiles(tot);

    // convolution epilogue on 64x64 tiles (short K loops, residual layers): fetch the residual values before
    // the main loop so that their latency overlaps the operand loads and the MFMAs
    constexpr bool PRE_RES = (EPI == 2 && TM * TN == 1);
    float pre_res[PRE_RES ? 16 : 1];
    // epilogue addressing of the convolution mode: wave-uniform row pointers (SGPRs) + one 32-bit lane offset
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    if (PRE_RES) {
        const float* res = reinterpret_cast<const float*>(gflag);
        if (res) {
            const auto rr = conv_tile_rsrc(res, m0, M, ldc, BM);
            const unsigned lo = conv_lane_off(n0 + wn_u * 32 + l31, N, wm_u * 32 + 4 * half, ldc);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                pre_res[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0));
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) pre_res[e] = 0.0f;
        }
    }

    // larger convolution tiles: the residual of the WHOLE tile is requested in one go right behind the main loop, into the registers the first-level
    // chains leave free, and consumed tile by tile as it arrives -- one round trip instead of TM x TN serialised ones (gemm_tile.hpp, epilogue_fetch).
    // (Requested one k-tile earlier, under the last MFMAs, the 64 values of a 128x128 tile push the kernel past 256 VGPRs: 204 B of scratch.)
    constexpr bool LATE_RES = (EPI == 2 && TM * TN == 4 && ISX_EPI_LOADS_FIRST && !ISX_EPI_LDS);       // (128x64 tiles at their 128-register bound: 24-112 B of scratch with it)
    float late_res[LATE_RES ? TM : 1][LATE_RES ? TN : 1][16];
    const float* res_ptr = (EPI == 2) ? reinterpret_cast<const float*>(gflag) : nullptr;

    float4 ra[BM * BK / 1024], rb[BN * BK / 1024];
    const int nk = (D + BK - 1) / BK;
    if (EPI == 2) ISX_STAMP(0);
    load_tile<ALIGNED, BM, BK, (EPI == 2 ? ISX_A_NT : 0)>(Q, M, D, m0, 0, ra);
    load_tile<ALIGNED, BN, BK>(G, N, D, n0, 0, rb);
    store_tile<BM, BK>(As, ra);
    store_tile<BN, BK>(Bs, rb);
    __syncthreads();
    if (EPI == 2) ISX_STAMP(1);

    const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
    const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
    constexpr bool PINNED = CHUNK != 0 && TM * TN == 4 && ISX_PIN_KTILE;
    KtilePtrs<BK> pins;
    if constexpr (PINNED) pins = pin_ktile_ptrs<BK, LDA, LDB>(a_base, b_base);

    // outer loop: chunks of the two-level sum (one pass when CHUNK == 0); inner loop: the staged k-tiles of a chunk.  The FIRST k-tile of a chunk is
    // a second copy of the body whose first MFMAs take C = 0 (no zeroing pass), the chain is added to tot behind the chunk's last barrier.
    f32x16 (*totp)[TN] = nullptr;
    if constexpr (CHUNK != 0) totp = tot;
    auto body = [&](int kt, auto zero_c) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_tile<ALIGNED, BM, BK, (EPI == 2 ? ISX_A_NT : 0)>(Q, M, D, m0, (kt + 1) * BK, ra);
            load_tile<ALIGNED, BN, BK>(G, N, D, n0, (kt + 1) * BK, rb);
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
    if (EPI == 2) ISX_STAMP(2);
    // the TN bias values of this lane's columns BEFORE everything else of the epilogue: a bias load behind the residual requests would make the first
    // add wait for all of them, and one between two tiles' stores would wait for those stores (vmcnt counts both on gfx9)
    float bias_pre[TN];
    if constexpr (EPI == 2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = (int)(n0 + wn_u * (32 * TN) + j * 32) + l31;
            bias_pre[j] = ncol < N ? thr[ncol] : 0.0f;
        }
    }
    if constexpr (LATE_RES) {
        if (res_ptr) {
            epilogue_fetch<TM, TN>(late_res, res_ptr, m0, M, n0, N, ldc, BM, wm_u * (32 * TM), wn_u * (32 * TN), l31, half);
            __builtin_amdgcn_sched_barrier(0);               // every load above the first store
            if (ISX_STAMPS) { ISX_STAMP_DRAIN(); ISX_STAMP(3); }
        }
    }
    if constexpr (CHUNK != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];       // the epilogues below read acc
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    if (EPI == 3) {
        // backward.hip: C = (acc (+ add)) . [mask > 0];  thr = mask (M x N, layout of C) or null, gflag = add (float, layout of C) or null
        conv_epilogue_buffers<TM, TN, 1>(acc, C, reinterpret_cast<const float*>(gflag), nullptr, 0, m0, M, n0, N, ldc, BM, wm_u * (32 * TM), wn_u * (32 * TN), l31,
                                      half, thr);
        return;
    }
    if (EPI == 2 && TM * TN == 4 && ISX_EPI_LDS) {
        // 128x128 convolution tiles: float4 epilogue through a wave-private LDS transpose (gemm_tile.hpp) when the shapes allow 16-B accesses
        const float* res = reinterpret_cast<const float*>(gflag);
        if ((N & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)C | (uintptr_t)(res ? res : C) | (uintptr_t)thr) & 15) == 0) {        // uniform
            conv_epilogue_lds<TM, TN>(acc, lds + (threadIdx.x >> 6) * (32 * TM) * (32 * TN + 4), C, res, thr, ngrp, m0, M, n0, N, ldc, BM, wm_u * (32 * TM),
                                      wn_u * (32 * TN), lane);
            if (ISX_STAMPS) { ISX_STAMP(3); ISX_STAMP(4); ISX_STAMP_DRAIN(); ISX_STAMP(5); }      // (the LDS epilogue's residual wait is inside it: stamp 3 = 4)
            return;
        }
    }
    if (EPI == 2) {
        // Convolution epilogue through BUFFER instructions: a wave-uniform descriptor of the tile's rows (clipped at row M by the
        // hardware), one 32-bit lane offset per 32x32 MFMA tile (a column >= N gets an offset outside the descriptor: its loads return
        // 0, its stores are dropped) and the row offset (e & 3) + 8 (e >> 2) as an SGPR: ~4 instructions per output element and no
        // branch, where per-element 64-bit addresses and edge tests cost ~20 (a fifth of a K = 256 tile's time).
        const float* res = reinterpret_cast<const float*>(gflag);
        const auto rc = conv_tile_rsrc(C, m0, M, ldc, BM);
        const auto rr = conv_tile_rsrc(res ? res : C, m0, M, ldc, BM);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol = (int)(n0 + wn_u * (32 * TN) + j * 32) + l31;
                const float bias_v = bias_pre[j];
                const unsigned lo = conv_lane_off(ncol, N, wm_u * (32 * TM) + i * 32 + 4 * half, ldc);
                float rv[16];
                if (!PRE_RES && !LATE_RES && res) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0));
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float y = acc[i][j][e] + bias_v;
                    if (PRE_RES) { if (res) y += pre_res[e]; }
                    else if (LATE_RES) { if (res) y += late_res[LATE_RES ? i : 0][LATE_RES ? j : 0][e]; }
                    else if (res) y += rv[e];
                    if (ngrp) y = fmaxf(y, 0.0f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rc, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), ISX_ST_NT);
                }
            }
        }
        if (ISX_STAMPS) { ISX_STAMP(4); ISX_STAMP_DRAIN(); ISX_STAMP(5); }
        return;
    }
    if (EPI == 0 && (int64_t)BM * ldc * 4 < (1ll << 32)) {
        // plain score store through BUFFER instructions (round 4; the convolution epilogue has used them since round 2): a wave-uniform
        // descriptor of the tile's rows clipped at row M, one 32-bit lane offset per MFMA tile (a column >= N is sent outside the
        // descriptor and its store dropped), the row offset as an SGPR -- no per-element 64-bit address, no edge branch.  Short-K
        // problems (D = 464: 29 k-tiles per 128 x 128 tile) spent a tenth of their time in the old epilogue.
        const auto rc = conv_tile_rsrc(C, m0, M, ldc, BM);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol = (int)(n0 + wn_u * (32 * TN) + j * 32) + l31;
                const unsigned lo = conv_lane_off(ncol, N, wm_u * (32 * TM) + i * 32 + 4 * half, ldc);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][j][e];             // (a scalar copy first: bit_cast applied to the vector element itself stored element 0 sixteen times, hipcc 7.2)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // wave-uniform row pointers + 32-bit lane offsets (64-bit per-element addresses cost ~35 VGPRs and a wave per SIMD)
            const int64_t ng = n0 + wn_u * (32 * TN) + j * 32;      // first column of this 32-column group (uniform)
            const int ncol = (int)ng + l31;
            const bool n_ok = ncol < N;
            const int lane_off = 4 * half * (int)ldc + ncol;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t mu = m0 + wm_u * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2);     // uniform; this lane's row = mu + 4 * half
                const bool row_ok = (mu + 4 * half < M);
                const float v = acc[i][j][e];
                if (EPI == 1) {
                    // fused top-k filter: a 32-column group of row m is stored only if one of its scores
                    // can still enter the row's top-k (score >= thr[m], a lower bound of the final k-th
                    // score); one flag byte per (row, group) tells the select kernel which groups exist.
                    // The test is a compare + wave ballot (no cross-lane data movement).
                    const float t = row_ok ? (thr + mu)[4 * half] : INFINITY;
                    const unsigned long long qm = __ballot(n_ok && v >= t);
                    const bool q = ((half ? (qm >> 32) : qm) & 0xFFFFFFFFull) != 0ull;
                    if (row_ok && ng < N) {
                        if (l31 == 0) (gflag + mu * ngrp + (ng >> 5))[4 * half * ngrp] = q ? 1 : 0;
                        if (q && n_ok) (C + mu * ldc)[lane_off] = v;
                    }
                } else {
                    if (row_ok && n_ok) (C + mu * ldc)[lane_off] = v;
                }
            }
        }
    }
}

// (convolution mode: two accumulator sets.  128x128 tiles: two workgroups per CU -- without the bound hipcc takes 296 registers and one fits;
// the smaller tiles serve the HBM-bound layers and keep four -- unbounded, the 128x64 shape took 164 registers and lost a fifth on 256 -> 64 at 56x56)
template <bool ALIGNED, int TM, int TN, int EPI, int BK>
__global__ __launch_bounds__(256, EPI != 2 ? 1 : TM * TN == 4 ? ISX_WG_PER_CU_128 : 4) void cosine_gemm_kernel(const float* __restrict__ Q, int64_t M,
                                                          const float* __restrict__ G, int64_t N, int D,
                                                          float* __restrict__ C, int64_t ldc, TileMap tm,
                                                          const float* __restrict__ thr, uint8_t* __restrict__ gflag,
                                                          int ngrp) {
    constexpr int kStage = BK * (64 * TM + 64 * TN + 2 * lds_pad(BK)), kEpi = (EPI == 2 && TM * TN == 4 && ISX_EPI_LDS) ? epilogue_lds_floats<TM, TN>() : 0;
    __shared__ float lds[kStage > kEpi ? kStage : kEpi];
    int tile_m, tile_n;
    tile_of_block(tm, tile_m, tile_n);
    const int64_t m0 = (int64_t)tile_m * (64 * TM), n0 = (int64_t)tile_n * (64 * TN);
    if (tm.m_active && m0 >= *tm.m_active) return;          // uniform: whole tile beyond the live rows
    cosine_gemm_tile<ALIGNED, TM, TN, EPI, BK>(lds, Q, M, G, N, D, C, ldc, m0, n0, thr, gflag, ngrp);
}

// 1x1-convolution GEMM (EPI = 2) as 128x128 tiles with a 64x64 TAIL: the rows past the last whole round of 1024 resident workgroups run
// as 64x64 tiles in the same grid (see conv3x3_tail_kernel in conv.hip: a few 128x128 tiles alone on their CUs at the end of a launch of
// three to twelve rounds cost 3-10 % of it).  Same arithmetic per output element.
template <bool ALIGNED>
__global__ __launch_bounds__(256, ISX_WG_PER_CU_128) void conv1x1_tail_kernel(const float* __restrict__ Q, int64_t M, const float* __restrict__ G, int64_t N, int D,
                                                              float* __restrict__ C, int64_t ldc, TileMap tm_big, TileMap tm_small, int64_t m_split,
                                                              const float* __restrict__ bias, uint8_t* __restrict__ res, int relu) {
    constexpr int kBig0 = 16 * (128 + 128 + 2 * lds_pad(16)), kEpi = ISX_EPI_LDS ? epilogue_lds_floats<2, 2>() : 0, kBig = kBig0 > kEpi ? kBig0 : kEpi;
    __shared__ float lds[kBig > kTailLdsFloats ? kBig : kTailLdsFloats];
    const int nbig = tm_big.tiles_m * tm_big.tiles_n;                 // a multiple of 8: a block's XCD is the same in both numberings
    int tile_m, tile_n;
    if ((int)blockIdx.x < nbig) {
        tile_of_block(tm_big, tile_m, tile_n, (int)blockIdx.x, nbig);
        cosine_gemm_tile<ALIGNED, 2, 2, 2, 16>(lds, Q, M, G, N, D, C, ldc, (int64_t)tile_m * 128, (int64_t)tile_n * 128, bias, res, relu);
    } else {
        tile_of_block(tm_small, tile_m, tile_n, (int)blockIdx.x - nbig, tm_small.tiles_m * tm_small.tiles_n);
        cosine_gemm_tile<ALIGNED, ISX_TAIL_TM, 1, 2, 32>(lds, Q, M, G, N, D, C, ldc, m_split + (int64_t)tile_m * (64 * ISX_TAIL_TM), (int64_t)tile_n * 64, bias, res, relu);
    }
}

// ---- PERSISTENT 128x128 tiles for the 1x1 convolutions (round 5) ----------------------------------------------------------------------------------
// A short-K layer (K = 128 ... 512: 8 ... 32 k-tiles) pays the fill of its load pipeline once per TILE: the first k-tile's operands take an
// HBM / L2 round trip (~2 us) before the first MFMA can issue, a quarter of a K = 128 tile's matrix time, and two workgroups per CU cannot
// hide it for each other (128 -> 512 + residual at 28x28: 0.62 of its own roofline, neither HBM- nor MFMA-bound).  Here 512 workgroups
// (two per CU) each walk tiles b, b + G, b + 2 G, ... of the same XCD-aware order, and the operands of the NEXT tile's first k-tile are
// requested before the current tile's epilogue: they land while the epilogue's loads and stores are in flight.  The staging registers are
// free at that point, so the prefetch costs none.  Same arithmetic per output element as cosine_gemm_tile<..., EPI 2>.
// MEASURED: bit-identical and 4 % slower than one workgroup per tile (see launch_gemm_any): kept as an A/B (ISX_CONV_PERSIST=1), not dispatched.
template <bool ALIGNED>
__global__ __launch_bounds__(256, ISX_WG_PER_CU_128) void conv1x1_persist_kernel(const float* __restrict__ Q, int64_t M, const float* __restrict__ G, int64_t N, int D,
                                                                                float* __restrict__ C, int64_t ldc, TileMap tm, int ntiles, TileMap tm_small,
                                                                                int64_t m_split, const float* __restrict__ bias, const float* __restrict__ res,
                                                                                int relu, int* __restrict__ tickets) {
    // tickets != nullptr (ISX_CONV_PERSIST=2, round 6): the next tile is DRAWN, not strided -- one counter per XCD (8 ints, zeroed by the launcher), a
    // workgroup on XCD x = blockIdx & 7 takes the next undone tile of x's own contiguous range, so that the dispatcher's dynamic balance is kept and the
    // tiles of an XCD still share their operands in its L2.  The ticket is drawn at the top of a tile and has landed by its epilogue.
    constexpr int TM = 2, TN = 2, BK = 16, BM = 128, BN = 128, LDA = BM + lds_pad(BK), LDB = BN + lds_pad(BK), CHUNK = kConvChunk;
    __shared__ float lds[BK * (LDA + LDB)];
    __shared__ int s_next;
    float* As = lds;
    float* Bs = lds + BK * LDA;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
    const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
    constexpr bool PINNED = CHUNK != 0 && ISX_PIN_KTILE;
    KtilePtrs<BK> pins;
    if constexpr (PINNED) pins = pin_ktile_ptrs<BK, LDA, LDB>(a_base, b_base);
    const int nk = (D + BK - 1) / BK;

    float4 ra[BM * BK / 1024], rb[BN * BK / 1024];
    int t = (int)blockIdx.x;
    int tile_m = 0, tile_n = 0;
    if (t < ntiles) {
        tile_of_block(tm, tile_m, tile_n, t, ntiles);
        load_tile<ALIGNED, BM, BK>(Q, M, D, (int64_t)tile_m * BM, 0, ra);
        load_tile<ALIGNED, BN, BK>(G, N, D, (int64_t)tile_n * BN, 0, rb);
    }
    while (t < ntiles) {
        const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
        f32x16 acc[TM][TN], tot[CHUNK ? TM : 1][CHUNK ? TN : 1];
        zero_tiles(acc);
        zero_tiles(tot);
        int drawn = 0;                                                       // thread 0: requested now, consumed in the last k-tile (the atomic's round trip hides behind the main loop)
        if (tickets && threadIdx.x == 0) drawn = atomicAdd(tickets + ((int)blockIdx.x & 7), 1);
        store_tile<BM, BK>(As, ra);
        store_tile<BN, BK>(Bs, rb);
        __syncthreads();
        f32x16 (*totp)[TN] = nullptr;
        if constexpr (CHUNK != 0) totp = tot;
        auto body = [&](int kt, auto zero_c) {
            const bool more = (kt + 1 < nk);
            if (more) {
                load_tile<ALIGNED, BM, BK>(Q, M, D, m0, (kt + 1) * BK, ra);
                load_tile<ALIGNED, BN, BK>(G, N, D, n0, (kt + 1) * BK, rb);
            }
            if (!more && tickets && threadIdx.x == 0) s_next = ((int)blockIdx.x & 7) + 8 * ((int)(gridDim.x >> 3) + drawn);       // published by the barrier below
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
                body(kt++, std::true_type());
                for (; kt < kend; ++kt) body(kt, std::false_type());
                if (!((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2)) add_chunk<TM, TN>(tot, acc);
            }
            if ((PINNED && ISX_FOLD_INTERLEAVE) || ISX_FOLD_INTERLEAVE >= 2) add_chunk<TM, TN>(tot, acc);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
        }
        // the next tile's first operands: on their way while this tile's epilogue runs (every wave has left the LDS: the loop ended on a barrier)
        int tn = t + (int)gridDim.x;
        if (tickets) tn = __builtin_amdgcn_readfirstlane(s_next);
        if (tn < ntiles) {
            tile_of_block(tm, tile_m, tile_n, tn, ntiles);
            load_tile<ALIGNED, BM, BK>(Q, M, D, (int64_t)tile_m * BM, 0, ra);
            load_tile<ALIGNED, BN, BK>(G, N, D, (int64_t)tile_n * BN, 0, rb);
        }
        conv_epilogue_buffers<TM, TN>(acc, C, res, bias, relu, m0, M, n0, N, ldc, BM, wm_u * (32 * TM), wn_u * (32 * TN), l31, half);
        t = tn;
    }
    // rows past the last whole round of 128x128 tiles: 64x64 tiles (conv1x1_tail_kernel's scheme), spread over the same workgroups
    const int nsmall = tm_small.tiles_m * tm_small.tiles_n;
    for (int ts = (int)blockIdx.x; ts < nsmall; ts += (int)gridDim.x) {
        tile_of_block(tm_small, tile_m, tile_n, ts, nsmall);
        cosine_gemm_tile<ALIGNED, 1, 1, 2, 32>(lds, Q, M, G, N, D, C, ldc, m_split + (int64_t)tile_m * 64, (int64_t)tile_n * 64, bias, (uint8_t*)res, relu);
    }
}

// rows covered by whole rounds of 128x128 tiles when the rest of the grid is a partial round (0: no split)
std::atomic<int> g_tail_split{1};
int64_t gemm_tail_split_rows(int64_t M, int64_t N, int64_t slots) {
    static const bool env_off = [] { const char* e = getenv("ISX_TAIL_SPLIT"); return e && e[0] == '0'; }();      // A/B from the environment
    if (env_off) return 0;
    const int64_t tn = (N + 127) / 128, tiles = ((M + 127) / 128) * tn;
    if (!g_tail_split || tn > slots || slots % tn != 0) return 0;
    const int64_t rounds = tiles / slots, rem = tiles - rounds * slots;
    if (rounds < 1 || rem == 0 || rem > slots * 4 / 5) return 0;
    return rounds * (slots / tn) * 128;
}

// ---- tile-shape selection ------------------------------------------------------------------
// Candidate block tiles with their measured steady-state efficiency (fraction of the fp32-MFMA
// peak on a large problem) and resident workgroups per CU.  The launcher picks the shape with the
// smallest estimated time  rounds(T tiles over S slots) * tile_work / efficiency  -- large
// problems get 128x128, mid-size ones (bench: 512 x 10k) avoid a half-empty last round.
struct TileCfg { int tm, tn, wg_per_cu; float eff; };
static const TileCfg kCfgs[] = { {2, 2, 4, 0.92f}, {1, 2, 4, 0.87f}, {2, 1, 4, 0.89f}, {1, 1, 6, 0.84f} };     // 10 000 x 32 768 x 2048: 145 / 137 / 140 / 132 TFLOP/s
// The tile shape (index into kCfgs: 0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64) with the smallest estimated time among those in
// `mask`; eff[c]: steady-state efficiency of shape c for the calling kernel family; split > 0: shape 0 runs with a 64x64 tail.
int pick_tile_cfg(int64_t M, int64_t N, int64_t split, const float* eff, unsigned mask, int wg_per_cu_128) {
    int best = -1;
    double best_t = 1e300;
    for (int c = 0; c < 4; ++c) {
        if (!((mask >> c) & 1u)) continue;
        TileCfg k = kCfgs[c];
        if (c == 0) k.wg_per_cu = wg_per_cu_128;
        const double tiles = (double)((M + 64 * k.tm - 1) / (64 * k.tm)) * (double)((N + 64 * k.tn - 1) / (64 * k.tn));
        const double slots = 256.0 * k.wg_per_cu;
        // time in units of "one full round" (= wg_per_cu tiles on every CU).  Workgroups finish unevenly,
        // so a launch costs its tile count plus a tail that is ~0.2 round for launches below one round and
        // fades quadratically for longer ones (fitted on MI355X, 256 ... 10k query rows x 10k ... 100k)
        const double x = tiles / slots;
        const double t0 = (c == 0 ? 0.22 : c == 3 ? 0.15 : 0.25);
        double rounds = x + (x <= 0.7 ? t0 : t0 * (0.7 / x) * (0.7 / x));
        // a CU works its tiles off at the rate of its matrix pipe however many of them are resident: the launch cannot end before the
        // CU with one tile more than the average is done (50 176 x 512 x 2048: 1568 tiles of 128x128 = 6.1 per CU took 7 tile times)
        const double per_cu = (double)((int64_t)((tiles + 255.0) / 256.0)) / k.wg_per_cu;
        rounds = rounds > per_cu ? rounds : per_cu;
        if (c == 0 && split > 0) rounds = x + 0.05;                          // the tail runs as small tiles: no round quantisation
        const double t = rounds * k.wg_per_cu * (k.tm * k.tn) / eff[c];
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

static std::atomic<int> g_force_cfg{[] { const char* e = getenv("ISX_DEBUG_GEMM_CFG"); return e ? atoi(e) : -1; }()};            // debug / A-B hook
void set_gemm_cfg(int c) { g_force_cfg = c; }

template <int TM, int TN, int BK>
static void launch_cfg(bool aligned, const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                       const float* thr, uint8_t* gmax, hipStream_t st, const int* m_active, int epi, int relu) {
    TileMap tm;
    tm.m_active = m_active;
    tm.tiles_m = (int)((M + 64 * TM - 1) / (64 * TM));
    tm.tiles_n = (int)((N + 64 * TN - 1) / (64 * TN));
    const int ngrp = (int)((N + 31) / 32);
    const dim3 grid((unsigned)(tm.tiles_m * tm.tiles_n)), block(256);
    if (epi == 2) {
        if (aligned) hipLaunchKernelGGL((cosine_gemm_kernel<true, TM, TN, 2, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, relu);
        else hipLaunchKernelGGL((cosine_gemm_kernel<false, TM, TN, 2, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, relu);
    } else if (epi == 3) {
        if (aligned) hipLaunchKernelGGL((cosine_gemm_kernel<true, TM, TN, 3, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, 0);
        else hipLaunchKernelGGL((cosine_gemm_kernel<false, TM, TN, 3, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, 0);
    } else if (gmax) {
        if (aligned) hipLaunchKernelGGL((cosine_gemm_kernel<true, TM, TN, true, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, ngrp);
        else hipLaunchKernelGGL((cosine_gemm_kernel<false, TM, TN, true, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, ngrp);
    } else {
        if (aligned) hipLaunchKernelGGL((cosine_gemm_kernel<true, TM, TN, false, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, ngrp);
        else hipLaunchKernelGGL((cosine_gemm_kernel<false, TM, TN, false, BK>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tm, thr, gmax, ngrp);
    }
}

static int launch_gemm_any(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                           const float* thr, uint8_t* gmax, hipStream_t st, const int* m_active = nullptr, int epi = -1, int relu = 0) {
    if (epi < 0) epi = gmax ? 1 : 0;
    if (M == 0 || N == 0) return ISX_OK;
    if (((M + 63) / 64) * ((N + 63) / 64) >= (1ll << 31)) { isx_set_error("cosine gemm: too many tiles for one grid"); return ISX_ERR_ARG; }
    const bool aligned = (D % 32 == 0) && (((uintptr_t)Q | (uintptr_t)G) % 16 == 0);     // no k tail for BK = 16 or 32
    const bool aligned16 = aligned || ((D % 16 == 0) && (((uintptr_t)Q | (uintptr_t)G) % 16 == 0));   // enough for the BK = 16 (128x128) tiles: D = 464
    // convolutions: 128x128 tiles (two workgroups per CU: the two-level sum) + 64x64 tail in one grid
    const int64_t split = (epi == 2) ? gemm_tail_split_rows(M, N, 256 * ISX_WG_PER_CU_128) : 0;
    static const float eff_gemm[4] = {kCfgs[0].eff, kCfgs[1].eff, kCfgs[2].eff, kCfgs[3].eff};
    int best = pick_tile_cfg(M, N, split, eff_gemm, 0xF, epi == 2 ? ISX_WG_PER_CU_128 : 4);
    // (Round 1 forced 64x64 tiles on residual layers and 128x64 on the others: the per-element epilogue was a visible share of a tile.
    // With the buffer-instruction epilogue the same round / tail model as for the score GEMM picks the convolution tiles: 128x128
    // wherever the grid fills the chip -- 256->1024 + residual 0.90 -> 0.87 ms, 512->2048 + residual 0.85 -> 0.81, 512->256 1.64 -> 1.58 --
    // and 128x64 for Cout = 64.)
    if (g_force_cfg >= 0 && g_force_cfg < 4) best = g_force_cfg;
    if (epi == 0 && best == 0 && g_force_cfg < 0 && g_tail_split && !m_active) {
        // Score matrix of FEW query rows against a long gallery (configs[1] / [2] retrieval: 1 000 x 100 000): 8 x 782 tiles of 128x128 are
        // 6.1 rounds of the 1024 resident workgroups and cost 7 -- the last 112 tiles run alone.  The gallery columns covered by whole rounds
        // go out as 128x128 tiles, the remaining columns as a second launch of 64x64 tiles (a quarter of the work each, 1536 resident).
        // Every score is the same k-ordered chain in either tile shape.
        const int64_t tm_ = (M + 127) / 128, tn_ = (N + 127) / 128, slots = 1024;
        const int64_t rounds = tm_ * tn_ / slots, rem = tm_ * tn_ - rounds * slots;
        const int64_t n_big = rounds * slots / tm_ * 128;                  // columns of the whole rounds
        if (rounds >= 1 && rem > 0 && rem <= slots * 3 / 5 && n_big > 0 && n_big < N && (n_big * D * 4) % 16 == 0) {
            launch_cfg<2, 2, 16>(aligned16, Q, M, G, n_big, D, C, ldc, thr, gmax, st, m_active, epi, relu);
            launch_cfg<1, 1, 32>(aligned, Q, M, G + n_big * D, N - n_big, D, C + n_big, ldc, thr, gmax, st, m_active, epi, relu);
            ISX_CHECK_LAUNCH("cosine_gemm");
            return ISX_OK;
        }
    }
    // A/B (round 5), OFF by default: persistent workgroups with the next tile's first operands prefetched across the epilogue are
    // 4 % SLOWER on the ten 1x1 shapes of the lab (11.37 vs 10.90 ms; 128 -> 512 + residual 1.10 vs 1.07): the hardware dispatcher's dynamic
    // placement of one workgroup per tile beats the static walk, and the pipeline fill of a tile is not what the short-K layers wait for.
    static const int persist_mode = [] { const char* e = getenv("ISX_CONV_PERSIST"); return e ? atoi(e) : 0; }();      // 1 = strided tiles (round 5), 2 = drawn tiles (round 6)
    static const bool use_persist = persist_mode == 1 || persist_mode == 2;
    if (epi == 2 && best == 0 && use_persist && (g_force_cfg < 0 || g_force_cfg == 0)) {
        TileMap tmap, small;
        tmap.m_active = small.m_active = nullptr;
        tmap.tiles_m = (int)((split > 0 ? split : M + 127) / 128); tmap.tiles_n = (int)((N + 127) / 128);
        small.tiles_m = split > 0 ? (int)((M - split + 63) / 64) : 0; small.tiles_n = (int)((N + 63) / 64);
        const int ntiles = tmap.tiles_m * tmap.tiles_n;
        const int slots = 256 * ISX_WG_PER_CU_128;
        const int want = ntiles + small.tiles_m * small.tiles_n;
        const dim3 grid((unsigned)(want < slots ? want : slots)), block(256);
        int* tickets = nullptr;
        if (persist_mode == 2 && grid.x % 8 == 0) {                           // A/B only: the counters live in a lazily allocated device buffer, zeroed per launch
            static int* g_tickets = nullptr;
            if (!g_tickets && hipMalloc((void**)&g_tickets, 64) != hipSuccess) g_tickets = nullptr;
            if (g_tickets && hipMemsetAsync(g_tickets, 0, 64, st) == hipSuccess) tickets = g_tickets;
        }
        if (aligned) hipLaunchKernelGGL((conv1x1_persist_kernel<true>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tmap, ntiles, small, split, thr, (const float*)gmax, relu, tickets);
        else hipLaunchKernelGGL((conv1x1_persist_kernel<false>), grid, block, 0, st, Q, M, G, N, D, C, ldc, tmap, ntiles, small, split, thr, (const float*)gmax, relu, tickets);
        ISX_CHECK_LAUNCH("conv1x1_persist");
        return ISX_OK;
    }
    if (best == 0 && split > 0) {
        TileMap big, small;
        big.m_active = small.m_active = nullptr;
        big.tiles_m = (int)(split / 128); big.tiles_n = (int)((N + 127) / 128);
        small.tiles_m = (int)((M - split + 64 * ISX_TAIL_TM - 1) / (64 * ISX_TAIL_TM)); small.tiles_n = (int)((N + 63) / 64);
        const dim3 grid((unsigned)(big.tiles_m * big.tiles_n + small.tiles_m * small.tiles_n)), block(256);
        if (aligned) hipLaunchKernelGGL((conv1x1_tail_kernel<true>), grid, block, 0, st, Q, M, G, N, D, C, ldc, big, small, split, thr, gmax, relu);
        else hipLaunchKernelGGL((conv1x1_tail_kernel<false>), grid, block, 0, st, Q, M, G, N, D, C, ldc, big, small, split, thr, gmax, relu);
        ISX_CHECK_LAUNCH("conv1x1_tail");
        return ISX_OK;
    }
    switch (best) {
        case 0: launch_cfg<2, 2, 16>(aligned16, Q, M, G, N, D, C, ldc, thr, gmax, st, m_active, epi, relu); break;
        case 1: launch_cfg<1, 2, 32>(aligned, Q, M, G, N, D, C, ldc, thr, gmax, st, m_active, epi, relu); break;
        case 2: launch_cfg<2, 1, 32>(aligned, Q, M, G, N, D, C, ldc, thr, gmax, st, m_active, epi, relu); break;
        default: launch_cfg<1, 1, 32>(aligned, Q, M, G, N, D, C, ldc, thr, gmax, st, m_active, epi, relu); break;
    }
    ISX_CHECK_LAUNCH("cosine_gemm");
    return ISX_OK;
}


// gradient of a 1x1 convolution wrt its input (backward.hip): C = (A . Bt^T (+ add)) . [mask > 0], epilogue mode 3
int launch_gemm_masked(const float* A, int64_t M, const float* Bt, int64_t N, int D, float* C, const float* mask, const float* add, hipStream_t st) {
    return launch_gemm_any(A, M, Bt, N, D, C, N, mask, (uint8_t*)add, st, nullptr, 3, 0);
}

// 1x1 convolution = the same GEMM with the bias / residual / ReLU epilogue (conv.hip)
int launch_conv1x1_gemm(const float* x, int64_t M, const float* w, int64_t N, int D, float* y, const float* bias, const float* residual, int relu,
                        hipStream_t st) {
    return launch_gemm_any(x, M, w, N, D, y, N, bias, (uint8_t*)residual, st, nullptr, 2, relu);
}

int launch_cosine_gemm(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc, hipStream_t st,
                       const int* m_active) {
    return launch_gemm_any(Q, M, G, N, D, C, ldc, nullptr, nullptr, st, m_active);
}

int launch_cosine_gemm_filter(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                              const float* thr, uint8_t* gmax, hipStream_t st, const int* m_active) {
    return launch_gemm_any(Q, M, G, N, D, C, ldc, thr, gmax, st, m_active);
}

// Workspace layout of isx_cosine_topk:
//   [ carry keys: M*k u64 | thr: M f32 | group flags: M*ceil(Nc/32) u8 | score chunk: M*Nc f32 ]
// The first column chunk (<= kFirstChunk columns) is materialised and selected in full; it leaves a
// per-row lower bound thr of the final k-th score.  Every later chunk runs the FILTERING GEMM: only
// 32-column groups whose best score reaches thr are stored and read back, so for typical data the
// M x N matrix is never written -- just one float per 32 scores.  Exact for any data: in the worst
// case (every group qualifies) the chunk is simply materialised in full, as in round 0.
static const int64_t kFirstChunk = [] { const char* e = getenv("ISX_TOPK_FIRST"); const long long v = e ? atoll(e) : 0; return (int64_t)(v >= 256 ? v / 256 * 256 : 8192); }();      // A/B knob; default 8192
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- chunked running top-k driver (shared by isx_cosine_topk and the fast path of fast.hip) ------
size_t topk_fixed_bytes(int64_t M, int k) { return align256((size_t)M * k * 8) + align256((size_t)M * 4); }
size_t topk_chunk_bytes(int64_t M, int64_t nc) { return align256((size_t)M * ((nc + 31) / 32)) + (size_t)M * nc * 4; }

int64_t topk_recommended_chunk(int64_t M, int64_t N) {
    // whole matrix if it is <= 1 GiB, else column chunks of ~1 GiB (multiple of 2048 columns)
    static const size_t budget = [] { const char* e = getenv("ISX_TOPK_CHUNK_MB"); const long long v = e ? atoll(e) : 0; return (size_t)(v >= 64 ? v : 1024) << 20; }();   // A/B knob; default 1 GiB
    int64_t nc = N;
    if ((size_t)M * N * 4 > budget) {
        nc = (int64_t)(budget / ((size_t)M * 4));
        nc = nc / 2048 * 2048;
        if (nc < 2048) nc = 2048;
        if (nc > N) nc = N;
    }
    return nc;
}

int run_topk_chunks(const TopkJob& j) {
    const int64_t M = j.M, N = j.N;
    const int k = j.k, D = j.D;
    hipStream_t st = j.st;
    const size_t fixed_b = topk_fixed_bytes(M, k);
    const int64_t min_nc = N < 128 ? N : 128;
    if (!j.ws || ((uintptr_t)j.ws % 256) != 0 || j.ws_bytes < fixed_b + topk_chunk_bytes(M, min_nc)) {
        isx_set_error("%s: workspace of %zu bytes too small or misaligned (need >= %zu, 256-B aligned)", j.who, j.ws_bytes,
                      fixed_b + topk_chunk_bytes(M, min_nc));
        return ISX_ERR_WORKSPACE;
    }
    // largest chunk width (multiple of 128 unless it covers N) whose flags + scores fit
    int64_t nc = (int64_t)((j.ws_bytes - fixed_b) / ((size_t)M * 4));
    if (nc > N) nc = N;
    while (nc > min_nc && topk_chunk_bytes(M, nc) > j.ws_bytes - fixed_b) nc -= (nc > 4096 ? 1024 : 128);
    if (nc < N) nc = nc >= 128 ? nc / 128 * 128 : nc;
    if (nc < N && nc >= 4096) {
        // a chunk launch runs ceil(tiles / slots) lock-step rounds of tiles (fp32: 128x128, 512 resident at the
        // filter kernel's 2 workgroups per CU ... measured best with 512; fp16: 256x256, one per CU): trim the width
        // (by at most 16 tiles) so that the last round is >= 90 % full
        const int64_t tile = j.Qh ? 256 : 128, slots = j.Qh ? 256 : 512;
        const int64_t tm_ = (M + tile - 1) / tile;
        for (int64_t tn = nc / tile, tries = 0; tries < 16 && tn > 16; --tn, ++tries) {
            const int64_t rem = (tm_ * tn) % slots;
            if (rem == 0 || rem >= slots * 9 / 10) { nc = tn * tile; break; }
        }
    }
    uint64_t* carry = (uint64_t*)j.ws;
    float* thr = (float*)((char*)j.ws + align256((size_t)M * k * 8));
    uint8_t* gflag = (uint8_t*)((char*)j.ws + fixed_b);
    float* chunk = (float*)((char*)gflag + align256((size_t)M * ((nc + 31) / 32)));
    const bool filter = (k <= kGroupSelectMaxK);
    if (!filter && (j.Qh || !j.emit || j.m_active)) { isx_set_error("%s: k=%d unsupported on this path", j.who, k); return ISX_ERR_ARG; }
    auto gemm = [&](int64_t c0, int64_t w, const float* t, uint8_t* gf) -> int {
        if (j.Qh) return launch_gemm_f16(j.Qh, M, j.Gh + c0 * D, w, D, chunk, w, t, gf, st);
        if (gf) return launch_cosine_gemm_filter(j.Q, M, j.G + c0 * D, w, D, chunk, w, t, gf, st, j.m_active);
        return launch_cosine_gemm(j.Q, M, j.G + c0 * D, w, D, chunk, w, st, j.m_active);
    };
    int64_t c0 = 0;
    while (c0 < N) {
        const bool first = (c0 == 0);
        int64_t w = N - c0 < nc ? N - c0 : nc;
        if (first && filter && w > kFirstChunk && N >= 4 * kFirstChunk) w = kFirstChunk;   // short bootstrap chunk
        const bool last = (c0 + w >= N);
        const bool emit = last && j.emit;
        int rc;
        if (!filter) {
            rc = gemm(c0, w, nullptr, nullptr);
            if (rc) return rc;
            rc = launch_select(chunk, M, w, w, c0, k, carry, first, last, j.idx_base, j.top_score, j.top_idx, st, thr);
        } else if (first) {
            // bootstrap chunk: plain GEMM, every group present, empty carry (all-zero keys sort last)
            rc = gemm(c0, w, nullptr, nullptr);
            if (rc) return rc;
            if (hipMemsetAsync(carry, 0, (size_t)M * k * 8, st) != hipSuccess) {
                isx_set_error("%s: hipMemsetAsync failed", j.who);
                return ISX_ERR_HIP;
            }
            rc = launch_select_groups(chunk, nullptr /* all groups present */, M, w, w, c0, k, carry, thr, emit ? 1 : (last ? 2 : 0), j.idx_base, j.top_score,
                                      j.top_idx, st, j.m_active, j.row_map, j.win, j.k_win);
        } else {
            rc = gemm(c0, w, thr, gflag);
            if (rc) return rc;
            rc = launch_select_groups(chunk, gflag, M, w, w, c0, k, carry, thr, emit ? 1 : (last ? 2 : 0), j.idx_base, j.top_score, j.top_idx, st, j.m_active,
                                      j.row_map, j.win, j.k_win);
        }
        if (rc) return rc;
        c0 += w;
    }
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

// Debug / A-B hook (not declared in include/isx.h): force a tile shape (0..3), -1 = automatic.
ISX_API void isx_debug_set_gemm_cfg(int c) { set_gemm_cfg(c); }

ISX_API int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0, "isx_cosine_sim: bad shape M=%lld N=%lld D=%d", (long long)M, (long long)N, D);
    ISX_REQUIRE((Q && G && sim) || M * N == 0, "isx_cosine_sim: null pointer");
    return launch_cosine_gemm(Q, M, G, N, D, sim, N, (hipStream_t)stream);
}

ISX_API size_t isx_cosine_topk_workspace(int64_t M, int64_t N, int D, int k) {
    (void)D;
    if (M <= 0 || N <= 0 || k <= 0) return 256;
    return topk_fixed_bytes(M, k) + topk_chunk_bytes(M, topk_recommended_chunk(M, N));
    // minimum accepted by isx_cosine_topk: the same formula with nc = min(N, 128)
}

ISX_API int isx_cosine_topk(const float* Q, int64_t M, const float* G, int64_t N, int D, int k, int64_t idx_base,
                            float* top_score, int64_t* top_idx, void* ws, size_t ws_bytes, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0, "isx_cosine_topk: bad shape M=%lld N=%lld D=%d", (long long)M, (long long)N, D);
    ISX_REQUIRE(k >= 1 && k <= kSelectMaxK, "isx_cosine_topk: k=%d outside [1,%d]", k, kSelectMaxK);
    ISX_REQUIRE(idx_base >= 0 && idx_base + N <= 0xFFFFFFFFll && N <= 0x7FFFFFFFll, "isx_cosine_topk: gallery indices must stay below 2^32");
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(Q && top_score && top_idx && (G || N == 0), "isx_cosine_topk: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) return launch_select(nullptr, M, 0, 0, 0, k, nullptr, true, true, idx_base, top_score, top_idx, st);
    TopkJob j{};
    j.who = "isx_cosine_topk";
    j.Q = Q; j.G = G; j.M = M; j.N = N; j.D = D; j.k = k;
    j.idx_base = idx_base; j.top_score = top_score; j.top_idx = top_idx; j.emit = true;
    j.ws = ws; j.ws_bytes = ws_bytes; j.st = st;
    return run_topk_chunks(j);
}

#if ISX_STAMPS
// lab builds only: where the convolution GEMM's workgroups record their phase stamps (8 x u64 per workgroup); nullptr = off
extern "C" __attribute__((visibility("default"))) int isx_debug_set_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(isx::g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
