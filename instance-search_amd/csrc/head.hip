// head.hip -- the descriptor head's Linear(100352 -> 2048) of siamese training (reference model/siamese.py:104-114: ONE 822 MB weight),
// forward for ALL micro-batches of a step at once.
//
// The reference (and torch) run the head once per micro-batch of 8 triplets: 24 rows against an 822 MB weight, i.e. the weight crosses
// HBM sixteen times per optimizer step (8 forward + 8 input-gradient passes, 13 GB).  Batching the rows needs a GEMM whose result for a
// row does NOT depend on how many rows ride along (the update has to stay bit-identical for 1, 2, 4, 8 ranks, isx/dp.py); a library GEMM
// picks its tiling -- and with it its summation order -- from the shape.  Here:
//
//   isx_head_linear_fwd     y[m][n] = bias[n] + sum_s ( sum over k in split s of x[m][k] * w[n][k] ),  S = splits(K) a function of K ONLY;
//                           every partial is a k-ordered fp32 fma chain (v_mfma_f32_32x32x2_f32), the S partials are added in split order.
//                           x arrives TRANSPOSED (K, Mp): K-major, tiles go to LDS as they lie; w as stored by nn.Linear (N, K): staged
//                           through the transposing LDS store of the score GEMM.  grid = (Mp / 64) x (N / 64) x S blocks.
//   isx_head_linear_dgrad   dx = dy . w: the TN GEMM of the weight gradients (wgrad_kernel.hpp) on (dy^T, w), both K-major as stored; no split.
//   isx_head_sgd_step       the weight gradient dW = dY^T X over the rows of the whole mini-batch AND the SGD update of the 822 MB weight as ONE kernel:
//                           the TN GEMM's epilogue reads the weight / momentum tile, applies weight decay, momentum and the step and writes both back.
//                           No dW tensor (822 MB written, then re-read by the optimizer), no separate optimizer pass over W, dW and the momentum:
//                           3.3 GB of HBM traffic per step instead of 5.8 GB.
//   isx_colsum_leaves       per-micro-batch column sums (bias / Shift gradients kept apart per leaf): one thread per (leaf, column),
//                           rows added in order.
#include <stdlib.h>

#include "wgrad_kernel.hpp"

namespace isx {

constexpr int kHeadBK = 32;

// number of k splits: ~64 k-tiles each, at most 32 -- from K alone
static int head_splits(int64_t K) {
    int64_t s = K / (kHeadBK * 64);
    if (s > 32) s = 32;
    return (int)(s < 1 ? 1 : s);
}

// TM: 32-row MFMA tiles per wave along M (block tile 64 TM x 64).  TM = 2 / 3 cover Mp = 128 / 192 rows with ONE m-tile, so the weight is
// read once instead of once per 64 rows; the tile shape only groups outputs, every output is the same chain in any of them.
// ROWS: x is given as stored, (M, K) row-major (xT = x, Mrows = M): its tiles take the transposing LDS store of the weight tiles; rows past M read
// zeros through the buffer descriptor.  Otherwise xT is (K, Mp), K-major, and goes to the LDS as it lies.  Same chain per output either way.
#ifndef ISX_LB_HEADFWD
#define ISX_LB_HEADFWD 2
#endif
#ifndef ISX_SGD_AHEAD
#define ISX_SGD_AHEAD 1         // MFMA tiles of the fused gradient + SGD kernel whose w / momentum values are requested before the first store.  MEASURED (round 6,
                                // tools/head_lab.py, 192 x 100352 x 2048): 1 tile / 3 workgroups per CU 0.954 ms (3.45 TB/s), 2 tiles / 3 WG 0.957, all 4 tiles / 2 WG 0.996 --
                                // bytes in flight are not what holds the kernel at 3.4 TB/s (torch's fused elementwise SGD moves the same read + write mix at 3.7)
#endif
#ifndef ISX_SGD_NT
#define ISX_SGD_NT 2            // (0.957 -> 0.932 ms, round 6; A/B: 0) aux bits of the w / momentum loads and stores of the fused SGD epilogue (2 = nt: read once, written once per step)
#endif
#ifndef ISX_LB_SGD
#define ISX_LB_SGD 3
#endif
template <int TM, int TN = 1, bool ROWS = false>
__global__ __launch_bounds__(256, TM * TN >= 6 ? ISX_LB_HEADFWD : TM * TN >= 3 ? 3 : 4) void head_fwd_gemm_kernel(const float* __restrict__ xT, int Mp, const float* __restrict__ Wn, int N, int K, int kt_per,
                                                            float* __restrict__ part, int tiles_m, int Mrows = 0) {
    constexpr int BK = kHeadBK, BM = 64 * TM, BN = 64 * TN, LDA = ROWS ? BM + lds_pad(BK) : BM + 4, LDB = BN + lds_pad(BK);
    constexpr int CA = BM / 4, NA = ROWS ? BM * BK / 1024 : BK * CA / 256;
    __shared__ float lds[BK * (LDA + LDB)];
    float* As = lds;
    float* Bs = lds + BK * LDA;
    // row tiles fastest, and XCD x gets a contiguous range of tile ids (blocks b and b + 8 share an XCD): the row tiles that read the same slice of
    // the weight run next to each other on ONE XCD and share it in that L2
    const int nwg = (int)gridDim.x, bx = (int)blockIdx.x;
    const int xcd = bx & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bx >> 3);
    const int tile_m = wg % tiles_m, tile_n = wg / tiles_m, split = (int)blockIdx.y;
    const int m0 = tile_m * BM;
    const int64_t n0 = (int64_t)tile_n * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[NA], rb[BN * BK / 1024];
    auto load = [&](int kt) {
        const int64_t k0 = (int64_t)kt * BK;
        if constexpr (ROWS) {
            load_tile<true, BM, BK>(xT, Mrows, K, m0, (int)k0, ra);
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int idx = j * 256 + threadIdx.x;
                ra[j] = *reinterpret_cast<const float4*>(xT + (k0 + idx / CA) * Mp + m0 + ((idx % CA) << 2));
            }
        }
        load_tile<true, BN, BK>(Wn, N, K, n0, (int)k0, rb);
    };
    auto store = [&]() {
        if constexpr (ROWS) {
            store_tile<BM, BK>(As, ra);
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int idx = j * 256 + threadIdx.x;
                *reinterpret_cast<float4*>(As + (idx / CA) * LDA + ((idx % CA) << 2)) = ra[j];
            }
        }
        store_tile<BN, BK>(Bs, rb);
    };
    const int nk_all = K / BK;
    const int kt0 = split * kt_per;
    const int kt1 = kt0 + kt_per < nk_all ? kt0 + kt_per : nk_all;
    if (kt0 < kt1) {
        load(kt0);
        store();
        __syncthreads();
        const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
        const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) load(kt + 1);
            mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
            __syncthreads();
            if (more) {
                store();
                __syncthreads();
            }
        }
    }
    float* P = part + (int64_t)split * Mp * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + wn * (32 * TN) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const float v = acc[i][j][e];
                P[(int64_t)row * N + col] = v;
            }
        }
}

// y[m][n] = (part[0] + part[1] + ... + part[S-1])[m][n] + bias[n], rows m < M only
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, int S, int64_t plane, int64_t total, int N, const float* __restrict__ bias,
                                                          float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = part[i];
    for (int s = 1; s < S; ++s) v += part[s * plane + i];
    y[i] = bias ? v + bias[i % N] : v;
}

// ---- weight gradient + SGD update of the head's Linear in one kernel ---------------------------------------------------------------------
// g[n][k] = sum_r dy[r][n] * x[r][k]  (ONE fp32 fma chain over the rows in row order: the bits of isx_conv_wgrad_nhwc on the same rows, and the
// same at any number of ranks -- every rank holds all rows of the mini-batch), then torch.optim.SGD's update of element (n, k)
// (torch/optim/sgd.py _single_tensor_sgd; reference: optim.SGD in train/siamese_descriptor.py:136-139 stepped from utils/train_general.py:53):
//   g += weight_decay * w;   buf = first ? g : momentum * buf + (1 - dampening) * g;   w -= lr * (nesterov ? g + momentum * buf : buf)
// Tile 128 x 128 of the (N, K) weight per workgroup, R reduced in k-tiles of 32 rows (a tail of rows is zero-filled: fma(0, 0, acc) = acc).
struct SgdParams { float lr, momentum, dampening, weight_decay; int nesterov, first, use_momentum; };

// TM: 128 (2) or 64 (1) rows of the weight per tile -- a shard of a sharded head can be as narrow as 64 output features; the tile only groups outputs
template <int TM>
__global__ __launch_bounds__(256, ISX_LB_SGD) void head_sgd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t R, int N, int64_t K,
                                                       float* __restrict__ w, float* __restrict__ mom, SgdParams sp, int tiles_k) {
    constexpr int TN = 2, BK = 32, BM = 64 * TM, BN = 128, LDA = BM + 4, LDB = BN + 4;
    constexpr int CA = BM / 4, CB = BN / 4, NA = BK * CA / 256, NB = BK * CB / 256;
    __shared__ float lds[BK * (LDA + LDB)];
    float* As = lds;
    float* Bs = lds + BK * LDA;
    const int tile_n = (int)(blockIdx.x / tiles_k);
    const int64_t n0 = (int64_t)tile_n * BM, k0 = (int64_t)(blockIdx.x % tiles_k) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);

    const auto rw = conv_tile_rsrc(w, n0, N, K, BM);
    const auto rmom = conv_tile_rsrc(mom ? mom : w, n0, N, K, BM);
    f32x16 acc[TM][TN];
    zero_tiles(acc);

    float4 ra[NA], rb[NB];
    auto load = [&](int64_t r0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const int64_t r = r0 + idx / CA;
            ra[j] = r < R ? *reinterpret_cast<const float4*>(dy + r * N + n0 + ((idx % CA) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const int64_t r = r0 + idx / CB;
            rb[j] = r < R ? *reinterpret_cast<const float4*>(x + r * K + k0 + ((idx % CB) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    const int64_t nk = (R + BK - 1) / BK;
    if (nk > 0) {
        load(0);
        store();
        __syncthreads();
        const float* a_base = As + half * LDA + wm * (32 * TM) + l31;
        const float* b_base = Bs + half * LDB + wn * (32 * TN) + l31;
        for (int64_t kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
            if (more) load((kt + 1) * BK);
            mfma_ktile<TM, TN, BK, LDA, LDB>(a_base, b_base, acc);
            __syncthreads();
            if (more) {
                store();
                __syncthreads();
            }
        }
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); one 32-bit lane offset per MFMA tile
    // into the tile's rows of w / mom (buffer instructions: rows past N and columns past K fall outside the descriptor).
    // The kernel is HBM-bound (w and momentum read + written: 3.3 GB per step): what it needs is BYTES IN FLIGHT.  Round 5 read and wrote one MFMA
    // tile at a time -- 16 + 16 loads, a wait for them AND for the previous tile's stores (one counter on gfx9), 32 stores -- i.e. 8 KB in flight per
    // wave and four load + store round trips per workgroup.  ISX_SGD_AHEAD tiles (default: all TM x TN) are requested before the first store.
    const float one_minus_damp = 1.0f - sp.dampening;
    constexpr int NT = TM * TN, STEP = ISX_SGD_AHEAD < 1 ? 1 : (ISX_SGD_AHEAD > NT ? NT : ISX_SGD_AHEAD);
    const bool need_m = sp.use_momentum && !sp.first;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += STEP) {
        float wv[STEP][16], mv[STEP][16];
#pragma unroll
        for (int t = t0; t < t0 + STEP && t < NT; ++t) {
            const int i = t / TN, j = t % TN;
            const int64_t col = k0 + wn_u * (32 * TN) + j * 32 + l31;
            const unsigned lo = conv_lane_off(col, K, wm_u * (32 * TM) + i * 32 + 4 * half, K);
#pragma unroll
            for (int e = 0; e < 16; ++e) wv[t - t0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * K * 4), ISX_SGD_NT));
        }
        if (need_m) {
#pragma unroll
            for (int t = t0; t < t0 + STEP && t < NT; ++t) {
                const int i = t / TN, j = t % TN;
                const int64_t col = k0 + wn_u * (32 * TN) + j * 32 + l31;
                const unsigned lo = conv_lane_off(col, K, wm_u * (32 * TM) + i * 32 + 4 * half, K);
#pragma unroll
                for (int e = 0; e < 16; ++e) mv[t - t0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmom, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * K * 4), ISX_SGD_NT));
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // the batch's loads above its first store
#pragma unroll
        for (int t = t0; t < t0 + STEP && t < NT; ++t) {
            const int i = t / TN, j = t % TN;
            const int64_t col = k0 + wn_u * (32 * TN) + j * 32 + l31;
            const unsigned lo = conv_lane_off(col, K, wm_u * (32 * TM) + i * 32 + 4 * half, K);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float g = acc[i][j][e];
                if (sp.weight_decay != 0.0f) g = g + sp.weight_decay * wv[t - t0][e];
                float upd = g;
                if (sp.use_momentum) {
                    const float buf = sp.first ? g : sp.momentum * mv[t - t0][e] + one_minus_damp * g;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, buf), rmom, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * K * 4), ISX_SGD_NT);
                    upd = sp.nesterov ? g + sp.momentum * buf : buf;
                }
                const float nw = wv[t - t0][e] - sp.lr * upd;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nw), rw, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * K * 4), ISX_SGD_NT);
            }
        }
    }
}

// out[l][c] = sum over the R rows of leaf l of x[l * R + r][c], rows in order
__global__ __launch_bounds__(256) void colsum_leaves_kernel(const float* __restrict__ x, int R, int64_t C, float* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int leaf = (int)blockIdx.y;
    const float* p = x + (int64_t)leaf * R * C + c;
    float s = 0.0f;
    for (int r = 0; r < R; ++r) s += p[(int64_t)r * C];
    out[(int64_t)leaf * C + c] = s;
}

// Canonical tree sum of L rows (isx/dp.py tree_sum: sum(lo, hi) = sum(lo, mid) + sum(mid, hi), mid = lo + (hi - lo) / 2), all adds in registers.
template <int N> struct TreeSum {
    template <typename T> static __device__ __forceinline__ T of(const T* v) { return TreeSum<N / 2>::of(v) + TreeSum<N - N / 2>::of(v + N / 2); }
};
template <> struct TreeSum<1> {
    template <typename T> static __device__ __forceinline__ T of(const T* v) { return v[0]; }
};

template <int L, typename T>
__global__ __launch_bounds__(256) void tree_sum_rows_kernel(const T* __restrict__ rows, int64_t stride, int64_t n, T* __restrict__ out) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t)gridDim.x * 256) {
        T v[L];
#pragma unroll
        for (int l = 0; l < L; ++l) v[l] = rows[(int64_t)l * stride + c];
        out[c] = TreeSum<L>::of(v);
    }
}

__device__ __forceinline__ float4 operator+(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int L>
static void launch_tree_sum(const float* rows, int64_t stride, int64_t n, float* out, hipStream_t st) {
    const bool vec = n % 4 == 0 && stride % 4 == 0 && (((uintptr_t)rows | (uintptr_t)out) % 16) == 0;
    const int64_t items = vec ? n / 4 : n;
    const unsigned grid = (unsigned)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    if (vec)
        hipLaunchKernelGGL((tree_sum_rows_kernel<L, float4>), dim3(grid), dim3(256), 0, st, (const float4*)rows, stride / 4, items, (float4*)out);
    else
        hipLaunchKernelGGL((tree_sum_rows_kernel<L, float>), dim3(grid), dim3(256), 0, st, rows, stride, items, out);
}

}  // namespace isx

using namespace isx;

ISX_API int isx_tree_sum_rows(const float* rows, int L, int64_t stride, int64_t n, float* out, isx_stream_t stream) {
    ISX_REQUIRE(L >= 1 && L <= 16 && n >= 0 && stride >= 0, "isx_tree_sum_rows: bad shape L=%d stride=%lld n=%lld (1 <= L <= 16)", L, (long long)stride, (long long)n);
    if (n == 0) return ISX_OK;
    ISX_REQUIRE(rows && out && (L == 1 || stride >= n), "isx_tree_sum_rows: null pointer or overlapping rows");
    hipStream_t st = (hipStream_t)stream;
    switch (L) {
#define ISX_TS(l) case l: launch_tree_sum<l>(rows, stride, n, out, st); break;
        ISX_TS(1) ISX_TS(2) ISX_TS(3) ISX_TS(4) ISX_TS(5) ISX_TS(6) ISX_TS(7) ISX_TS(8) ISX_TS(9) ISX_TS(10) ISX_TS(11) ISX_TS(12) ISX_TS(13) ISX_TS(14) ISX_TS(15) ISX_TS(16)
#undef ISX_TS
    }
    ISX_CHECK_LAUNCH("isx_tree_sum_rows");
    return ISX_OK;
}

// Splits of the K dimension isx_head_linear_fwd uses (host arithmetic; sizes the workspace: splits * Mp * N floats).
ISX_API int isx_head_linear_splits(int64_t K) { return K > 0 ? head_splits(K) : 0; }

// y = x . w^T + bias for M rows at once, each row's value independent of M (see the file header).  xT: (K, Mp) = x transposed, Mp >= M a
// multiple of 64 (columns M .. Mp - 1 are padding, any finite values); w: (N, K) as nn.Linear stores it; bias: (N) or NULL; y: (M, N);
// ws: isx_head_linear_splits(K) * Mp * N floats.  K % 32 == 0, N % 64 == 0, 16-B aligned pointers.
ISX_API int isx_head_linear_fwd(const float* xT, int64_t M, int64_t Mp, int64_t K, const float* w, int N, const float* bias, float* y, float* ws, size_t ws_bytes,
                                isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && Mp >= M && Mp % 64 == 0 && Mp < (1 << 24) && K > 0 && K % 32 == 0 && K < (1ll << 31) && N > 0 && N % 64 == 0,
                "isx_head_linear_fwd: bad shape M=%lld Mp=%lld K=%lld N=%d (Mp %% 64 == 0, K %% 32 == 0, N %% 64 == 0)", (long long)M, (long long)Mp, (long long)K, N);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(xT && w && y && ws, "isx_head_linear_fwd: null pointer");
    ISX_REQUIRE((((uintptr_t)xT | (uintptr_t)w | (uintptr_t)ws) % 16) == 0, "isx_head_linear_fwd: xT, w and ws must be 16-B aligned");
    const int S = head_splits(K);
    ISX_REQUIRE(ws_bytes >= (size_t)S * (size_t)Mp * (size_t)N * 4, "isx_head_linear_fwd: workspace of %zu bytes, need %zu", ws_bytes, (size_t)S * (size_t)Mp * (size_t)N * 4);
    const int nk = (int)(K / kHeadBK), kt_per = (nk + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
    static const bool wide = [] { const char* e = getenv("ISX_HEAD_WIDE"); return !(e && e[0] == '0'); }();      // A/B: 192 x 128 tiles (default) vs 192 x 64
    if (Mp == 192 && N % 128 == 0 && wide)
        hipLaunchKernelGGL((head_fwd_gemm_kernel<3, 2>), dim3((unsigned)(N / 128), (unsigned)S), dim3(256), 0, st, xT, (int)Mp, w, N, (int)K, kt_per, ws, 1);
    else if (Mp == 192)
        hipLaunchKernelGGL(head_fwd_gemm_kernel<3>, dim3((unsigned)(N / 64), (unsigned)S), dim3(256), 0, st, xT, (int)Mp, w, N, (int)K, kt_per, ws, 1);
    else if (Mp == 128)
        hipLaunchKernelGGL(head_fwd_gemm_kernel<2>, dim3((unsigned)(N / 64), (unsigned)S), dim3(256), 0, st, xT, (int)Mp, w, N, (int)K, kt_per, ws, 1);
    else
        hipLaunchKernelGGL(head_fwd_gemm_kernel<1>, dim3((unsigned)((Mp / 64) * (N / 64)), (unsigned)S), dim3(256), 0, st, xT, (int)Mp, w, N, (int)K, kt_per, ws, (int)(Mp / 64));
    ISX_CHECK_LAUNCH("isx_head_linear_fwd(gemm)");
    const int64_t total = M * N;
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, S, Mp * (int64_t)N, total, N, bias, y);
    ISX_CHECK_LAUNCH("isx_head_linear_fwd(reduce)");
    return ISX_OK;
}

// The same with x as stored, (M, K) row-major: no transposed copy of the activation (77 MB per training step).  Tiles: up to 64 rows 64 x 64,
// up to 128 rows 128 x 64, above that 192 x 128 (192 x 64 when N is not a multiple of 128) over ceil(M / 192) row tiles -- the weight crosses HBM
// once per 192 rows (a 1024-row inference batch: 6 times, 4.9 GB for 421 GFLOP).  The tile shape only groups outputs: every output is the same S
// chains added in split order, so a row's result does not depend on M.  Bit-identical to isx_head_linear_fwd on the transposed rows.
static int64_t head_rows_padded(int64_t M) { return M <= 64 ? 64 : M <= 128 ? 128 : (M + 191) / 192 * 192; }

ISX_API size_t isx_head_linear_rows_workspace(int64_t M, int64_t K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return (size_t)head_splits(K) * (size_t)head_rows_padded(M) * (size_t)N * 4;
}

ISX_API int isx_head_linear_fwd_rows(const float* x, int64_t M, int64_t K, const float* w, int N, const float* bias, float* y, float* ws, size_t ws_bytes,
                                     isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && M < (1 << 24) && K > 0 && K % 32 == 0 && K < (1ll << 31) && N > 0 && N % 64 == 0 && (int64_t)192 * K * 4 < (1ll << 32),
                "isx_head_linear_fwd_rows: bad shape M=%lld K=%lld N=%d (K %% 32 == 0, N %% 64 == 0)", (long long)M, (long long)K, N);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(x && w && y && ws, "isx_head_linear_fwd_rows: null pointer");
    ISX_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)ws) % 16) == 0, "isx_head_linear_fwd_rows: x, w and ws must be 16-B aligned");
    const int S = head_splits(K);
    const int64_t Mp = head_rows_padded(M);
    ISX_REQUIRE(ws_bytes >= (size_t)S * (size_t)Mp * (size_t)N * 4, "isx_head_linear_fwd_rows: workspace of %zu bytes, need %zu", ws_bytes, (size_t)S * (size_t)Mp * (size_t)N * 4);
    const int nk = (int)(K / kHeadBK), kt_per = (nk + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
    if (Mp == 64)
        hipLaunchKernelGGL((head_fwd_gemm_kernel<1, 1, true>), dim3((unsigned)(N / 64), (unsigned)S), dim3(256), 0, st, x, (int)Mp, w, N, (int)K, kt_per, ws, 1, (int)M);
    else if (Mp == 128)
        hipLaunchKernelGGL((head_fwd_gemm_kernel<2, 1, true>), dim3((unsigned)(N / 64), (unsigned)S), dim3(256), 0, st, x, (int)Mp, w, N, (int)K, kt_per, ws, 1, (int)M);
    else if (N % 128 == 0)
        hipLaunchKernelGGL((head_fwd_gemm_kernel<3, 2, true>), dim3((unsigned)((Mp / 192) * (N / 128)), (unsigned)S), dim3(256), 0, st, x, (int)Mp, w, N, (int)K, kt_per, ws,
                           (int)(Mp / 192), (int)M);
    else
        hipLaunchKernelGGL((head_fwd_gemm_kernel<3, 1, true>), dim3((unsigned)((Mp / 192) * (N / 64)), (unsigned)S), dim3(256), 0, st, x, (int)Mp, w, N, (int)K, kt_per, ws,
                           (int)(Mp / 192), (int)M);
    ISX_CHECK_LAUNCH("isx_head_linear_fwd_rows(gemm)");
    const int64_t total = M * N;
    hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, S, Mp * (int64_t)N, total, N, bias, y);
    ISX_CHECK_LAUNCH("isx_head_linear_fwd_rows(reduce)");
    return ISX_OK;
}

// dx = dy . w for M rows at once (the input gradient of y = x . w^T): dx[m][k] = sum_n dy[m][n] * w[n][k].  Canonical summation (round 5): the N
// output features are cut into isx_head_groups(N) = 8 consecutive GROUPS (1 when N is not a multiple of 256); per group one k-ordered fp32 chain over
// its n, the group sums added in group order into a second accumulator -- whatever M.  A data-parallel run shards the head by output features:
// rank r holds the groups [r G / P, (r + 1) G / P) of w, computes THEIR chains for all rows (isx_head_linear_dgrad_parts) and the owner of a row adds
// the G pieces in group order: the same bits as this call on one GPU.  dyT: (N, Mp) = dy TRANSPOSED, zero-padded to Mp (a multiple of 64); w: (N, K);
// dx: (Mp, K) (rows >= M: padding).  K % 64 == 0.  Both operands are K-major for this contraction (rows indexed by n): tiles go to LDS as they lie.
static int head_groups(int64_t N) { return N % 256 == 0 ? 8 : 1; }
ISX_API int isx_head_groups(int64_t N) { return N > 0 ? head_groups(N) : 0; }

template <bool FOLD>
static void launch_head_dgrad(const float* dyT, int64_t Mp, int64_t Nrows, const float* w, int64_t K, float* out, int splits, int kt_per, int fold_kt, hipStream_t st) {
    WgradGeom g;
    g.ident = 1; g.H = g.W = g.Ho = g.Wo = 1; g.stride = 1;
    if (Mp == 192)            // all rows in ONE 192 x 64 tile: the weight is read once
        hipLaunchKernelGGL((wgrad_gemm_kernel<3, 1, FOLD>), dim3((unsigned)(K / 64), 1, (unsigned)splits), dim3(256), 0, st, dyT, Nrows, (int)Mp, w, (int)K, g, 1, out, K,
                           (int)(K / 64), kt_per, splits, (float*)nullptr, fold_kt);
    else if (Mp % 128 == 0 && K % 128 == 0)
        hipLaunchKernelGGL((wgrad_gemm_kernel<2, 2, FOLD>), dim3((unsigned)((Mp / 128) * (K / 128)), 1, (unsigned)splits), dim3(256), 0, st, dyT, Nrows, (int)Mp, w, (int)K,
                           g, 1, out, K, (int)(K / 128), kt_per, splits, (float*)nullptr, fold_kt);
    else
        hipLaunchKernelGGL((wgrad_gemm_kernel<1, 1, FOLD>), dim3((unsigned)((Mp / 64) * (K / 64)), 1, (unsigned)splits), dim3(256), 0, st, dyT, Nrows, (int)Mp, w, (int)K, g,
                           1, out, K, (int)(K / 64), kt_per, splits, (float*)nullptr, fold_kt);
}

ISX_API int isx_head_linear_dgrad(const float* dyT, int64_t Mp, int N, const float* w, int64_t K, float* dx, isx_stream_t stream) {
    ISX_REQUIRE(Mp >= 0 && Mp % 64 == 0 && Mp < (1 << 24) && N > 0 && K > 0 && K % 64 == 0 && K < (1ll << 31),
                "isx_head_linear_dgrad: bad shape Mp=%lld N=%d K=%lld (Mp %% 64 == 0, K %% 64 == 0)", (long long)Mp, N, (long long)K);
    if (Mp == 0) return ISX_OK;
    ISX_REQUIRE(dyT && w && dx, "isx_head_linear_dgrad: null pointer");
    ISX_REQUIRE((((uintptr_t)dyT | (uintptr_t)w | (uintptr_t)dx) % 16) == 0, "isx_head_linear_dgrad: pointers must be 16-B aligned");
    const int nk = (N + 31) / 32, G = head_groups(N);
    if (G > 1) launch_head_dgrad<true>(dyT, Mp, N, w, K, dx, 1, nk, nk / G, (hipStream_t)stream);
    else launch_head_dgrad<false>(dyT, Mp, N, w, K, dx, 1, nk, 0, (hipStream_t)stream);
    ISX_CHECK_LAUNCH("isx_head_linear_dgrad");
    return ISX_OK;
}

// The chains of `groups` consecutive groups of Ng output features each, NOT added: parts[g][m][k] = sum over the n of group g of dy[m][n] * w[n][k].
// dyT: (groups * Ng, Mp) = this rank's rows of dy^T, w: (groups * Ng, K) = its rows of the weight, parts: (groups, Mp, K).  Ng % 32 == 0, K % 64 == 0.
// Adding the parts of ALL groups in group order gives isx_head_linear_dgrad's result bit for bit.
ISX_API int isx_head_linear_dgrad_parts(const float* dyT, int64_t Mp, int Ng, int groups, const float* w, int64_t K, float* parts, isx_stream_t stream) {
    ISX_REQUIRE(Mp >= 0 && Mp % 64 == 0 && Mp < (1 << 24) && Ng > 0 && Ng % 32 == 0 && groups >= 1 && groups <= 65535 && K > 0 && K % 64 == 0 && K < (1ll << 31),
                "isx_head_linear_dgrad_parts: bad shape Mp=%lld Ng=%d groups=%d K=%lld (Mp %% 64 == 0, Ng %% 32 == 0, K %% 64 == 0)", (long long)Mp, Ng, groups, (long long)K);
    if (Mp == 0) return ISX_OK;
    ISX_REQUIRE(dyT && w && parts, "isx_head_linear_dgrad_parts: null pointer");
    ISX_REQUIRE((((uintptr_t)dyT | (uintptr_t)w | (uintptr_t)parts) % 16) == 0, "isx_head_linear_dgrad_parts: pointers must be 16-B aligned");
    // split s of the TN kernel owns the k-tiles [s * kt_per, (s + 1) * kt_per) of the reduction and writes its own (Mp, K) partial
    launch_head_dgrad<false>(dyT, Mp, (int64_t)groups * Ng, w, K, parts, groups, Ng / 32, 0, (hipStream_t)stream);
    ISX_CHECK_LAUNCH("isx_head_linear_dgrad_parts");
    return ISX_OK;
}

// Weight gradient of y = x . w^T over R rows and torch.optim.SGD's update of w, fused (see head_sgd_kernel): dy: (R, N), x: (R, K), w: (N, K) updated
// in place, mom: (N, K) momentum buffer updated in place (NULL when momentum == 0); first != 0: the buffer is (re)initialised with the gradient, as
// torch does on the first step.  N % 64 == 0, K % 128 == 0, 16-B aligned pointers.  (w, mom, dy may be the rows / columns of ONE shard of the layer.)  (Reference: the optimizer step of utils/train_general.py:53
// on the Linear of model/siamese.py:104-114.)
ISX_API int isx_head_sgd_step(const float* dy, const float* x, int64_t R, int N, int64_t K, float* w, float* mom, int first, float lr, float momentum,
                              float dampening, float weight_decay, int nesterov, isx_stream_t stream) {
    ISX_REQUIRE(R >= 0 && N > 0 && N % 64 == 0 && K > 0 && K % 128 == 0 && (int64_t)128 * K * 4 < (1ll << 31),
                "isx_head_sgd_step: bad shape R=%lld N=%d K=%lld (N %% 64 == 0, K %% 128 == 0, 128 rows of w below 2 GiB)", (long long)R, N, (long long)K);
    ISX_REQUIRE(w && (R == 0 || (dy && x)) && (momentum == 0.0f || mom), "isx_head_sgd_step: null pointer");
    ISX_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)w | (uintptr_t)mom) % 16) == 0, "isx_head_sgd_step: pointers must be 16-B aligned");
    ISX_REQUIRE(!(nesterov && (momentum <= 0.0f || dampening != 0.0f)), "isx_head_sgd_step: Nesterov momentum requires a momentum and zero dampening");
    const int bm = N % 128 == 0 ? 128 : 64;
    const int64_t tiles = (int64_t)(N / bm) * (K / 128);
    ISX_REQUIRE(tiles < (1ll << 31), "isx_head_sgd_step: too many tiles");
    SgdParams sp;
    sp.lr = lr; sp.momentum = momentum; sp.dampening = dampening; sp.weight_decay = weight_decay;
    sp.nesterov = nesterov ? 1 : 0; sp.first = first ? 1 : 0; sp.use_momentum = momentum != 0.0f ? 1 : 0;
    if (bm == 128) hipLaunchKernelGGL(head_sgd_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, dy, x, R, N, K, w, mom, sp, (int)(K / 128));
    else hipLaunchKernelGGL(head_sgd_kernel<1>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, dy, x, R, N, K, w, mom, sp, (int)(K / 128));
    ISX_CHECK_LAUNCH("isx_head_sgd_step");
    return ISX_OK;
}

// out[l][c] = sum_{r < R} x[l * R + r][c]: the column sums of `leaves` consecutive groups of R rows (per-micro-batch bias / Shift gradients),
// rows added in order.  x: (leaves * R, C); out: (leaves, C).
ISX_API int isx_colsum_leaves(const float* x, int leaves, int R, int64_t C, float* out, isx_stream_t stream) {
    ISX_REQUIRE(leaves >= 0 && leaves <= 65535 && R >= 0 && C >= 0, "isx_colsum_leaves: bad shape leaves=%d R=%d C=%lld", leaves, R, (long long)C);
    if (leaves == 0 || C == 0) return ISX_OK;
    ISX_REQUIRE(out && (R == 0 || x), "isx_colsum_leaves: null pointer");
    hipLaunchKernelGGL(colsum_leaves_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)leaves), dim3(256), 0, (hipStream_t)stream, x, R, C, out);
    ISX_CHECK_LAUNCH("isx_colsum_leaves");
    return ISX_OK;
}
