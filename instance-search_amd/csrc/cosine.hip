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
// wave 64x64 = 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs), BK = 32.  Q and G tiles
// are staged K-major in LDS ([k][row], row stride 129 floats): the staging loads are
// 16 B/lane with 8 lanes covering one 128-B row segment (coalesced), the transposed
// ds_write_b32 are bank-conflict-free by the odd stride, and the MFMA operand reads are
// 32 consecutive floats per half-wave (conflict-free ds_read_b32).  Global loads of
// tile t+1 are issued before the MFMAs of tile t (register prefetch).  Workgroups are
// remapped XCD-aware: each XCD walks a contiguous range of tiles, ordered so that
// concurrently resident tiles share Q / G panels in that XCD's L2.
#include "isx_internal.hpp"

namespace isx {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BM + 1;            // K-major LDS row stride (floats), odd
constexpr int GROUP_N = 16;            // n-tiles per scheduling group

struct TileMap {
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void tile_of_block(const TileMap tm, int& tile_m, int& tile_n) {
    // bijective XCD remap (blocks b and b+8 share an XCD): XCD x gets a contiguous id range
    const int nwg = tm.tiles_m * tm.tiles_n;
    const int b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    // grouped order: GROUP_N n-tiles wide, all m-tiles tall, n fastest inside a group row
    const int per_group = GROUP_N * tm.tiles_m;
    const int gid = wg / per_group;
    const int first_n = gid * GROUP_N;
    const int gsz = min(GROUP_N, tm.tiles_n - first_n);
    const int within = wg - gid * per_group;
    tile_m = within / gsz;
    tile_n = first_n + within % gsz;
}

template <bool ALIGNED>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t rows, int D, int64_t row0, int k0,
                                          float4 (&reg)[4]) {
    // 128 rows x 32 k = 1024 float4; thread t takes idx = j*256 + t: row = idx/8, chunk = idx%8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 256 + threadIdx.x;
        int64_t r = row0 + (idx >> 3);
        r = r < rows ? r : rows - 1;                       // clamp: rows past the edge are never stored
        const int k = k0 + ((idx & 7) << 2);
        const float* src = P + r * D + k;
        if (ALIGNED) {
            reg[j] = (k < D) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            reg[j].x = (k + 0 < D) ? src[0] : 0.f;
            reg[j].y = (k + 1 < D) ? src[1] : 0.f;
            reg[j].z = (k + 2 < D) ? src[2] : 0.f;
            reg[j].w = (k + 3 < D) ? src[3] : 0.f;
        }
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&reg)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = j * 256 + threadIdx.x;
        const int r = idx >> 3, k = (idx & 7) << 2;
        T[(k + 0) * LDT + r] = reg[j].x;
        T[(k + 1) * LDT + r] = reg[j].y;
        T[(k + 2) * LDT + r] = reg[j].z;
        T[(k + 3) * LDT + r] = reg[j].w;
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void cosine_gemm_kernel(const float* __restrict__ Q, int64_t M,
                                                          const float* __restrict__ G, int64_t N, int D,
                                                          float* __restrict__ C, int64_t ldc, TileMap tm) {
    __shared__ float lds[2 * BK * LDT];
    float* As = lds;
    float* Bs = lds + BK * LDT;

    int tile_m, tile_n;
    tile_of_block(tm, tile_m, tile_n);
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 ra[4], rb[4];
    const int nk = (D + BK - 1) / BK;
    load_tile<ALIGNED>(Q, M, D, m0, 0, ra);
    load_tile<ALIGNED>(G, N, D, n0, 0, rb);
    store_tile(As, ra);
    store_tile(Bs, rb);
    __syncthreads();

    const float* a_base = As + half * LDT + wm * 64 + l31;
    const float* b_base = Bs + half * LDT + wn * 64 + l31;

    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            load_tile<ALIGNED>(Q, M, D, m0, (kt + 1) * BK, ra);
            load_tile<ALIGNED>(G, N, D, n0, (kt + 1) * BK, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = a_base[(2 * kk) * LDT];
            const float a1 = a_base[(2 * kk) * LDT + 32];
            const float b0 = b_base[(2 * kk) * LDT];
            const float b1 = b_base[(2 * kk) * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_tile(As, ra);
            store_tile(Bs, rb);
            __syncthreads();
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M && n < N) C[m * ldc + n] = acc[i][j][e];
            }
        }
    }
}

int launch_cosine_gemm(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                       hipStream_t st) {
    if (M == 0 || N == 0) return ISX_OK;
    TileMap tm;
    const int64_t tmm = (M + BM - 1) / BM, tnn = (N + BN - 1) / BN;
    if (tmm * tnn >= (1ll << 31)) { isx_set_error("cosine gemm: %lld x %lld tiles exceed the grid limit", (long long)tmm, (long long)tnn); return ISX_ERR_ARG; }
    tm.tiles_m = (int)tmm;
    tm.tiles_n = (int)tnn;
    const bool aligned = (D % 4 == 0) && (((uintptr_t)Q | (uintptr_t)G) % 16 == 0);
    const dim3 grid((unsigned)(tmm * tnn)), block(256);
    if (aligned) hipLaunchKernelGGL(cosine_gemm_kernel<true>, grid, block, 0, st, Q, M, G, N, D, C, ldc, tm);
    else hipLaunchKernelGGL(cosine_gemm_kernel<false>, grid, block, 0, st, Q, M, G, N, D, C, ldc, tm);
    ISX_CHECK_LAUNCH("cosine_gemm");
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

ISX_API int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && D > 0, "isx_cosine_sim: bad shape M=%lld N=%lld D=%d", (long long)M, (long long)N, D);
    ISX_REQUIRE((Q && G && sim) || M * N == 0, "isx_cosine_sim: null pointer");
    return launch_cosine_gemm(Q, M, G, N, D, sim, N, (hipStream_t)stream);
}

// Workspace layout of isx_cosine_topk: [ carry keys: M*k u64 | score chunk: M*Nc f32 ].
static size_t topk_carry_bytes(int64_t M, int k) { return (((size_t)M * k * 8) + 255) & ~(size_t)255; }

ISX_API size_t isx_cosine_topk_workspace(int64_t M, int64_t N, int D, int k) {
    (void)D;
    if (M <= 0 || N <= 0 || k <= 0) return 256;
    // recommended: whole matrix if it is <= 1 GiB, else column chunks of ~1 GiB
    // (multiple of 2048 columns so that every chunk launch fills the chip evenly).
    const size_t budget = (size_t)1 << 30;
    int64_t nc = N;
    if ((size_t)M * N * 4 > budget) {
        nc = (int64_t)(budget / ((size_t)M * 4));
        nc = nc / 2048 * 2048;
        if (nc < 2048) nc = 2048;
        if (nc > N) nc = N;
    }
    return topk_carry_bytes(M, k) + (size_t)M * nc * 4;
    // minimum accepted by isx_cosine_topk: carry + M * min(N, 128) * 4 bytes
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
    const size_t carry_b = topk_carry_bytes(M, k);
    const int64_t min_nc = N < 128 ? N : 128;
    if (!ws || ((uintptr_t)ws % 256) != 0 || ws_bytes < carry_b + (size_t)M * min_nc * 4) {
        isx_set_error("isx_cosine_topk: workspace of %zu bytes too small or misaligned (need >= %zu, 256-B aligned)", ws_bytes,
                      carry_b + (size_t)M * min_nc * 4);
        return ISX_ERR_WORKSPACE;
    }
    uint64_t* carry = (uint64_t*)ws;
    float* chunk = (float*)((char*)ws + carry_b);
    int64_t nc = (int64_t)((ws_bytes - carry_b) / ((size_t)M * 4));
    if (nc >= N) nc = N;
    else nc = nc >= 128 ? nc / 128 * 128 : nc;            // whole tiles per chunk
    for (int64_t c0 = 0; c0 < N; c0 += nc) {
        const int64_t w = (N - c0 < nc) ? N - c0 : nc;
        int rc = launch_cosine_gemm(Q, M, G + c0 * D, w, D, chunk, w, st);
        if (rc) return rc;
        const bool first = (c0 == 0), last = (c0 + w >= N);
        rc = launch_select(chunk, M, w, w, c0, k, carry, first, last, idx_base, top_score, top_idx, st);
        if (rc) return rc;
    }
    return ISX_OK;
}
