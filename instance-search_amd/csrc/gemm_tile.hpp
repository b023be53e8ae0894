// gemm_tile.hpp -- device helpers shared by the fp32-MFMA GEMM (cosine.hip) and the implicit-GEMM 3x3 convolution
// (conv.hip): XCD-aware tile mapping and the K-major LDS staging of operand tiles.
#pragma once
#ifndef ISX_SIMPLE_KLOOP
#define ISX_SIMPLE_KLOOP 0      // A/B: 1 = the plain per-step k loop on every tile shape
#endif
#ifndef ISX_KLOOP_PREFETCH
#define ISX_KLOOP_PREFETCH 1    // A/B: 1 = operand fragments of k-step kk + 1 fetched before the MFMAs of step kk (large tiles; +0.7-1.5 % on the big GEMMs)
#endif
#include <type_traits>

#include "isx_internal.hpp"

namespace isx {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef ISX_GROUP_N
#define ISX_GROUP_N 16
#endif
constexpr int GROUP_N = ISX_GROUP_N;            // n-tiles per scheduling group

struct TileMap {
    int tiles_m, tiles_n;
    const int* m_active;       // optional device scalar: only rows < *m_active are live (fast.hip fallback)
};

// b / nwg: block index and block count of the tile range (defaults: the whole grid)
__device__ __forceinline__ void tile_of_block(const TileMap tm, int& tile_m, int& tile_n, int b, int nwg) {
    // bijective XCD remap (blocks b and b+8 share an XCD): XCD x gets a contiguous id range
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

__device__ __forceinline__ void tile_of_block(const TileMap tm, int& tile_m, int& tile_n) {
    tile_of_block(tm, tile_m, tile_n, (int)blockIdx.x, tm.tiles_m * tm.tiles_n);
}

// Raw buffer descriptor over [base, base + nbytes) from WAVE-UNIFORM inputs (readfirstlane makes that provable to hipcc: no waterfall
// loops around the buffer instructions).  The two address halves go through unsigned temporaries: readfirstlane returns int, and
// a sign-extended low half would corrupt the high half of the pointer.
__device__ __forceinline__ auto uniform_rsrc(const void* base, int64_t nbytes) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)base);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
    const unsigned nb = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nbytes < 0xFFFFFFFFll ? (nbytes > 0 ? nbytes : 0) : 0xFFFFFFFFll));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | (uintptr_t)lo), 0, (int)nb, 0x00020000);
}

// ROWS x BK k = ROWS*CH float4; thread t takes idx = j*256 + t: row = idx/CH, chunk = idx%CH.
// ALIGNED: 16-B aligned rows AND D a multiple of BK (unconditional 16-B loads); otherwise scalar loads with a zero-filled k tail.
// AUX: cache-policy bits of the buffer loads of the ALIGNED path (0 = default, 2 = nt: a stream that no later access of this CU re-reads)
template <bool ALIGNED, int ROWS, int BK, int AUX = 0>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int64_t rows, int D, int64_t row0, int k0,
                                          float4 (&reg)[ROWS * BK / 1024]) {
    constexpr int CH = BK / 4;               // 16-B chunks per staged row
    if (ALIGNED) {
        // ALIGNED: D % BK == 0, 16-B aligned rows.  Buffer loads: a wave-uniform descriptor of the tile's rows (loop invariant), one
        // constant 32-bit offset per staged chunk, the k offset as SGPR: no address arithmetic inside the k loop.  Rows past the edge
        // fall outside the descriptor and load zeros (they are never stored).
        const int64_t left = rows - row0;
        const auto rs = uniform_rsrc(P + row0 * D, (left < ROWS ? left : ROWS) * (int64_t)D * 4);
#pragma unroll
        for (int j = 0; j < ROWS * CH / 256; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const unsigned vo = (unsigned)(idx / CH) * (unsigned)D * 4u + (unsigned)((idx % CH) << 4);
            reg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)k0 * 4u, AUX));
        }
        return;
    }
    if ((D & 3) == 0 && ((uintptr_t)P & 15) == 0) {
        // D a multiple of 4 but not of BK (class-score descriptors: D = 464) on 16-B aligned rows: the same buffer loads; in the last
        // k-tile the chunks past D -- they would read the next row -- are sent outside the descriptor and load zeros (fma(0, 0, acc) = acc)
        const int64_t left = rows - row0;
        const auto rs = uniform_rsrc(P + row0 * D, (left < ROWS ? left : ROWS) * (int64_t)D * 4);
#pragma unroll
        for (int j = 0; j < ROWS * CH / 256; ++j) {
            const int idx = j * 256 + threadIdx.x;
            const unsigned vo = k0 + ((idx % CH) << 2) < D ? (unsigned)(idx / CH) * (unsigned)D * 4u + (unsigned)((idx % CH) << 4) : 0x80000000u;
            reg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)k0 * 4u, 0));
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < ROWS * CH / 256; ++j) {
        const int idx = j * 256 + threadIdx.x;
        int64_t r = row0 + (idx / CH);
        r = r < rows ? r : rows - 1;                       // clamp: rows past the edge are never stored
        const int k = k0 + ((idx % CH) << 2);
        const float* src = P + r * D + k;
        reg[j].x = (k + 0 < D) ? src[0] : 0.f;
        reg[j].y = (k + 1 < D) ? src[1] : 0.f;
        reg[j].z = (k + 2 < D) ? src[2] : 0.f;
        reg[j].w = (k + 3 < D) ? src[3] : 0.f;
    }
}

// K-major LDS image [k][row] with row stride ROWS + pad.  A staging lane (chunk c = idx % CH, row r = idx / CH) writes
// element (4c + i, r): bank = ((4c + i) * stride + r) mod 32.  BK = 32 (CH = 8, 4 rows per 32-lane group): stride = 1 mod 32 gives
// 4c + r, all distinct.  BK = 16 (CH = 4, 8 rows per group): stride = 1 would give 4c + r with 4-way collisions (measured:
// SQ_LDS_BANK_CONFLICT = 25 % of the LDS cycles of the 128x128 kernel); stride = 2 mod 32 gives 8c + r, all distinct.  Operand
// reads are 32 consecutive floats per half-wave for either stride.
__host__ __device__ constexpr int lds_pad(int BK) { return BK == 16 ? 2 : 1; }

template <int ROWS, int BK>
__device__ __forceinline__ void store_tile(float* __restrict__ T, const float4 (&reg)[ROWS * BK / 1024]) {
    constexpr int CH = BK / 4;
    constexpr int LD = ROWS + lds_pad(BK);                  // K-major stride chosen for conflict-free transposed writes
#pragma unroll
    for (int j = 0; j < ROWS * CH / 256; ++j) {
        const int idx = j * 256 + threadIdx.x;
        const int r = idx / CH, k = (idx % CH) << 2;
        T[(k + 0) * LD + r] = reg[j].x;
        T[(k + 1) * LD + r] = reg[j].y;
        T[(k + 2) * LD + r] = reg[j].z;
        T[(k + 3) * LD + r] = reg[j].w;
    }
}

// Buffer descriptor of the rows [m0, min(m0 + BM, M)) of a row-major (M, ldc) fp32 matrix: wave-uniform by construction (kernel
// arguments and the tile index), so hipcc keeps it in SGPRs (no waterfall loop around the buffer instructions).
__device__ __forceinline__ auto conv_tile_rsrc(const float* base, int64_t m0, int64_t M, int64_t ldc, int BM) {
    const int64_t rows = (M - m0) < BM ? (M - m0) : BM;
    return uniform_rsrc(base + m0 * ldc, rows * ldc * 4);
}
// byte offset of (tile row `row`, column ncol) inside that descriptor; a column past N is sent outside it
__device__ __forceinline__ unsigned conv_lane_off(int64_t ncol, int64_t N, int row, int64_t ldc) {
    return ncol < N ? (unsigned)((row * ldc + ncol) * 4) : 0x80000000u;
}


// Convolution epilogue of one wave's TM x TN accumulator tiles through BUFFER instructions: y = act(acc + bias[n] (+ residual)).
// rc / rr: descriptors of the block tile's rows of C / the residual; row0 = the wave's first tile row (wave-uniform).
#ifndef ISX_RES_NT
#define ISX_RES_NT 0            // A/B: aux bits of the residual loads of the convolution epilogues (2 = nt: a residual is read once)
#endif
#ifndef ISX_EPI_LOADS_FIRST
#define ISX_EPI_LOADS_FIRST 1   // A/B (round 6): 1 = every residual / mask value of the wave's TM x TN tiles requested before the first store
#endif
// lane offset and SGPR row offsets of the 16 C elements a lane holds of one 32x32 MFMA tile
__device__ __forceinline__ constexpr int mfma_row_of(int e) { return (e & 3) + 8 * (e >> 2); }

// Residual (or mask) values of ALL of a wave's TM x TN accumulator tiles, requested back to back: ONE round trip to memory for the whole
// epilogue.  (Round 6: the per-tile form -- 16 loads, s_waitcnt vmcnt(0), 16 stores, next tile -- serialised TM x TN round trips, and on gfx9
// vmcnt counts stores too, so each wait also drained the previous tile's stores: four load + store latencies per 128x128 tile, as long as the
// matrix work of a K = 128 tile.)  The caller issues this ahead of its last k-tile's MFMAs where the registers allow.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_fetch(float (&rv)[TM][TN][16], const float* __restrict__ src, int64_t m0, int64_t M, int64_t n0, int64_t N, int64_t ldc,
                                               int BM, int row0, int col0, int l31, int half) {
    const auto rr = conv_tile_rsrc(src, m0, M, ldc, BM);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = (int)n0 + col0 + j * 32 + l31;
            const unsigned lo = conv_lane_off(ncol, N, row0 + i * 32 + 4 * half, ldc);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                rv[i][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo, (unsigned)(mfma_row_of(e) * ldc * 4), ISX_RES_NT));
        }
}

// Convolution epilogue of one wave's TM x TN accumulator tiles through BUFFER instructions: y = act(acc + bias[n] (+ residual)).
// rc / rr: descriptors of the block tile's rows of C / the residual; row0 = the wave's first tile row (wave-uniform).
// AHEAD: how many of the TM x TN tiles have their residual (or mask) values requested before the first of them is stored -- 16 VGPRs each.
// TM x TN (default): one round trip for the whole epilogue; kernels at their register bound pass fewer (the fused expand kernel at 106 of 128
// VGPRs: 2; the gradient kernels, which read a mask: 1, and with residual AND mask the per-tile form).  The bias values are always read first: a
// bias load issued between two tiles' stores waits for those stores (vmcnt counts both on gfx9).
template <int TM, int TN, int AHEAD = TM * TN>
__device__ __forceinline__ void conv_epilogue_buffers(const f32x16 (&acc)[TM][TN], float* __restrict__ C, const float* __restrict__ res,
                                                      const float* __restrict__ bias, int relu, int64_t m0, int64_t M, int64_t n0, int64_t N,
                                                      int64_t ldc, int BM, int row0, int col0, int l31, int half, const float* __restrict__ mask = nullptr) {
    // bias == nullptr: no bias;  mask (same layout as C): y = mask > 0 ? y : 0 (backward of the ReLU whose OUTPUT is `mask`, backward.hip)
    const auto rc = conv_tile_rsrc(C, m0, M, ldc, BM);
    float bias_pre[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ncol = (int)n0 + col0 + j * 32 + l31;
        bias_pre[j] = (bias && ncol < N) ? bias[ncol] : 0.0f;
    }
    if (ISX_EPI_LOADS_FIRST && !(res && mask)) {
        // one operand stream besides the accumulators (residual OR mask, or none): fetch AHEAD tiles' worth, then add / select / store them
        constexpr int NT = TM * TN, STEP = AHEAD < 1 ? 1 : (AHEAD > NT ? NT : AHEAD);
        const float* src = res ? res : mask;
        const auto rr = conv_tile_rsrc(src ? src : C, m0, M, ldc, BM);
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += STEP) {
            float rv[STEP][16];
            if (src) {
#pragma unroll
                for (int t = t0; t < t0 + STEP && t < NT; ++t) {
                    const int i = t / TN, j = t % TN;
                    const unsigned lo = conv_lane_off((int)n0 + col0 + j * 32 + l31, N, row0 + i * 32 + 4 * half, ldc);
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        rv[t - t0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo, (unsigned)(mfma_row_of(e) * ldc * 4), ISX_RES_NT));
                }
                __builtin_amdgcn_sched_barrier(0);           // keep the batch's loads above its first store
            }
#pragma unroll
            for (int t = t0; t < t0 + STEP && t < NT; ++t) {
                const int i = t / TN, j = t % TN;
                const unsigned lo = conv_lane_off((int)n0 + col0 + j * 32 + l31, N, row0 + i * 32 + 4 * half, ldc);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = src ? rv[t - t0][e] : 0.0f;
                    float y = acc[i][j][e] + bias_pre[j];
                    if (res) y += v;
                    if (relu) y = fmaxf(y, 0.0f);
                    if (mask) y = v > 0.0f ? y : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rc, lo, (unsigned)(mfma_row_of(e) * ldc * 4), 0);
                }
            }
        }
        return;
    }
    const auto rr = conv_tile_rsrc(res ? res : C, m0, M, ldc, BM);
    const auto rm = conv_tile_rsrc(mask ? mask : C, m0, M, ldc, BM);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ncol = (int)n0 + col0 + j * 32 + l31;
            const float bias_v = bias_pre[j];
            const unsigned lo = conv_lane_off(ncol, N, row0 + i * 32 + 4 * half, ldc);
            float rv[16], mv[16];
            if (res) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0));
            }
            if (mask) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    mv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float y = acc[i][j][e] + bias_v;
                if (res) y += rv[e];
                if (relu) y = fmaxf(y, 0.0f);
                if (mask) y = mv[e] > 0.0f ? y : 0.0f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rc, lo, (unsigned)(((e & 3) + 8 * (e >> 2)) * ldc * 4), 0);
            }
        }
    }
}

// Convolution epilogue through the LDS (round 5): the C layout of the 32x32 MFMA gives a lane ONE column and 16 scattered rows -- 4-byte stores,
// a wave instruction covers 2 rows x 128 B.  Here every wave parks its TM x TN tiles in a wave-private LDS region as a row-major (32 TM) x
// (32 TN) image and reads it back as float4 along the rows: residual loads and stores are 16 B per lane, a wave instruction covers 64 / (8 TN)
// rows x 128 TN B -- a quarter of the memory instructions, twice the contiguous run.  The residual is requested before the transpose.
// wl: this wave's (32 TM) x (32 TN + 4) floats (the +4 keeps rows 16-B aligned and the transposed reads conflict-light); needs N % 4 == 0,
// ldc % 4 == 0 and 16-B aligned C / res (the caller checks); same arithmetic per element as conv_epilogue_buffers: y = act(acc + bias (+ res)).
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_lds(const f32x16 (&acc)[TM][TN], float* __restrict__ wl, float* __restrict__ C, const float* __restrict__ res,
                                                  const float* __restrict__ bias, int relu, int64_t m0, int64_t M, int64_t n0, int64_t N, int64_t ldc, int BM,
                                                  int row0, int col0, int lane) {
    constexpr int LDW = 32 * TN + 4, LPR = 8 * TN, RPI = 64 / LPR, NIT = 32 * TM / RPI;       // lanes per row, rows per instruction, instructions
    const int l31 = lane & 31, half = lane >> 5;
    const int r4 = lane / LPR, c4 = lane % LPR;
    const auto rc = conv_tile_rsrc(C, m0, M, ldc, BM);
    const auto rr = conv_tile_rsrc(res ? res : C, m0, M, ldc, BM);
    const int64_t col = n0 + col0 + 4 * c4;
    const unsigned lo = col < N ? (unsigned)(((row0 + r4) * ldc + col) * 4) : 0x80000000u;
    float4 rv[NIT];
    if (res) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            rv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, lo, (unsigned)(it * RPI * ldc * 4), 0));
    }
    const float4 b4 = (bias && col < N) ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) wl[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * LDW + j * 32 + l31] = acc[i][j][e];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float4 v = *reinterpret_cast<const float4*>(wl + (it * RPI + r4) * LDW + 4 * c4);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        if (res) { v.x += rv[it].x; v.y += rv[it].y; v.z += rv[it].z; v.w += rv[it].w; }
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rc, lo, (unsigned)(it * RPI * ldc * 4), 0);
    }
}
template <int TM, int TN>
constexpr int epilogue_lds_floats() { return 4 * (32 * TM) * (32 * TN + 4); }
#ifndef ISX_EPI_LDS
#define ISX_EPI_LDS 0        // A/B (round 5): float4 epilogue through an LDS transpose: bit-identical, NO gain (1x1 lab 10.96 vs 10.91 ms) -- store width is not what limits the short-K layers
#endif

// ---- two-level accumulation of the trunk convolutions (round 5) -----------------------------------------------------------------
// A convolution output is NOT one fp32 chain over the whole reduction any more: the flattened reduction (kh, kw, ci) is cut into chunks of
// ISX_CONV_CHUNK terms; inside a chunk the k-ordered fma chain of the matrix core starts from +0, and the chunk sums are added, in order, into
// a second accumulator:   tot = 0;  for each chunk c: tot = tot + chain_c;   y = act(tot + bias (+ residual)).
// Why: a chain over K terms drifts from the exact sum like eps . K, chunks of c terms like eps . sqrt(K c + K^2 / c); measured end to end
// (scratch/chunk_study.py, ResNet-50 / ResNet-152 descriptors against a float64 evaluation): one chain is 1.2x further from float64 than torch's
// CPU fp32 path, chunks of 64 land at 0.7x.  K <= 64 (one chunk): tot = 0 + chain, the bits of rounds 1-4.  Restated by oracle/isx_oracle.c.
#ifdef ISX_CONV_CHUNK_AB
constexpr int kConvChunk = ISX_CONV_CHUNK_AB;      // A/B builds only (tools/build_variant.sh): the oracle follows ISX_CONV_CHUNK
#else
constexpr int kConvChunk = ISX_CONV_CHUNK;
#endif
#ifndef ISX_WG_PER_CU_128
#define ISX_WG_PER_CU_128 2     // resident workgroups per CU of the 128x128 convolution tiles: two accumulator sets = 128 VGPRs + ~55 -> two waves per SIMD
#endif
static_assert(kConvChunk % 32 == 0, "a chunk is a whole number of k-tiles");

// tot += acc (the chain of the chunk that just ended; acc itself is overwritten by the first MFMAs of the next chunk, which take C = 0)
template <int TM, int TN>
__device__ __forceinline__ void add_chunk(f32x16 (&tot)[TM][TN], const f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) tot[i][j] = tot[i][j] + acc[i][j];
}

// a += t on the 16 registers of one MFMA tile IN PLACE (tied asm operands: the sum stays in a's registers, hipcc cannot rename it into fresh ones)
__device__ __forceinline__ void add_tile_inplace(f32x16& a, const f32x16& t) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        f32x2_t x = {a[2 * p], a[2 * p + 1]};
        const f32x2_t y = {t[2 * p], t[2 * p + 1]};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        a[2 * p] = x[0];
        a[2 * p + 1] = x[1];
    }
}

// MFMAs of one staged k-tile.  Small tiles (TM * TN <= 2) have too few MFMAs per k-step to hide an LDS round trip
// behind: the operand reads of HALF a k-tile are issued back to back, then the MFMAs (counted lgkmcnt(n) waits instead
// of a drain per step; +4 % on the trunk's 1x1 convolutions).  a_base / b_base: this lane's first operand element.
// ZERO_C: the first k-step starts new chains (C = 0 as an inline constant; the old contents of acc are dead): chunk start of the two-level sum.
template <int TM, int TN, int BK, int LDA, int LDB, bool ZERO_C = false>
__device__ __forceinline__ void mfma_ktile(const float* __restrict__ a_base, const float* __restrict__ b_base, f32x16 (&acc)[TM][TN], f32x16 (*tot)[TN] = nullptr) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TM * TN <= 2 && !ISX_SIMPLE_KLOOP) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float af[BK / 4][TM], bf[BK / 4][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = a_base[(2 * (h * (BK / 4) + kk)) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = b_base[(2 * (h * (BK / 4) + kk)) * LDB + 32 * j];
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (ZERO_C && h == 0 && kk == 0 && tot) add_tile_inplace(tot[i][j], acc[i][j]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], (ZERO_C && h == 0 && kk == 0) ? zero : acc[i][j], 0, 0, 0);
                    }
        }
    } else if (ISX_KLOOP_PREFETCH) {
        // register prefetch one k-step ahead: the operand reads of step kk + 1 are issued BEFORE the MFMAs of step kk, so a wave
        // that finds itself alone on its SIMD (siblings parked at a barrier) does not expose an LDS round trip per step
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = a_base[32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = b_base[32 * j];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = a_base[(2 * kk + 2) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = b_base[(2 * kk + 2) * LDB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);             // keep the reads above the MFMAs
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ZERO_C && kk == 0 && tot) add_tile_inplace(tot[i][j], acc[i][j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], (ZERO_C && kk == 0) ? zero : acc[i][j], 0, 0, 0);
                }
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_base[(2 * kk) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = b_base[(2 * kk) * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ZERO_C && kk == 0 && tot) add_tile_inplace(tot[i][j], acc[i][j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], (ZERO_C && kk == 0) ? zero : acc[i][j], 0, 0, 0);
                }
        }
    }
}

// Operand addresses of the k-steps of a staged k-tile, PINNED in registers.  The K-major images put k-step kk at 2 kk LDA floats = more than the
// 1020 B a ds_read2_b32 offset reaches, so every k-step needs a base register of its own.  In the plain loops hipcc computes these 2 x BK / 2
// addresses once; with the chunk fold in the loop it re-derived them with a v_add_u32 in front of every ds_read2 instead (15 VALU instructions
// among the 32 MFMAs of a k-tile; the two-level kernels ran 10 % below the one-chain ones at ANY chunk length).  The empty asm makes each
// address an opaque 32-bit LDS pointer that cannot be rematerialised.
#ifndef ISX_TAIL_TM
#define ISX_TAIL_TM 1           // tail tiles of the 128x128 convolution launches: 64 ISX_TAIL_TM rows x 64 columns (A/B: 2)
#endif
constexpr int kTailLdsFloats = 32 * (64 * ISX_TAIL_TM + 64 + 2);      // BK = 32 stage of a tail tile (lds_pad(32) = 1 per operand)
#ifndef ISX_PIN_KTILE
#define ISX_PIN_KTILE 1
#endif
using lds_cfp = const __attribute__((address_space(3))) float*;
template <int BK>
struct KtilePtrs { lds_cfp a[BK / 2], b[BK / 2]; };
template <int BK, int LDA, int LDB>
__device__ __forceinline__ KtilePtrs<BK> pin_ktile_ptrs(const float* a_base, const float* b_base) {
    KtilePtrs<BK> p;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        p.a[kk] = (lds_cfp)(a_base + 2 * kk * LDA);
        p.b[kk] = (lds_cfp)(b_base + 2 * kk * LDB);
        asm volatile("" : "+v"(p.a[kk]));
        asm volatile("" : "+v"(p.b[kk]));
    }
    return p;
}
// the k loop of mfma_ktile's prefetch variant on pinned addresses
// tot (ZERO_C only, ISX_FOLD_INTERLEAVE): the chain that ended with the previous k-tile is added to tot tile by tile right in front of the MFMA that
// restarts that tile from C = 0 -- the adds of tile n + 1 issue while the MFMA of tile n runs.  (The MFMAs that produced acc are at least one
// barrier old: no MFMA -> VALU wait states needed.)
template <int TM, int TN, int BK, bool ZERO_C = false>
__device__ __forceinline__ void mfma_ktile_pinned(const KtilePtrs<BK>& p, f32x16 (&acc)[TM][TN], f32x16 (*tot)[TN] = nullptr) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = p.a[0][32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = p.b[0][32 * j];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
        if (kk + 1 < BK / 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = p.a[kk + 1][32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = p.b[kk + 1][32 * j];
        }
        __builtin_amdgcn_sched_barrier(0);             // keep the reads above the MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (ZERO_C && kk == 0 && tot) add_tile_inplace(tot[i][j], acc[i][j]);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], (ZERO_C && kk == 0) ? zero : acc[i][j], 0, 0, 0);
            }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_tiles(f32x16 (&t)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t[i][j][e] = 0.0f;
}

// Loop shape of the two-level sum (CHUNK terms per chain; 0 = one chain): the k loop is NESTED -- an outer loop over the chunks, the inner loop
// over the CHUNK / BK k-tiles of a chunk is the plain staged loop of rounds 1-4 (loads of the next k-tile, MFMAs, barrier, LDS stores, barrier),
// and fold_chunk runs between two inner loops:  tot = tot + acc;  acc = 0  (after the last chunk too: the value is tot).
// Why nested (round 5 A/B on the 3x3 / dual families, ms per lab pass; one chain: 9.89 / 6.61): a conditional fold INSIDE the k loop cost 10-15 %
// at any chunk length -- behind the MFMAs it needs 16 wait states and keeps hipcc from hoisting the loop's first barrier in between the last
// MFMAs, the operand addresses were re-derived with a v_add_u32 per ds_read2 (11.13 / 7.56; addresses pinned: 10.85 / 7.29); in front of the
// MFMAs hipcc duplicates the loop body and serialises the staging loads (11.37 / 7.61); C = 0 in the chunk's first MFMAs instead of zeroing
// (two copies of the k-tile body) was slower still.
template <int TM, int TN>
__device__ __forceinline__ void fold_chunk(f32x16 (&tot)[TM][TN], f32x16 (&acc)[TM][TN]) {
    add_chunk<TM, TN>(tot, acc);
    zero_tiles(acc);
}
#ifndef ISX_FOLD_INTERLEAVE
#define ISX_FOLD_INTERLEAVE 1
#endif
// returns true when the k-tile added the previous chunk's chain to tot itself (pinned 128x128 shape with ISX_FOLD_INTERLEAVE)
template <int TM, int TN, int BK, int LDA, int LDB, bool PINNED, bool ZERO_C = false>
__device__ __forceinline__ void mfma_ktile_sel(const float* __restrict__ a_base, const float* __restrict__ b_base, const KtilePtrs<BK>& pins, f32x16 (&acc)[TM][TN],
                                               f32x16 (*tot)[TN] = nullptr) {
    if constexpr (PINNED) mfma_ktile_pinned<TM, TN, BK, ZERO_C>(pins, acc, ISX_FOLD_INTERLEAVE ? tot : nullptr);
    else mfma_ktile<TM, TN, BK, LDA, LDB, ZERO_C>(a_base, b_base, acc, ISX_FOLD_INTERLEAVE >= 2 ? tot : nullptr);
}
}  // namespace isx
