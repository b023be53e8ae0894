// Lab: fp16 MFMA GEMM for the filter pass, "ping-pong" schedule (C = Qh . Gh^T, fp32 accumulate).
//   256x256 block tile, BK = 64, 512 threads = 8 waves.  Every 128x128 quadrant of the block tile is split 2 (M) x 4 (N)
//   over the waves: a wave owns a 64x32 piece of each quadrant (4 x 2 tiles of v_mfma_f32_16x16x32_f16).  One k-tile =
//   4 phases (one quadrant each); a phase = [LDS operand reads + LDS-DMA prefetch issue] barrier [16 MFMAs] barrier.
//   Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while its partner
//   (wave w + 4, same SIMD) is in its load segment.  Operand tiles arrive by LDS-DMA (global_load_lds_dwordx4) as 16-KB
//   half-tiles (128 rows x 64 k), DEPTH half-tiles ahead, retired by a counted vmcnt once per k-tile.
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/f16_pp_lab scratch/lab/f16_pp_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,$PWD/instance-search_amd/csrc
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

extern "C" int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, void* stream);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16w __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT_B = 128 * BK * 2;          // one half-tile: 128 rows x 128 B = 16 KB
constexpr int BUF_B = 4 * HT_B;             // [A0][A1][B0][B1] = 64 KB per k-tile, two k-tiles resident
// half-tile kinds in issue / consumption order within a k-tile: A0 (phase 0), B0 (phase 0), B1 (phase 1), A1 (phase 2)
__device__ __forceinline__ constexpr int ht_slot(int j) { return j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 3 : 1; }

#define LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int DEPTH, bool WB>
__global__ __launch_bounds__(512) void gemm_pp(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                               float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][bb][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS-DMA sources: half-tile kind j (issue order), instruction i: image rows wave*16 + i*8 + lane/8, slot lane%8 holds chunk slot ^ swz(row)
    const _Float16* gsrc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = ht_slot(j);
        const bool isb = slot >= 2;
        const int half_ = slot & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + (lane >> 3);
            int64_t gr = (isb ? n0 : m0) + half_ * 128 + r;
            const int64_t lim = isb ? N : M;
            gr = gr < lim ? gr : lim - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            gsrc[j][i] = (isb ? G : Q) + gr * D + c * 8;
        }
    }
    const int T = D / BK;                                   // D % 64 == 0 required
    auto issue = [&](int j, int t) {                        // half-tile j of k-tile t -> buffer t & 1
        char* dst = lds + (t & 1) * BUF_B + ht_slot(j) * HT_B + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j][i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    // operand fragment addresses (bytes inside a buffer, without the half-tile base): row (l & 15), chunk (4 s + l / 16) ^ swz
    const int x0 = (lane >> 4) ^ ((lane >> 1) & 7);
    const int lrow = (lane & 15) * 128;
    int a_ad[2], b_ad[2];
    a_ad[0] = wm * 64 * 128 + lrow + (x0 << 4);
    a_ad[1] = wm * 64 * 128 + lrow + ((x0 ^ 4) << 4);
    b_ad[0] = 2 * HT_B + wn * 32 * 128 + lrow + (x0 << 4);
    b_ad[1] = 2 * HT_B + wn * 32 * 128 + lrow + ((x0 ^ 4) << 4);

    half8 af[4][2], bf[2][2][2];                              // A piece of the running quadrant row; B pieces of both quadrant columns
    auto read_a = [&](const char* buf, int ah) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const half8*>(buf + ah * HT_B + i * 2048 + a_ad[s]);
    };
    auto read_b = [&](const char* buf, int bh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[bh][j][s] = *reinterpret_cast<const half8*>(buf + bh * HT_B + j * 2048 + b_ad[s]);
    };
    auto mfmas = [&](int ah, int bh) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ah][bh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][s], bf[bh][j][s], acc[ah][bh][i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: half-tiles 0 .. DEPTH-1 in flight, k-tile 0 landed
#pragma unroll
    for (int h = 0; h < DEPTH; ++h)
        if (h / 4 < T) issue(h % 4, h / 4);
    if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 4) * 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();              // waves 4-7 run one barrier behind

    for (int t = 0; t < T; ++t) {
        const char* buf = lds + (t & 1) * BUF_B;
        // ---- phase 0: quadrant (0, 0)
        read_a(buf, 0);
        read_b(buf, 0);
        { const int h = 4 * t + 0 + DEPTH; if (h / 4 < T) issue(h % 4, h / 4); }
        if (WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (!WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: quadrant (0, 1)
        read_b(buf, 1);
        { const int h = 4 * t + 1 + DEPTH; if (h / 4 < T) issue(h % 4, h / 4); }
        if (WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (!WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (1, 1)
        read_a(buf, 1);
        { const int h = 4 * t + 2 + DEPTH; if (h / 4 < T) issue(h % 4, h / 4); }
        if (WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (!WB) LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (1, 0); retire k-tile t + 1 before the barrier that precedes its first read
        { const int h = 4 * t + 3 + DEPTH; if (h / 4 < T) issue(h % 4, h / 4); }
        if (4 * t + 3 + DEPTH < 4 * T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 4) * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();              // pair the extra barrier of waves 4-7

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t n = n0 + bh * 128 + wn * 32 + j * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t m = m0 + ah * 128 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                        if (m < M && n < N) C[m * ldc + n] = acc[ah][bh][i][j][r];
                    }
                }
}


// ---- two phases per k-tile: phase A = quadrants (0,0) + (0,1) [reads A0, B0, B1], phase B = quadrants (1,1) + (1,0) [reads A1];
// 32 MFMAs per segment, operand reads retired BEFORE the barrier (the load segment has the slack)
template <int MF>
__global__ __launch_bounds__(512) void gemm_pp2(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                                float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][bb][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16* gsrc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = ht_slot(j);
        const bool isb = slot >= 2;
        const int half_ = slot & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + (lane >> 3);
            int64_t gr = (isb ? n0 : m0) + half_ * 128 + r;
            const int64_t lim = isb ? N : M;
            gr = gr < lim ? gr : lim - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            gsrc[j][i] = (isb ? G : Q) + gr * D + c * 8;
        }
    }
    const int T = D / BK;
    auto issue = [&](int j, int t) {
        char* dst = lds + (t & 1) * BUF_B + ht_slot(j) * HT_B + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j][i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    const int x0 = (lane >> 4) ^ ((lane >> 1) & 7);
    const int lrow = (lane & 15) * 128;
    int a_ad[2], b_ad[2];
    a_ad[0] = wm * 64 * 128 + lrow + (x0 << 4);
    a_ad[1] = wm * 64 * 128 + lrow + ((x0 ^ 4) << 4);
    b_ad[0] = 2 * HT_B + wn * 32 * 128 + lrow + (x0 << 4);
    b_ad[1] = 2 * HT_B + wn * 32 * 128 + lrow + ((x0 ^ 4) << 4);
    half8 af[4][2], bf[2][2][2];
    auto read_a = [&](const char* buf, int ah) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const half8*>(buf + ah * HT_B + i * 2048 + a_ad[s]);
    };
    auto read_b = [&](const char* buf, int bh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[bh][j][s] = *reinterpret_cast<const half8*>(buf + bh * HT_B + j * 2048 + b_ad[s]);
    };
    auto mfmas2 = [&](int ah) {
        __builtin_amdgcn_s_setprio(1);
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
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h / 4 < T) issue(h % 4, h / 4);
    if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; ++t) {
        const char* buf = lds + (t & 1) * BUF_B;
        // ---- phase A
        read_a(buf, 0);
        read_b(buf, 0);
        read_b(buf, 1);
        if (t + 1 < T) { issue(2, t + 1); issue(3, t + 1); }
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B
        read_a(buf, 1);
        if (t + 2 < T) { issue(0, t + 2); issue(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t n = n0 + bh * 128 + wn * 32 + j * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t m = m0 + ah * 128 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                        if (m < M && n < N) C[m * ldc + n] = acc[ah][bh][i][j][r];
                    }
                }
}

template <int MF>
__global__ __launch_bounds__(512) void gemm_pp2w(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                                float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // a wave's 64 x 32 piece of a quadrant as TWO v_mfma_f32_32x32x16_f16 tiles (half the operand-register reads per FLOP of the 16x16x32 form)
    f32x16w acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][bb][i][e] = 0.0f;
    const _Float16* gsrc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = ht_slot(j);
        const bool isb = slot >= 2;
        const int half_ = slot & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + (lane >> 3);
            int64_t gr = (isb ? n0 : m0) + half_ * 128 + r;
            const int64_t lim = isb ? N : M;
            gr = gr < lim ? gr : lim - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            gsrc[j][i] = (isb ? G : Q) + gr * D + c * 8;
        }
    }
    const int T = D / BK;
    auto issue = [&](int j, int t) {
        char* dst = lds + (t & 1) * BUF_B + ht_slot(j) * HT_B + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j][i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    // 32x32x16 fragments: lane l holds row (l & 31), chunk 2 s + (l >> 5) of its row, s = 0..3 over the 64-k tile
    const int l31 = lane & 31, hsw = (l31 >> 1) & 7;
    int a_ad[4], b_ad[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = ((2 * s + (lane >> 5)) ^ hsw) << 4;
        a_ad[s] = wm * 64 * 128 + l31 * 128 + c;
        b_ad[s] = 2 * HT_B + wn * 32 * 128 + l31 * 128 + c;
    }
    half8 af[2][4], bf[2][4];
    auto read_a = [&](const char* buf, int ah) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) af[i][s] = *reinterpret_cast<const half8*>(buf + ah * HT_B + i * 4096 + a_ad[s]);
    };
    auto read_b = [&](const char* buf, int bh) {
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[bh][s] = *reinterpret_cast<const half8*>(buf + bh * HT_B + b_ad[s]);
    };
    auto mfmas2 = [&](int ah) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[ah][bh][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][s], bf[bh][s], acc[ah][bh][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll
    for (int h = 0; h < 6; ++h)
        if (h / 4 < T) issue(h % 4, h / 4);
    if (T > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; ++t) {
        const char* buf = lds + (t & 1) * BUF_B;
        // ---- phase A
        read_a(buf, 0);
        read_b(buf, 0);
        read_b(buf, 1);
        if (t + 1 < T) { issue(2, t + 1); issue(3, t + 1); }
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B
        read_a(buf, 1);
        if (t + 2 < T) { issue(0, t + 2); issue(1, t + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t n = n0 + bh * 128 + wn * 32 + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = m0 + ah * 128 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m < M && n < N) C[m * ldc + n] = acc[ah][bh][i][e];
                }
            }
}

template <int MF>
__global__ __launch_bounds__(512) void gemm_pp3(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                                float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][bb][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16* gsrc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int slot = ht_slot(j);
        const bool isb = slot >= 2;
        const int half_ = slot & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + (lane >> 3);
            int64_t gr = (isb ? n0 : m0) + half_ * 128 + r;
            const int64_t lim = isb ? N : M;
            gr = gr < lim ? gr : lim - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            gsrc[j][i] = (isb ? G : Q) + gr * D + c * 8;
        }
    }
    const int T = D / BK;
    auto issue = [&](int j, int t) {
        char* dst = lds + (t & 1) * BUF_B + ht_slot(j) * HT_B + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j][i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    const int x0 = (lane >> 4) ^ ((lane >> 1) & 7);
    const int lrow = (lane & 15) * 128;
    int a_ad[2], b_ad[2];
    a_ad[0] = wm * 64 * 128 + lrow + (x0 << 4);
    a_ad[1] = wm * 64 * 128 + lrow + ((x0 ^ 4) << 4);
    b_ad[0] = 2 * HT_B + wn * 32 * 128 + lrow + (x0 << 4);
    b_ad[1] = 2 * HT_B + wn * 32 * 128 + lrow + ((x0 ^ 4) << 4);
    half8 af[4][2], bf[2][2][2];
    auto read_a = [&](const char* buf, int ah) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[i][s] = *reinterpret_cast<const half8*>(buf + ah * HT_B + i * 2048 + a_ad[s]);
    };
    auto read_b = [&](const char* buf, int bh) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[bh][j][s] = *reinterpret_cast<const half8*>(buf + bh * HT_B + j * 2048 + b_ad[s]);
    };
    // MFMA segment with the DMA issue of this wave riding in its own instruction stream (phase A: A1 of k-tile t + 1; phase B:
    // A0, B0, B1 of k-tile t + 2)
    auto mfmas2 = [&](int ah, int t) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ah][bh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][s], bf[bh][j][s], acc[ah][bh][i][j], 0, 0, 0);
                if (ah == 0) {
                    if (s == 0 && bh == 0 && t + 1 < T) issue(3, t + 1);
                } else if (t + 2 < T) {
                    if (s == 0 && bh == 0) issue(0, t + 2);
                    if (s == 0 && bh == 1) issue(1, t + 2);
                    if (s == 1 && bh == 0) issue(2, t + 2);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    };
    issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
    if (T > 1) { issue(0, 1); issue(1, 1); issue(2, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < T; ++t) {
        const char* buf = lds + (t & 1) * BUF_B;
        // ---- phase A
        read_a(buf, 0);
        read_b(buf, 0);
        read_b(buf, 1);
        // retire A1 of this k-tile (issued in MFMA segment A of the previous one); younger: A0 / B0 / B1 of k-tile t + 1
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(0, t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase B
        read_a(buf, 1);
        // retire A0 / B0 / B1 of k-tile t + 1 (issued in MFMA segment B of the previous k-tile); younger: A1 of k-tile t + 1
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LGKMCNT0();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfmas2(1, t);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int64_t n = n0 + bh * 128 + wn * 32 + j * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t m = m0 + ah * 128 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                        if (m < M && n < N) C[m * ldc + n] = acc[ah][bh][i][j][r];
                    }
                }
}

static void launch_pp3(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_B)); once = true; }
    hipLaunchKernelGGL(gemm_pp3<0>, dim3(tm * tn), dim3(512), 2 * BUF_B, 0, Q, M, G, N, D, C, N, tm, tn);
}

static void launch_pp2w(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp2w<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_B)); once = true; }
    hipLaunchKernelGGL(gemm_pp2w<0>, dim3(tm * tn), dim3(512), 2 * BUF_B, 0, Q, M, G, N, D, C, N, tm, tn);
}
static void launch_pp2(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_B)); once = true; }
    hipLaunchKernelGGL(gemm_pp2<0>, dim3(tm * tn), dim3(512), 2 * BUF_B, 0, Q, M, G, N, D, C, N, tm, tn);
}

template <int DEPTH, bool WB>
static void launch_pp(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pp<DEPTH, WB>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_B)); once = true; }
    hipLaunchKernelGGL((gemm_pp<DEPTH, WB>), dim3(tm * tn), dim3(512), 2 * BUF_B, 0, Q, M, G, N, D, C, N, tm, tn);
}

template <class F>
static float time_ms(F f, int it = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) f();
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        best = fminf(best, ms / it);
    }
    return best;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 10000, N = argc > 2 ? atoll(argv[2]) : 32768;
    const int D = argc > 3 ? atoi(argv[3]) : 2048;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    std::vector<_Float16> hq((size_t)M * D), hg((size_t)N * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    // as in the search: unit-norm-like rows scaled so that the largest element sits in [2^13, 2^14)
    for (auto& v : hq) v = (_Float16)(rnd() * 16000.0f);
    for (auto& v : hg) v = (_Float16)(rnd() * 16000.0f);
    _Float16 *dq, *dg; float *c0, *c1;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * D;
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    auto compare = [&](const char* name) {
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; double mx = 0, ref = 0;
        for (size_t i = 0; i < h0.size(); ++i) ref = fmax(ref, fabs((double)h0[i]));
        for (size_t i = 0; i < h0.size(); ++i) {
            const double d = fabs((double)h0[i] - h1[i]);
            if (d > 2e-6 * ref || h1[i] != h1[i]) ++bad;
            mx = fmax(mx, d);
        }
        printf("%s vs shipped kernel: %zu outside tolerance, max |diff| %.4g (max |ref| %.4g)\n", name, bad, mx, ref);
    };
    for (int rep = 0; rep < reps; ++rep) {
        float t = time_ms([&] { isx_cosine_sim_f16(dq, M, dg, N, D, c0, nullptr); });
        printf("shipped 256x256 reg-staged : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_pp<6, false>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("ping-pong 4 phases         : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_pp<6, true>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("ping-pong 4 ph, wait before: %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_pp2(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("ping-pong 2 phases         : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_pp2w(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("ping-pong 2 ph, 32x32x16   : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_pp3(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("ping-pong, DMA in MFMA segs: %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        fflush(stdout);
    }
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
    launch_pp<6, false>(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
    compare("ping-pong 4 phases");
    CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
    launch_pp<6, true>(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
    compare("ping-pong 4 ph, wait before");
    CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
    launch_pp2(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
    compare("ping-pong 2 phases");
    CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
    launch_pp2w(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
    compare("ping-pong 2 phases, 32x32x16");
    CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
    launch_pp3(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
    compare("ping-pong, DMA in MFMA segs");
    return 0;
}
