// Lab: fp16 MFMA GEMM for the filter pass with ONE wave per SIMD (C = Qh . Gh^T, fp32 accumulate).
//   256x256 block tile, 256 threads = 4 waves, each wave a 128x128 quarter = 4 x 4 tiles of v_mfma_f32_32x32x16_f16 (256 accumulator
//   registers per lane: the unified 512-register file of a one-wave-per-SIMD kernel).  Against the shipped ping-pong kernel (8 waves, a
//   wave = 128 x 64): a k-tile costs 128 KB of LDS operand reads instead of 192 KB, and no wave waits for a partner's phase.
//   Operands arrive by LDS-DMA in k-slices of 32 halves (512 rows x 64 B = 32 KB) through a ring of four slices, three slices ahead;
//   one workgroup barrier per slice.  The wave software-pipelines its own operand reads: the fragments of the next 16-k step are read
//   while the 16 MFMAs of the current one execute.
// build: hipcc -O3 --offload-arch=gfx950 -o scratch/lab/f16_w4_lab scratch/lab/f16_w4_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,$PWD/instance-search_amd/csrc
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <cstring>

extern "C" int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, void* stream);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256;
constexpr int SBK = 32;                       // halves per k-slice: 64 B per row = 4 chunks of 16 B
constexpr int NST = 4;                        // slices in the ring
constexpr int ST_B = (BM + BN) * SBK * 2;     // 32 KB per slice

__device__ __forceinline__ auto uniform_rsrc16(const void* base, int64_t nbytes) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)base);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
    const unsigned nb = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nbytes < 0xFFFFFFFFll ? (nbytes > 0 ? nbytes : 0) : 0xFFFFFFFFll));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | (uintptr_t)lo), 0, (int)nb, 0x00020000);
}

template <int VAR>
__global__ __launch_bounds__(256) void gemm_w4(const _Float16* __restrict__ Q, int64_t M, const _Float16* __restrict__ G, int64_t N, int D,
                                               float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n, unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(1024))) char lds[NST * ST_B];
    auto stamp = [&](int t, int i) { if (VAR == 2 && blockIdx.x == 300 && threadIdx.x == 0 && t < 64) stamps[t * 4 + i] = __builtin_amdgcn_s_memtime(); };
    const int nwg = tiles_m * tiles_n;
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
    constexpr int GN = 8;
    const int per_group = GN * tiles_m, gid = wg / per_group, first_n = gid * GN;
    const int gsz = min(GN, tiles_n - first_n), within = wg - gid * per_group;
    const int64_t m0 = (int64_t)(within / gsz) * BM, n0 = (int64_t)(first_n + within % gsz) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // DMA: instruction i of wave w fills image rows (8 w + i) * 16 + lane / 4 of a slice (rows 0..255 = A, 256..511 = B: waves 0, 1 move A,
    // waves 2, 3 move B); slot lane % 4 of a row holds chunk slot ^ swz(row), swz(row) = (row >> 2) & 3 (applied to the SOURCE address)
    const bool isB = wave >= 2;
    const int64_t r0 = isB ? n0 : m0, rows = isB ? N : M;
    constexpr bool SLICED = (VAR == 3);           // operand images stored slice-major: [k / 32][row][32 halves] -- a slice of 256 rows is ONE contiguous 16-KB block
    const auto rs = SLICED ? uniform_rsrc16((isB ? G : Q) + r0 * SBK, (int64_t)rows * D * 2 - r0 * SBK * 2)
                           : uniform_rsrc16((isB ? G : Q) + r0 * D, ((rows - r0) < 256 ? (rows - r0) : 256) * (int64_t)D * 2);
    const unsigned slice_stride = SLICED ? (unsigned)(rows * SBK * 2) : (unsigned)(SBK * 2);
    unsigned gvo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = ((wave & 1) * 8 + i) * 16 + (lane >> 2);                  // row inside the operand's 256
        gvo[i] = (unsigned)r * (SLICED ? (unsigned)(SBK * 2) : (unsigned)D * 2u) + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) << 4);
    }
    const int T = D / SBK;                                                       // D % 32 == 0
    auto issue = [&](int t, int slot_of) {                                       // k-slice t -> ring slot slot_of % NST
        char* dst = lds + (slot_of & (NST - 1)) * ST_B + wave * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, gvo[i], (unsigned)t * slice_stride, 0, 0);
    };
    // operand fragments of the 32x32x16 MFMA: lane l holds row (l & 31), chunk 2 ks + (l >> 5) of its row
    const int sw = (l31 >> 2) & 3;
    int a_ad[2], b_ad[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_ad[ks] = (wm * 128 + l31) * 64 + (((2 * ks + half) ^ sw) << 4);
        b_ad[ks] = (256 + wn * 128 + l31) * 64 + (((2 * ks + half) ^ sw) << 4);
    }
    half8 fa[2][4], fb[2][4];
    auto read_frags = [&](int t, int ks, int set) {
        const char* buf = lds + (t & (NST - 1)) * ST_B;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const half8*>(buf + a_ad[ks] + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = *reinterpret_cast<const half8*>(buf + b_ad[ks] + j * 2048);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    // prologue: four slices in flight, slice 0 landed and visible, its first fragments read
#pragma unroll
    for (int t = 0; t < NST; ++t)
        issue(t < T ? t : T - 1, t);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);

    for (int t = 0; t < T; ++t) {
        // first half: MFMAs of (t, ks = 0) beside the reads of (t, ks = 1)
        stamp(t, 0);
        read_frags(t, 1, 1);
        if (VAR >= 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);           // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // 1 DS read
            }
        }
        mfmas(0);
        // slice t + 1 landed (in-order counter: the slices behind it may still be in flight); every wave is past its reads of slice t
        __builtin_amdgcn_sched_barrier(0);                                       // the halves of the trip stay apart: hipcc would move MFMAs across the barrier
        stamp(t, 1);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                        // two younger slices stay in flight (the issue below is unconditional)
        stamp(t, 2);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                      // lgkmcnt(0) as an instruction the wait-count pass SEES: behind an asm
                                                                                 // statement it keeps the reads of set 1 pending and makes mfmas(1) wait for set 0's
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp(t, 3);
        // refills the slot slice t occupied.  Unconditional (one basic block: the scheduler can spread the DMAs and the reads over the MFMAs):
        // past the end the last slice is fetched again into a slot nobody reads, and the fragments of a slice that does not exist are never used
        issue(t + NST < T ? t + NST : T - 1, t + NST);
        read_frags(t + 1, 0, 0);
        if (VAR >= 1) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);           // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);           // VMEM read (the DMA)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);           // DS read
            }
        }
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // no DMA may outlive the workgroup's LDS allocation
    // plain store: 32x32 C layout, col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn * 128 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (m < M && n < N) C[m * ldc + n] = acc[i][j][e];
            }
        }
}

static unsigned long long* g_stamps = nullptr;
template <int VAR>
static void launch_w4(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C) {
    const int tm = (int)((M + BM - 1) / BM), tn = (int)((N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_w4<VAR>), dim3((unsigned)(tm * tn)), dim3(256), 0, 0, Q, M, G, N, D, C, N, tm, tn, g_stamps);
}

template <class F>
static float time_ms(F f, int it = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < it; ++i) f();
        hipEventRecord(b); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        best = fminf(best, ms / it);
    }
    return best;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 10240, N = argc > 2 ? atoll(argv[2]) : 16384;
    const int D = argc > 3 ? atoi(argv[3]) : 2048;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    std::vector<_Float16> hq((size_t)M * D), hg((size_t)N * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = (_Float16)(rnd() * 16000.0f);
    for (auto& v : hg) v = (_Float16)(rnd() * 16000.0f);
    _Float16 *dq, *dg; float *c0, *c1;
    CK(hipMalloc(&dq, hq.size() * 2)); CK(hipMalloc(&dg, hg.size() * 2));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 2, hipMemcpyHostToDevice));
    // slice-major copies: [k / 32][row][k % 32]
    std::vector<_Float16> sq(hq.size()), sg(hg.size());
    for (int64_t r = 0; r < M; ++r) for (int k = 0; k < D; ++k) sq[((size_t)(k / SBK) * M + r) * SBK + k % SBK] = hq[(size_t)r * D + k];
    for (int64_t r = 0; r < N; ++r) for (int k = 0; k < D; ++k) sg[((size_t)(k / SBK) * N + r) * SBK + k % SBK] = hg[(size_t)r * D + k];
    _Float16 *dqs, *dgs;
    CK(hipMalloc(&dqs, sq.size() * 2)); CK(hipMalloc(&dgs, sg.size() * 2));
    CK(hipMemcpy(dqs, sq.data(), sq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgs, sg.data(), sg.size() * 2, hipMemcpyHostToDevice));
    const double fl = 2.0 * M * N * D;
    for (int rep = 0; rep < reps; ++rep) {
        float t = time_ms([&] { isx_cosine_sim_f16(dq, M, dg, N, D, c0, nullptr); });
        printf("shipped ping-pong (8 waves)  : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_w4<0>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("one wave per SIMD            : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_w4<1>(dq, M, dg, N, D, c1); });
        CK(hipGetLastError());
        printf("one wave per SIMD, sched grps: %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        t = time_ms([&] { launch_w4<3>(dqs, M, dgs, N, D, c1); });
        CK(hipGetLastError());
        printf("  + slice-major operands     : %.3f ms  %.0f TF\n", t, fl / t * 1e-9);
        fflush(stdout);
    }
    {   // where a trip of the k loop spends its time (workgroup 300, wave 0): shader-clock stamps
        CK(hipMalloc(&g_stamps, 64 * 4 * 8)); CK(hipMemset(g_stamps, 0, 64 * 4 * 8));
        launch_w4<2>(dq, M, dg, N, D, c1); CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(256);
        CK(hipMemcpy(st.data(), g_stamps, 2048, hipMemcpyDeviceToHost));
        printf("trip: first half (16 MFMA + reads) | vmcnt wait | lgkm + barrier | second half  [memtime ticks, 100 MHz]\n");
        for (int t = 8; t < 24; ++t)
            printf("  t=%2d  %5llu %5llu %5llu %5llu\n", t, st[t * 4 + 1] - st[t * 4], st[t * 4 + 2] - st[t * 4 + 1], st[t * 4 + 3] - st[t * 4 + 2], st[(t + 1) * 4] - st[t * 4 + 3]);
        printf("  trips 8..40: %.1f ticks per trip\n", (double)(st[40 * 4] - st[8 * 4]) / 32);
    }
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
    for (int var = 0; var < 3; ++var) {
        CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
        if (var == 0) launch_w4<0>(dq, M, dg, N, D, c1); else if (var == 1) launch_w4<1>(dq, M, dg, N, D, c1); else launch_w4<3>(dqs, M, dgs, N, D, c1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < h0.size(); ++i) diff += (memcmp(&h0[i], &h1[i], 4) != 0);
        printf("variant %d vs shipped kernel: %zu of %zu scores differ in bits\n", var, diff, h0.size());
    }
    return 0;
}
