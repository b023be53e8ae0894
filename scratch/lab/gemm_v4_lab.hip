// Lab: PERSISTENT fp32-MFMA NT GEMM (v4).  Same k loop as the shipped cosine_gemm_kernel (K-major LDS image, ds_read_b32 fragments,
// one LDS stage, 99 % MFMA-busy inside the loop by in-kernel stamps); what changes is everything AROUND the loop, where the shipped
// kernel loses ~10 % on a K = 2048 GEMM and most of its time on the short-K 1x1 convolutions:
//   * grid = resident workgroups only; every workgroup walks tiles b, b + G, b + 2G, ... of the XCD-aware tile order;
//   * the first k-tile of the NEXT tile is fetched during the last k-tile of the current one: no cold prologue per tile;
//   * MFMA operand roles swapped (A <- W rows, B <- X rows): a lane owns 4 consecutive output columns of one row -> float4 epilogue
//     stores / residual loads / bias loads instead of 16 scalar accesses per 32x32 tile.
// Numerics unchanged: one ascending-k fma chain per output (bit-identical to libisx).
// build: hipcc -O3 --offload-arch=gfx950 -I instance-search_amd/csrc -o scratch/lab/gemm_v4_lab scratch/lab/gemm_v4_lab.hip -Linstance-search_amd/csrc -lisx -Wl,-rpath,'$ORIGIN/../../instance-search_amd/csrc'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <cmath>

#include "gemm_tile.hpp"

using namespace isx;

extern "C" int isx_conv1x1_nhwc(const float* x, int64_t M, int Cin, const float* w, int Cout, const float* bias, const float* residual, int relu,
                                float* y, void* stream);

__device__ __forceinline__ void tile_of_id(int tiles_m, int tiles_n, int v, int& tile_m, int& tile_n) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = v & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    const int per_group = GROUP_N * tiles_m;
    const int gid = wg / per_group;
    const int first_n = gid * GROUP_N;
    const int gsz = min(GROUP_N, tiles_n - first_n);
    const int within = wg - gid * per_group;
    tile_m = within / gsz;
    tile_n = first_n + within % gsz;
}

// EPI: 0 = store, 2 = convolution epilogue y = act(acc + bias[n] (+ res[m][n]))
template <int TM, int TN, int BK, int EPI, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_v4(const float* __restrict__ X, int64_t M, const float* __restrict__ W, int64_t N, int D,
                                                    float* __restrict__ C, int64_t ldc, int tiles_m, int tiles_n,
                                                    const float* __restrict__ bias, const float* __restrict__ res, int relu) {
    constexpr int BM = 64 * TM, BN = 64 * TN, LDA = BM + lds_pad(BK), LDB = BN + lds_pad(BK);
    __shared__ float lds[BK * (LDA + LDB)];
    float* As = lds;                // X tile, K-major
    float* Bs = lds + BK * LDA;     // W tile, K-major
    const int ntiles = tiles_m * tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const float* x_base = As + half * LDA + wm * (32 * TM) + l31;      // B operand (output rows)
    const float* w_base = Bs + half * LDB + wn * (32 * TN) + l31;      // A operand (output columns)
    const int nk = D / BK;                                             // lab: D % BK == 0, 16-B aligned rows

    int t = blockIdx.x;
    if (t >= ntiles) return;
    int tile_m, tile_n;
    tile_of_id(tiles_m, tiles_n, t, tile_m, tile_n);
    int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    float4 ra[BM * BK / 1024], rb[BN * BK / 1024];
    load_tile<true, BM, BK>(X, M, D, m0, 0, ra);
    load_tile<true, BN, BK>(W, N, D, n0, 0, rb);

    while (true) {
        f32x16 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        store_tile<BM, BK>(As, ra);
        store_tile<BN, BK>(Bs, rb);
        __syncthreads();
        const int t_next = t + gridDim.x;
        const bool have_next = t_next < ntiles;
        int64_t m0n = 0, n0n = 0;
        if (have_next) {
            int tmn, tnn;
            tile_of_id(tiles_m, tiles_n, t_next, tmn, tnn);
            m0n = (int64_t)tmn * BM; n0n = (int64_t)tnn * BN;
        }
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (more) {
                load_tile<true, BM, BK>(X, M, D, m0, (kt + 1) * BK, ra);
                load_tile<true, BN, BK>(W, N, D, n0, (kt + 1) * BK, rb);
            } else if (have_next) {                                    // first k-tile of the NEXT output tile, in flight across the epilogue
                load_tile<true, BM, BK>(X, M, D, m0n, 0, ra);
                load_tile<true, BN, BK>(W, N, D, n0n, 0, rb);
            }
            mfma_ktile<TN, TM, BK, LDB, LDA>(w_base, x_base, acc);
            __syncthreads();
            if (more) {
                store_tile<BM, BK>(As, ra);
                store_tile<BN, BK>(Bs, rb);
                __syncthreads();
            }
        }
        // D[i = n][j = m]: lane holds output row m = .. + l31, output columns n = .. + 8q + 4*half + (0..3) in acc[4q .. 4q+3]
        const bool vec = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (EPI != 2 || !res || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int64_t m = m0 + wm * (32 * TM) + 32 * j + l31;
            if (m < M) {
                float* crow = C + m * ldc;
                const float* rrow = (EPI == 2 && res) ? res + m * ldc : nullptr;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t n = n0 + wn * (32 * TN) + 32 * i + 8 * q + 4 * half;
                        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        if (vec && n + 3 < N) {
                            if (EPI == 2) {
                                const float4 bv = *reinterpret_cast<const float4*>(bias + n);
                                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                                if (rrow) {
                                    const float4 rv = *reinterpret_cast<const float4*>(rrow + n);
                                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                                }
                                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                            }
                            *reinterpret_cast<float4*>(crow + n) = v;
                        } else {
                            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (n + c < N) {
                                    float y = vv[c];
                                    if (EPI == 2) { y += bias[n + c]; if (rrow) y += rrow[n + c]; if (relu) y = fmaxf(y, 0.f); }
                                    crow[n + c] = y;
                                }
                        }
                    }
                }
            }
        }
        if (!have_next) break;
        t = t_next; m0 = m0n; n0 = n0n;
    }
}

static int g_wg_per_cu = 4;

template <int TM, int TN, int BK, int EPI, int MINW>
static void launch_v4(const float* X, int64_t M, const float* W, int64_t N, int D, float* C, const float* bias, const float* res, int relu) {
    const int tm = (int)((M + 64 * TM - 1) / (64 * TM)), tn = (int)((N + 64 * TN - 1) / (64 * TN));
    int grid = 256 * g_wg_per_cu;
    if (grid > tm * tn) grid = tm * tn;
    hipLaunchKernelGGL((gemm_v4<TM, TN, BK, EPI, MINW>), dim3(grid), dim3(256), 0, 0, X, M, W, N, D, C, N, tm, tn, bias, res, relu);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <class F>
static float time_ms(F f, int it) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); CK(hipDeviceSynchronize());
    hipEventRecord(a);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(b); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / it;
}

struct Variant { const char* name; int wg; void (*fn)(const float*, int64_t, const float*, int64_t, int, float*, const float*, const float*, int); };

int main(int argc, char** argv) {
    // usage: gemm_v4_lab M N D mode rounds      mode: 0 plain GEMM (vs isx_cosine_sim), 1 conv epilogue bias+relu, 2 conv + residual
    const int64_t M = argc > 1 ? atoll(argv[1]) : 10000, N = argc > 2 ? atoll(argv[2]) : 32768;
    const int D = argc > 3 ? atoi(argv[3]) : 2048;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int rounds = argc > 5 ? atoi(argv[5]) : 3;
    std::vector<float> hq((size_t)M * D), hg((size_t)N * D), hb(N), hr(mode == 2 ? (size_t)M * N : 1);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hq) v = rnd() * 0.05f;
    for (auto& v : hg) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd() * 0.01f;
    for (auto& v : hr) v = rnd() * 0.01f;
    float *dq, *dg, *c0, *c1, *db, *dr = nullptr;
    CK(hipMalloc(&dq, hq.size() * 4)); CK(hipMalloc(&dg, hg.size() * 4)); CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    if (mode == 2) { CK(hipMalloc(&dr, hr.size() * 4)); CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice)); }
    const double fl = 2.0 * M * N * D;
    const double bytes = 4.0 * ((double)M * D + (double)M * N * (mode == 2 ? 2 : 1) + (double)N * D);
    const int it = fl > 5e11 ? 3 : 20;
    printf("shape M=%lld N=%lld D=%d mode=%d  (%.2f GFLOP, %.1f MB algorithmic)\n", (long long)M, (long long)N, D, mode, fl * 1e-9, bytes * 1e-6);
    auto ref = [&](float* out) {
        if (mode == 0) isx_cosine_sim(dq, M, dg, N, D, out, nullptr);
        else isx_conv1x1_nhwc(dq, M, D, dg, (int)N, db, dr, 1, out, nullptr);
    };
#define V(TM_, TN_, BK_, W_, WG_) {"v4 " #TM_ "x" #TN_ " BK" #BK_ " w" #W_ " wg" #WG_, WG_, mode == 0 ? launch_v4<TM_, TN_, BK_, 0, W_> : launch_v4<TM_, TN_, BK_, 2, W_>}
    const Variant vs[] = {
        V(2, 2, 16, 4, 4), V(2, 2, 16, 4, 3), V(2, 2, 32, 3, 3), V(2, 1, 32, 4, 4), V(2, 1, 32, 4, 5), V(1, 2, 32, 4, 4), V(1, 1, 32, 6, 6), V(1, 1, 32, 6, 4), V(2, 1, 16, 4, 4), V(1, 1, 16, 6, 6),
    };
    const int nv = (int)(sizeof(vs) / sizeof(vs[0]));
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemset(c0, 0, (size_t)M * N * 4));
    ref(c0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost));
    for (int v = 0; v < nv; ++v) {
        CK(hipMemset(c1, 0xFF, (size_t)M * N * 4));
        g_wg_per_cu = vs[v].wg;
        vs[v].fn(dq, M, dg, N, D, c1, db, dr, 1);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += (memcmp(&h0[i], &h1[i], 4) != 0);
        printf("%-24s mismatches vs libisx: %zu\n", vs[v].name, bad);
    }
    std::vector<std::vector<float>> t(nv + 1);
    for (int r = 0; r < rounds; ++r) {
        t[nv].push_back(time_ms([&] { ref(c0); }, it));
        for (int v = 0; v < nv; ++v) {
            g_wg_per_cu = vs[v].wg;
            t[v].push_back(time_ms([&] { vs[v].fn(dq, M, dg, N, D, c1, db, dr, 1); }, it));
        }
    }
    auto report = [&](const char* name, std::vector<float>& x) {
        std::vector<float> y = x;
        for (size_t i = 0; i < y.size(); ++i) for (size_t j = i + 1; j < y.size(); ++j) if (y[j] < y[i]) { float tt = y[i]; y[i] = y[j]; y[j] = tt; }
        const float med = y[y.size() / 2], mn = y[0];
        printf("%-24s median %.3f ms %.1f TF %.0f GB/s | best %.3f ms %.1f TF\n", name, med, fl / med * 1e-9, bytes / med * 1e-6, mn, fl / mn * 1e-9);
    };
    report("libisx (shipped)", t[nv]);
    for (int v = 0; v < nv; ++v) report(vs[v].name, t[v]);
    return 0;
}
