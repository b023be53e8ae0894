// isx_common.hpp -- shared device/host helpers of libisx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/isx.h"

#define ISX_API extern "C" __attribute__((visibility("default")))

void isx_set_error(const char* fmt, ...);

#define ISX_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            isx_set_error(__VA_ARGS__);         \
            return ISX_ERR_ARG;                 \
        }                                       \
    } while (0)

#define ISX_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            isx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return ISX_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

namespace isx {

constexpr int kWave = 64;

// ---- canonical ranking key ---------------------------------------------------
// larger key == ranked earlier: (score desc, index asc); -0.0 folded onto +0.0.
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t ob) {
    uint32_t u = (ob & 0x80000000u) ? (ob & 0x7FFFFFFFu) : ~ob;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t rank_key(float score, uint32_t idx) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ uint32_t key_idx(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_score(uint64_t key) { return orderable_f32((uint32_t)(key >> 32)); }

// ---- wave / block reductions (deterministic butterfly) -------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_max(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// Sum over a block of NT threads (NT multiple of 64, <= 1024); every thread gets the
// result.  `red` is shared scratch of >= NT/64 floats.  Fixed order -> deterministic.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// In-LDS bitonic sort of n (power of two) u64 keys, DESCENDING, by NT threads.
template <int NT>
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n >> 1); t += NT) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool desc = ((lo & size) == 0);
                uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace isx
