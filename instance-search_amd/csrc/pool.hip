// pool.hip -- descriptor-head kernels: L2 row normalisation (+Shift), fused global
// average pool + L2, stride-1 box pooling.  All HBM-bound: coalesced 16 B/lane loads,
// LDS staging where a thread must walk a strided segment, wave-shuffle reductions.
//
// Reference behaviour restated (paths under the reference repo):
//   model/custom_modules.py:52-57  NormalizeL2Fun.forward
//   model/custom_modules.py:16-18  ShiftFun.forward
//   model/siamese.py:49-54         TuneClassif.forward (AvgPool2d(7) -> view)
//   model/siamese.py:67-71         nn.AvgPool2d(feature_size2d, stride=1)
#include "isx_common.hpp"

namespace isx {

// ------------------------------------------------------------------ l2norm ----
// One wave per row, the row cached in registers (D <= 64*4*ITERS, 16-B aligned rows).
template <int ITERS>
__global__ __launch_bounds__(256) void l2norm_rows_wave_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ shift, int64_t B, int D,
                                                               float eps, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const int nv = D >> 2;
    float4 v[ITERS];
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = lane + i * 64;
        v[i] = (c < nv) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    ss = wave_sum(ss);
    const float n = sqrtf(ss + eps);
    float4* yr = reinterpret_cast<float4*>(y + row * D);
    const float4* sh = reinterpret_cast<const float4*>(shift);
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            float4 o = make_float4(v[i].x / n, v[i].y / n, v[i].z / n, v[i].w / n);
            if (shift) { float4 s = sh[c]; o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w; }
            yr[c] = o;
        }
    }
}

// One 1024-thread block per row, two passes (the second re-reads the row from L2);
// any D, any alignment when VEC == false.
template <bool VEC>
__global__ __launch_bounds__(1024) void l2norm_rows_block_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ shift, int64_t D, float eps,
                                                                 float* __restrict__ y) {
    __shared__ float red[16];
    const float* xr = x + (int64_t)blockIdx.x * D;
    float* yr = y + (int64_t)blockIdx.x * D;
    float ss = 0.0f;
    if (VEC) {
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        for (int64_t c = threadIdx.x; c < (D >> 2); c += 1024) {
            float4 v = x4[c];
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (int64_t c = threadIdx.x; c < D; c += 1024) ss += xr[c] * xr[c];
    }
    ss = block_sum<1024>(ss, red);
    const float n = sqrtf(ss + eps);
    if (VEC) {
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        const float4* s4 = reinterpret_cast<const float4*>(shift);
        float4* y4 = reinterpret_cast<float4*>(yr);
        for (int64_t c = threadIdx.x; c < (D >> 2); c += 1024) {
            float4 v = x4[c];
            float4 o = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
            if (shift) { float4 s = s4[c]; o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w; }
            y4[c] = o;
        }
    } else {
        for (int64_t c = threadIdx.x; c < D; c += 1024) yr[c] = xr[c] / n + (shift ? shift[c] : 0.0f);
    }
}

// ------------------------------------------------------------------- gap_l2 ---
// One 256-thread block per image.  Channels are processed in passes of CP channels:
// the CP*HW floats of a pass are contiguous in NCHW, so they are streamed into LDS
// with 16-B coalesced loads; thread c then sums its channel's HW values IN ORDER (the
// summation order of torch's CPU avg_pool2d, so pooled values are bit-identical to
// the oracle's).  LDS channel stride is HW|1 (odd) -> conflict-free ds_read_b32.
// Pooled values stay in registers (<= MAXP passes); sum of squares by butterfly.
constexpr int kGapThreads = 256;
constexpr int kGapMaxPasses = 16;

template <bool VEC>
__global__ __launch_bounds__(kGapThreads) void gap_l2_kernel(const float* __restrict__ fmap, int C, int HW, int CP,
                                                            int stride, float eps, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[kGapThreads / 64];
    const int tid = threadIdx.x;
    const float* img = fmap + (int64_t)blockIdx.x * C * HW;
    const float inv_div = (float)HW;
    float pooled[kGapMaxPasses];
    float ss = 0.0f;
    const int passes = (C + CP - 1) / CP;
#pragma unroll 1
    for (int p = 0; p < passes; ++p) {
        const int c0 = p * CP;
        const int nch = min(CP, C - c0);
        const int nel = nch * HW;
        const float* src = img + (int64_t)c0 * HW;
        __syncthreads();   // previous pass fully consumed
        if (VEC && stride == HW) {
            // linear image: float4 copy (src 16-B aligned: host checked (CP*HW)%4==0)
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(lds);
            const int nv = nel >> 2;
            for (int i = tid; i < nv; i += kGapThreads) d4[i] = s4[i];
            for (int i = (nv << 2) + tid; i < nel; i += kGapThreads) lds[i] = src[i];
        } else if (VEC) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            const int nv = nel >> 2;
            for (int i = tid; i < nv; i += kGapThreads) {
                float4 v = s4[i];
                int e = i << 2;
                int ch = e / HW, off = e - ch * HW;
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lds[ch * stride + off] = vv[j];
                    if (++off == HW) { off = 0; ++ch; }
                }
            }
            for (int i = (nv << 2) + tid; i < nel; i += kGapThreads) { int ch = i / HW; lds[ch * stride + (i - ch * HW)] = src[i]; }
        } else {
            for (int i = tid; i < nel; i += kGapThreads) { int ch = i / HW; lds[ch * stride + (i - ch * HW)] = src[i]; }
        }
        __syncthreads();
        float s = 0.0f;
        if (tid < nch) {
            const float* r = lds + tid * stride;
            for (int i = 0; i < HW; ++i) s += r[i];
            s = s / inv_div;
        }
        // static register index (guide rule 20): unrolled select
#pragma unroll
        for (int q = 0; q < kGapMaxPasses; ++q) if (q == p) pooled[q] = s;
        ss += s * s;
    }
    ss = block_sum<kGapThreads>(ss, red);
    const float n = sqrtf(ss + eps);
    float* yr = y + (int64_t)blockIdx.x * C;
#pragma unroll
    for (int q = 0; q < kGapMaxPasses; ++q) {
        const int c = q * CP + tid;
        if (q < passes && tid < CP && c < C) yr[c] = pooled[q] / n;
    }
}

// ------------------------------------------------------------------ gap_l2 (NHWC) --
// Channels-last feature map (B, H*W, C) in memory -- what MIOpen's NHWC convolutions produce (the
// fp32 ResNet-50 trunk runs ~9 % faster in that layout).  One 512-thread workgroup per image; a
// thread owns up to QPT groups of 4 consecutive channels and walks the H*W positions IN ORDER with
// 16-B loads (fully coalesced rows of C floats; no LDS needed), so the pooled values have the same
// summation order as the NCHW kernel and the oracle.
constexpr int kNhwcThreads = 512;
#ifndef ISX_GAP_UNROLL
#define ISX_GAP_UNROLL 7        // positions requested per thread before the first is added (A/B)
#endif
#ifndef ISX_GAP_NT_BYTES
#define ISX_GAP_NT_BYTES (192ll << 20)     // maps above this size are read with NON-TEMPORAL loads (isx_gap_l2_nhwc); A/B: 0 = always, a huge value = never
#endif

// NT: the map is read with non-temporal loads.  A map that cannot stay in the 256 MB Infinity Cache anyway (the bench step's 1024 x 2048 x 7 x 7 is
// 401 MB) streams through faster when its lines are not allocated on the way: 73.8 -> 62.8 us back to back = 5.7 -> 6.7 TB/s (0.71 -> 0.83 of 8 TB/s), and
// 110 -> 63 us when other data went through the caches in between, as in the step; a map that fits (256 images, 100 MB) and is re-read hot loses
// (17.9 -> 26.4 us), hence the size switch in the launcher (tools/gap_lab.py; round 6).  Same arithmetic, same order.
template <int QPT, bool NT>
__global__ __launch_bounds__(kNhwcThreads) void gap_l2_nhwc_kernel(const float* __restrict__ fmap, int C, int HW, float eps,
                                                                   float* __restrict__ y) {
    __shared__ float red[kNhwcThreads / 64];
    const int nq = C >> 2;
    const float4* img = reinterpret_cast<const float4*>(fmap + (int64_t)blockIdx.x * HW * C);
    float4 acc[QPT];
#pragma unroll
    for (int q = 0; q < QPT; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll ISX_GAP_UNROLL
    for (int p = 0; p < HW; ++p) {
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const int c4 = threadIdx.x + q * kNhwcThreads;
            if (c4 < nq) {
                float4 v;
                if constexpr (NT) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v t4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(img) + ((int64_t)p * nq + c4));
                    v = make_float4(t4[0], t4[1], t4[2], t4[3]);
                } else {
                    v = img[(int64_t)p * nq + c4];
                }
                acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
            }
        }
    }
    const float div = (float)HW;
    float ss = 0.0f;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        acc[q].x /= div; acc[q].y /= div; acc[q].z /= div; acc[q].w /= div;
        ss += acc[q].x * acc[q].x + acc[q].y * acc[q].y + acc[q].z * acc[q].z + acc[q].w * acc[q].w;
    }
    ss = block_sum<kNhwcThreads>(ss, red);
    const float n = sqrtf(ss + eps);
    float4* yr = reinterpret_cast<float4*>(y + (int64_t)blockIdx.x * C);
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        const int c4 = threadIdx.x + q * kNhwcThreads;
        if (c4 < nq) yr[c4] = make_float4(acc[q].x / n, acc[q].y / n, acc[q].z / n, acc[q].w / n);
    }
}

// generic NHWC pool (any C / alignment): one thread per channel, pooled written to y
__global__ __launch_bounds__(256) void gap_only_nhwc_kernel(const float* __restrict__ fmap, int C, int HW, float* __restrict__ y) {
    const float* img = fmap + (int64_t)blockIdx.x * HW * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.0f;
        for (int p = 0; p < HW; ++p) s += img[(int64_t)p * C + c];
        y[(int64_t)blockIdx.x * C + c] = s / (float)HW;
    }
}

// Fallback for huge maps / channel counts: wave per (image, channel) sum, pooled written
// to y, then the row kernel normalises in place.
__global__ __launch_bounds__(256) void gap_only_kernel(const float* __restrict__ fmap, int64_t BC, int HW,
                                                       float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t bc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bc >= BC) return;
    const float* p = fmap + bc * HW;
    float s = 0.0f;
    for (int i = lane; i < HW; i += 64) s += p[i];
    s = wave_sum(s);
    if (lane == 0) y[bc] = s / (float)HW;
}

// ---------------------------------------------------------------- boxpool_s1 --
// PP whole (b,c) planes per block staged in LDS; each thread produces outputs by an
// in-order kh x kw walk (bit-identical to torch's CPU avg_pool2d).
__global__ __launch_bounds__(256) void boxpool_s1_kernel(const float* __restrict__ fmap, int64_t BC, int H, int W, int kh,
                                                         int kw, int PP, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int HWi = H * W, Ho = H - kh + 1, Wo = W - kw + 1, HWo = Ho * Wo;
    const int64_t p0 = (int64_t)blockIdx.x * PP;
    const int np = (int)min((int64_t)PP, BC - p0);
    const float* src = fmap + p0 * HWi;
    for (int i = threadIdx.x; i < np * HWi; i += 256) lds[i] = src[i];
    __syncthreads();
    const float div = (float)(kh * kw);
    float* dst = out + p0 * HWo;
    for (int o = threadIdx.x; o < np * HWo; o += 256) {
        const int pl = o / HWo, r = o - pl * HWo, i = r / Wo, j = r - i * Wo;
        const float* b = lds + pl * HWi + i * W + j;
        float s = 0.0f;
        for (int a = 0; a < kh; ++a)
            for (int c = 0; c < kw; ++c) s += b[a * W + c];
        dst[o] = s / div;
    }
}

__global__ __launch_bounds__(256) void boxpool_s1_direct_kernel(const float* __restrict__ fmap, int64_t total, int H, int W,
                                                                int kh, int kw, float* __restrict__ out) {
    const int Ho = H - kh + 1, Wo = W - kw + 1, HWo = Ho * Wo;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t pl = o / HWo;
        const int r = (int)(o - pl * HWo), i = r / Wo, j = r - i * Wo;
        const float* b = fmap + pl * H * W + i * W + j;
        float s = 0.0f;
        for (int a = 0; a < kh; ++a)
            for (int c = 0; c < kw; ++c) s += b[a * W + c];
        out[o] = s / (float)(kh * kw);
    }
}

// Channels-last variant: fmap (B,H,W,C), out (B,Ho,Wo,C).  One workgroup per (image, slice of CS channels): the slice of the whole map is
// staged in LDS with 16-B loads (a pixel's CS channels are contiguous), a thread owns 4 consecutive channels of an output location and walks the
// window rows-then-columns from +0 exactly as the NCHW kernel does (same values bit for bit).
__global__ __launch_bounds__(256) void boxpool_s1_nhwc_kernel(const float* __restrict__ fmap, int C, int H, int W, int kh, int kw, int CS,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Ho = H - kh + 1, Wo = W - kw + 1, HWi = H * W, HWo = Ho * Wo;
    const int q = CS >> 2;                                       // float4 per pixel of the slice
    const int c0 = blockIdx.x * CS;
    const float* src = fmap + (int64_t)blockIdx.y * HWi * C + c0;
    float4* l4 = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < HWi * q; i += 256) {
        const int p = i / q, cq = i - p * q;
        l4[i] = *reinterpret_cast<const float4*>(src + (int64_t)p * C + cq * 4);
    }
    __syncthreads();
    const float div = (float)(kh * kw);
    float* dst = out + (int64_t)blockIdx.y * HWo * C + c0;
    for (int o = threadIdx.x; o < HWo * q; o += 256) {
        const int p = o / q, cq = o - p * q, i = p / Wo, j = p - i * Wo;
        const float4* b = l4 + (i * W + j) * q + cq;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < kh; ++a)
            for (int c = 0; c < kw; ++c) {
                const float4 v = b[(a * W + c) * q];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        *reinterpret_cast<float4*>(dst + (int64_t)p * C + cq * 4) = make_float4(s.x / div, s.y / div, s.z / div, s.w / div);
    }
}

}  // namespace isx

using namespace isx;

static int launch_l2norm(const float* x, const float* shift, int64_t B, int64_t D, float eps, float* y, hipStream_t st) {
    if (B == 0 || D == 0) return ISX_OK;
    const bool aligned = (D % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)shift) % 16 == 0);
    if (aligned && D <= 64 * 4 * 8) {
        const unsigned grid = (unsigned)((B + 3) / 4);
        if (D <= 256) hipLaunchKernelGGL(l2norm_rows_wave_kernel<1>, dim3(grid), dim3(256), 0, st, x, shift, B, (int)D, eps, y);
        else if (D <= 512) hipLaunchKernelGGL(l2norm_rows_wave_kernel<2>, dim3(grid), dim3(256), 0, st, x, shift, B, (int)D, eps, y);
        else if (D <= 1024) hipLaunchKernelGGL(l2norm_rows_wave_kernel<4>, dim3(grid), dim3(256), 0, st, x, shift, B, (int)D, eps, y);
        else hipLaunchKernelGGL(l2norm_rows_wave_kernel<8>, dim3(grid), dim3(256), 0, st, x, shift, B, (int)D, eps, y);
    } else if (aligned) {
        hipLaunchKernelGGL(l2norm_rows_block_kernel<true>, dim3((unsigned)B), dim3(1024), 0, st, x, shift, D, eps, y);
    } else {
        hipLaunchKernelGGL(l2norm_rows_block_kernel<false>, dim3((unsigned)B), dim3(1024), 0, st, x, shift, D, eps, y);
    }
    ISX_CHECK_LAUNCH("isx_l2norm_rows");
    return ISX_OK;
}

ISX_API int isx_l2norm_rows(const float* x, int64_t B, int64_t D, float eps, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D >= 0 && B < (1ll << 31), "isx_l2norm_rows: bad shape B=%lld D=%lld", (long long)B, (long long)D);
    ISX_REQUIRE((x && y) || B * D == 0, "isx_l2norm_rows: null pointer");
    return launch_l2norm(x, nullptr, B, D, eps, y, (hipStream_t)stream);
}

ISX_API int isx_l2norm_shift_rows(const float* x, const float* shift, int64_t B, int64_t F, float eps, float* y,
                                  isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && F >= 0 && B < (1ll << 31), "isx_l2norm_shift_rows: bad shape B=%lld F=%lld", (long long)B, (long long)F);
    ISX_REQUIRE((x && y) || B * F == 0, "isx_l2norm_shift_rows: null pointer");
    return launch_l2norm(x, shift, B, F, eps, y, (hipStream_t)stream);
}

// ---- backward of y = x / sqrt(sum_j x_j^2 + eps) (reference model/custom_modules.py:59-67 NormalizeL2Fun.backward) ----
//   n2 = sum x^2 + eps, c = sum x dy:   dx = (n2 dy - x c) / (n2 sqrt(n2))
// one workgroup per row, both sums by the fixed-order block reduction, one elementwise pass; replaces six torch kernels per call
// (two calls per micro-batch of the siamese training step, whose head is host-bound).
namespace isx {
__global__ __launch_bounds__(1024) void l2norm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t D, float eps,
                                                               float* __restrict__ dx) {
    __shared__ float red[16];
    const float* xr = x + (int64_t)blockIdx.x * D;
    const float* gr = dy + (int64_t)blockIdx.x * D;
    float* o = dx + (int64_t)blockIdx.x * D;
    float s2 = 0.0f, sc = 0.0f;
    for (int64_t j = threadIdx.x; j < D; j += 1024) {
        const float v = xr[j];
        s2 += v * v;
        sc += v * gr[j];
    }
    const float n2 = block_sum<1024>(s2, red) + eps;
    const float c = block_sum<1024>(sc, red);
    const float inv = 1.0f / (n2 * sqrtf(n2));
    for (int64_t j = threadIdx.x; j < D; j += 1024) o[j] = (n2 * gr[j] - xr[j] * c) * inv;
}
}  // namespace isx

ISX_API int isx_l2norm_rows_bwd(const float* x, const float* dy, int64_t B, int64_t D, float eps, float* dx, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D >= 0 && B < (1ll << 31), "isx_l2norm_rows_bwd: bad shape B=%lld D=%lld", (long long)B, (long long)D);
    if (B * D == 0) return ISX_OK;
    ISX_REQUIRE(x && dy && dx, "isx_l2norm_rows_bwd: null pointer");
    hipLaunchKernelGGL(isx::l2norm_rows_bwd_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, x, dy, D, eps, dx);
    ISX_CHECK_LAUNCH("isx_l2norm_rows_bwd");
    return ISX_OK;
}

ISX_API int isx_gap_l2(const float* fmap, int64_t B, int C, int H, int W, float eps, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && B < (1ll << 31), "isx_gap_l2: bad shape B=%lld C=%d H=%d W=%d", (long long)B, C, H, W);
    ISX_REQUIRE(fmap && y, "isx_gap_l2: null pointer");
    if (B == 0) return ISX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const int stride = HW | 1;
    // 26 KB per workgroup (6 per CU) keeps more loads in flight: 4.6-5.4 TB/s vs 3.3-4.2 with 52 KB at
    // B >= 1024; small batches have too few workgroups for that to matter and prefer fewer passes
    const size_t budget = (B >= 512 ? 26 : 52) * 1024;
    int CP = kGapThreads;
    while (CP > 32 && (size_t)CP * stride * 4 > budget) CP >>= 1;
    const bool fits = (size_t)CP * stride * 4 <= budget && (C + CP - 1) / CP <= kGapMaxPasses;
    if (!fits) {
        const int64_t BC = B * C;
        hipLaunchKernelGGL(gap_only_kernel, dim3((unsigned)((BC + 3) / 4)), dim3(256), 0, st, fmap, BC, HW, y);
        ISX_CHECK_LAUNCH("isx_gap_l2(pool)");
        return launch_l2norm(y, nullptr, B, C, eps, y, st);
    }
    const bool vec = ((uintptr_t)fmap % 16 == 0) && (((int64_t)C * HW) % 4 == 0) && (((int64_t)CP * HW) % 4 == 0);
    const size_t lds = (size_t)CP * stride * 4;
    if (vec) hipLaunchKernelGGL(gap_l2_kernel<true>, dim3((unsigned)B), dim3(kGapThreads), lds, st, fmap, C, HW, CP, stride, eps, y);
    else hipLaunchKernelGGL(gap_l2_kernel<false>, dim3((unsigned)B), dim3(kGapThreads), lds, st, fmap, C, HW, CP, stride, eps, y);
    ISX_CHECK_LAUNCH("isx_gap_l2");
    return ISX_OK;
}

ISX_API int isx_gap_l2_nhwc(const float* fmap, int64_t B, int C, int H, int W, float eps, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && B < (1ll << 31), "isx_gap_l2_nhwc: bad shape B=%lld C=%d H=%d W=%d", (long long)B, C, H, W);
    ISX_REQUIRE(fmap && y, "isx_gap_l2_nhwc: null pointer");
    if (B == 0) return ISX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const bool vec = (C % 4 == 0) && (((uintptr_t)fmap | (uintptr_t)y) % 16 == 0) && (C / 4 <= 4 * kNhwcThreads);
    if (!vec) {
        hipLaunchKernelGGL(gap_only_nhwc_kernel, dim3((unsigned)B), dim3(256), 0, st, fmap, C, HW, y);
        ISX_CHECK_LAUNCH("isx_gap_l2_nhwc(pool)");
        return launch_l2norm(y, nullptr, B, C, eps, y, st);
    }
    const int nq = C / 4;
    const bool nt = (int64_t)B * C * HW * 4 > (int64_t)ISX_GAP_NT_BYTES;
    const dim3 grid((unsigned)B), block(kNhwcThreads);
    if (nq <= kNhwcThreads) {
        if (nt) hipLaunchKernelGGL((gap_l2_nhwc_kernel<1, true>), grid, block, 0, st, fmap, C, HW, eps, y);
        else hipLaunchKernelGGL((gap_l2_nhwc_kernel<1, false>), grid, block, 0, st, fmap, C, HW, eps, y);
    } else if (nq <= 2 * kNhwcThreads) {
        if (nt) hipLaunchKernelGGL((gap_l2_nhwc_kernel<2, true>), grid, block, 0, st, fmap, C, HW, eps, y);
        else hipLaunchKernelGGL((gap_l2_nhwc_kernel<2, false>), grid, block, 0, st, fmap, C, HW, eps, y);
    } else {
        if (nt) hipLaunchKernelGGL((gap_l2_nhwc_kernel<4, true>), grid, block, 0, st, fmap, C, HW, eps, y);
        else hipLaunchKernelGGL((gap_l2_nhwc_kernel<4, false>), grid, block, 0, st, fmap, C, HW, eps, y);
    }
    ISX_CHECK_LAUNCH("isx_gap_l2_nhwc");
    return ISX_OK;
}

ISX_API int isx_boxpool_s1(const float* fmap, int64_t B, int C, int H, int W, int kh, int kw, float* out,
                           isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && kh <= H && kw <= W,
                "isx_boxpool_s1: bad shape B=%lld C=%d H=%d W=%d k=%dx%d", (long long)B, C, H, W, kh, kw);
    ISX_REQUIRE(fmap && out, "isx_boxpool_s1: null pointer");
    if (B == 0) return ISX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t BC = B * C;
    const size_t plane = (size_t)H * W * 4;
    if (plane <= 48 * 1024) {
        int PP = (int)((48 * 1024) / plane);
        if (PP > 64) PP = 64;
        const int64_t grid = (BC + PP - 1) / PP;
        ISX_REQUIRE(grid < (1ll << 31), "isx_boxpool_s1: too many planes");
        hipLaunchKernelGGL(boxpool_s1_kernel, dim3((unsigned)grid), dim3(256), (size_t)PP * plane, st, fmap, BC, H, W, kh, kw, PP, out);
    } else {
        const int64_t total = BC * (H - kh + 1) * (W - kw + 1);
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        hipLaunchKernelGGL(boxpool_s1_direct_kernel, dim3(grid), dim3(256), 0, st, fmap, total, H, W, kh, kw, out);
    }
    ISX_CHECK_LAUNCH("isx_boxpool_s1");
    return ISX_OK;
}

ISX_API int isx_boxpool_s1_nhwc(const float* fmap, int64_t B, int C, int H, int W, int kh, int kw, float* out, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && B < 65536 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && kh <= H && kw <= W,
                "isx_boxpool_s1_nhwc: bad shape B=%lld C=%d H=%d W=%d k=%dx%d", (long long)B, C, H, W, kh, kw);
    ISX_REQUIRE(C % 4 == 0 && (((uintptr_t)fmap | (uintptr_t)out) % 16) == 0, "isx_boxpool_s1_nhwc: C=%d must be a multiple of 4 and the pointers 16-B aligned", C);
    ISX_REQUIRE(fmap && out && fmap != out, "isx_boxpool_s1_nhwc: null or aliased pointer");
    if (B == 0) return ISX_OK;
    int CS = 64;                                                  // channels per workgroup: the slice of the whole map must fit 64 KB of LDS
    while (CS > 4 && (C % CS != 0 || (size_t)H * W * CS * 4 > 64 * 1024)) CS >>= 1;
    ISX_REQUIRE(C % CS == 0 && (size_t)H * W * CS * 4 <= 64 * 1024, "isx_boxpool_s1_nhwc: a %dx%d map does not fit the LDS staging (use isx_boxpool_s1)", H, W);
    hipLaunchKernelGGL(boxpool_s1_nhwc_kernel, dim3((unsigned)(C / CS), (unsigned)B), dim3(256), (size_t)H * W * CS * 4, (hipStream_t)stream, fmap, C, H, W,
                       kh, kw, CS, out);
    ISX_CHECK_LAUNCH("isx_boxpool_s1_nhwc");
    return ISX_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused convolution epilogue for the inference trunk: y = act(y + bias[c] (+ residual)), in place.
// After BatchNorm folding every convolution of the backbone is followed by bias-add, (residual add,)
// ReLU; as separate framework kernels those are 4 (7 at the end of a residual block) full passes
// over the activation; fused they are 2 (3).  HBM-bound streaming, 16-B accesses.
// Channel of element i: (i / inner) % C  -- inner = 1 for channels-last (NHWC) memory, H*W for NCHW.
namespace isx {

template <bool NHWC4>
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       const float* __restrict__ res, int64_t n4, int C, int64_t inner,
                                                       int relu) {
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* r4 = reinterpret_cast<const float4*>(res);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = y4[i];
        float4 b;
        if (NHWC4) {
            b = *reinterpret_cast<const float4*>(bias + (int)((i * 4) % C));          // C % 4 == 0: 4 consecutive channels
        } else {
            const float bb = bias[(int)(((i * 4) / inner) % C)];                        // inner % 4 == 0: one channel
            b = make_float4(bb, bb, bb, bb);
        }
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (res) { const float4 r = r4[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        y4[i] = v;
    }
}

__global__ __launch_bounds__(256) void bias_act_scalar_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                              const float* __restrict__ res, int64_t n, int C, int64_t inner,
                                                              int relu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = y[i] + bias[(int)((i / inner) % C)] + (res ? res[i] : 0.0f);
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}

// Stem epilogue: relu(conv + bias) followed by MaxPool2d(3, stride 2, padding 1), channels-last, one pass.
// relu(. + b) is monotonic, so the window maximum is taken first: out = relu(max_window(y) + bias[c]).
// One thread per (output pixel, 4 channels): nine 16-B loads, consecutive threads on consecutive channels.
__global__ __launch_bounds__(256) void bias_relu_maxpool_nhwc_kernel(const float* __restrict__ y, const float* __restrict__ bias, int64_t n4,
                                                                     int C4, int H, int W, int Ho, int Wo, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const int64_t b = p / Ho;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = reinterpret_cast<const float4*>(y)[((b * H + hi) * W + wi) * C4 + c4];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        const float4 bb = reinterpret_cast<const float4*>(bias)[c4];
        m.x = fmaxf(m.x + bb.x, 0.f); m.y = fmaxf(m.y + bb.y, 0.f); m.z = fmaxf(m.z + bb.z, 0.f); m.w = fmaxf(m.w + bb.w, 0.f);
        reinterpret_cast<float4*>(out)[i] = m;
    }
}

// Image ingest: transforms.ToTensor() + Normalize(mean, std) of the reference's test mains
// (test/classif_finetune_test.py:62-73) on the GPU, from the decoded uint8 RGB pixels: 3 B read and 12 B written
// per pixel, and only the uint8 batch crosses PCIe (a quarter of the fp32 tensor).  Same fp32 operations in the same
// order as torch: x = u8 / 255, (x - mean[c]) / std[c].  One thread per 4 pixels (12 bytes = three aligned dwords).
__global__ __launch_bounds__(256) void images_u8_to_f32_kernel(const uint8_t* __restrict__ img, int64_t B, int HW, float m0, float m1,
                                                               float m2, float s0, float s1, float s2, int nhwc, float* __restrict__ out) {
    const int64_t quads = ((int64_t)B * HW + 3) / 4;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const int64_t p0 = q * 4;
        const int64_t total = (int64_t)B * HW;
        uint8_t u[12];
        if (p0 + 3 < total && (((uintptr_t)img) & 3) == 0) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(img + p0 * 3);
            const uint32_t a = w[0], b = w[1], c = w[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { u[i] = (a >> (8 * i)) & 255; u[4 + i] = (b >> (8 * i)) & 255; u[8 + i] = (c >> (8 * i)) & 255; }
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) u[i] = (p0 * 3 + i < total * 3) ? img[p0 * 3 + i] : 0;
        }
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = ((float)u[i] / 255.0f - mean[i % 3]) / sd[i % 3];
        if (p0 + 3 < total && nhwc) {
            // 4 pixels x 3 channels = 48 contiguous bytes, 16-B aligned (p0 is a multiple of 4)
            float4* o = reinterpret_cast<float4*>(out + p0 * 3);
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], v[7]);
            o[2] = make_float4(v[8], v[9], v[10], v[11]);
        } else if (p0 + 3 < total && (HW & 3) == 0) {
            // NCHW with HW % 4 == 0: the 4 pixels are in one image, one 16-B store per channel plane
            const int64_t b = p0 / HW, hw = p0 - b * HW;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<float4*>(out + (b * 3 + c) * (int64_t)HW + hw) = make_float4(v[c], v[3 + c], v[6 + c], v[9 + c]);
        } else {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const int64_t p = p0 + px;
                if (p < total) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (nhwc) out[p * 3 + c] = v[px * 3 + c];
                        else { const int64_t b = p / HW; out[(b * 3 + c) * (int64_t)HW + (p - b * HW)] = v[px * 3 + c]; }
                    }
                }
            }
        }
    }
}

}  // namespace isx

ISX_API int isx_images_u8_to_f32(const uint8_t* img, int64_t B, int H, int W, float mean0, float mean1, float mean2, float std0, float std1,
                                 float std2, int channels_last, float* out, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "isx_images_u8_to_f32: bad shape B=%lld H=%d W=%d", (long long)B, H, W);
    ISX_REQUIRE(std0 != 0.0f && std1 != 0.0f && std2 != 0.0f, "isx_images_u8_to_f32: zero std");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(img && out, "isx_images_u8_to_f32: null pointer");
    const int64_t quads = (B * H * W + 3) / 4, blocks = (quads + 255) / 256;
    hipLaunchKernelGGL(isx::images_u8_to_f32_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, img, B,
                       H * W, mean0, mean1, mean2, std0, std1, std2, channels_last ? 1 : 0, out);
    ISX_CHECK_LAUNCH("isx_images_u8_to_f32");
    return ISX_OK;
}

ISX_API int isx_bias_relu_maxpool_nhwc(const float* y, const float* bias, int64_t B, int H, int W, int C, float* out, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "isx_bias_relu_maxpool_nhwc: bad shape B=%lld H=%d W=%d C=%d (C %% 4 == 0)", (long long)B, H, W, C);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(y && bias && out && y != out, "isx_bias_relu_maxpool_nhwc: null or aliased pointer");
    ISX_REQUIRE((((uintptr_t)y | (uintptr_t)bias | (uintptr_t)out) % 16) == 0, "isx_bias_relu_maxpool_nhwc: 16-B alignment required");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t n4 = B * Ho * Wo * (C / 4);
    const int64_t blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(isx::bias_relu_maxpool_nhwc_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, y, bias,
                       n4, C / 4, H, W, Ho, Wo, out);
    ISX_CHECK_LAUNCH("isx_bias_relu_maxpool_nhwc");
    return ISX_OK;
}

ISX_API int isx_bias_act_inplace(float* y, const float* bias, const float* residual, int64_t n, int C, int64_t inner, int relu,
                                 isx_stream_t stream) {
    ISX_REQUIRE(n >= 0 && C > 0 && inner > 0, "isx_bias_act_inplace: bad shape n=%lld C=%d inner=%lld", (long long)n, C, (long long)inner);
    if (n == 0) return ISX_OK;
    ISX_REQUIRE(y && bias, "isx_bias_act_inplace: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)y | (uintptr_t)residual | (uintptr_t)bias) % 16 == 0) && (n % 4 == 0);
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    if (al && inner == 1 && C % 4 == 0) hipLaunchKernelGGL(isx::bias_act_kernel<true>, dim3(grid ? grid : 1), dim3(256), 0, st, y, bias, residual, n4, C, inner, relu);
    else if (al && inner % 4 == 0) hipLaunchKernelGGL(isx::bias_act_kernel<false>, dim3(grid ? grid : 1), dim3(256), 0, st, y, bias, residual, n4, C, inner, relu);
    else hipLaunchKernelGGL(isx::bias_act_scalar_kernel, dim3((unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384)), dim3(256), 0, st, y, bias, residual, n, C, inner, relu);
    ISX_CHECK_LAUNCH("isx_bias_act_inplace");
    return ISX_OK;
}
