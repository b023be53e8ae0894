// region.hip -- fully-convolutional region path: class-max map, best location /
// top-k locations (canonical tie-break), window gather + L2 + Shift.
//
// Reference behaviour restated:
//   train/classif_regions.py:118-128   best-location class-score descriptor
//   model/siamese.py:191-203           RegionDescriptorNet: c.max(1) -> topk -> windows
//   model/siamese.py:215-219           per-window NormalizeL2 -> Shift (-> Linear in caller)
#include "isx_common.hpp"

namespace isx {

// Key for the spatial arg-max of classif_regions: larger = better; ties -> smallest
// column, then smallest row.  low word = ~(col * Hp + row).
__device__ __forceinline__ uint64_t loc_key(float v, int row, int col, int Hp) {
    return ((uint64_t)f32_orderable(v) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(col * Hp + row));
}

// One 256-thread block per image.
__global__ __launch_bounds__(256) void best_location_desc_kernel(const float* __restrict__ cls, int K, int Hp, int Wp,
                                                                 float eps, float* __restrict__ desc,
                                                                 int64_t* __restrict__ loc) {
    __shared__ uint64_t kred[4];
    __shared__ float red[4];
    const int P = Hp * Wp;
    const float* c = cls + (int64_t)blockIdx.x * K * P;
    uint64_t best = 0;
    for (int p = threadIdx.x; p < P; p += 256) {
        float m = c[p];
        for (int k = 1; k < K; ++k) { float v = c[(int64_t)k * P + p]; m = v > m ? v : m; }
        uint64_t key = loc_key(m, p / Wp, p % Wp, Hp);
        best = key > best ? key : best;
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) kred[threadIdx.x >> 6] = best;
    __syncthreads();
    best = kred[0];
    for (int i = 1; i < 4; ++i) best = kred[i] > best ? kred[i] : best;
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
    const int col = flat / Hp, row = flat % Hp;
    const int p = row * Wp + col;
    float ss = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) { float v = c[(int64_t)k * P + p]; ss += v * v; }
    ss = block_sum<256>(ss, red);
    const float n = sqrtf(ss + eps);
    float* d = desc + (int64_t)blockIdx.x * K;
    for (int k = threadIdx.x; k < K; k += 256) d[k] = c[(int64_t)k * P + p] / n;
    if (threadIdx.x == 0) { loc[blockIdx.x * 2] = row; loc[blockIdx.x * 2 + 1] = col; }
}

// One block per image: class-max per location -> keys in LDS -> bitonic sort -> first k.
__global__ __launch_bounds__(256) void region_topk_kernel(const float* __restrict__ cls_all, int K, int P, int PP2, int k,
                                                          int64_t* __restrict__ flat_idx_all, float* __restrict__ score_all) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const float* cls = cls_all + (int64_t)blockIdx.x * K * P;
    int64_t* flat_idx = flat_idx_all + (int64_t)blockIdx.x * k;
    float* score = score_all + (int64_t)blockIdx.x * k;
    for (int p = threadIdx.x; p < PP2; p += 256) {
        uint64_t key = 0;
        if (p < P) {
            float m = cls[p];
            for (int c = 1; c < K; ++c) { float v = cls[(int64_t)c * P + p]; m = v > m ? v : m; }
            key = rank_key(m, (uint32_t)p);
        }
        keys[p] = key;
    }
    bitonic_sort_desc<256>(keys, PP2);
    for (int i = threadIdx.x; i < k; i += 256) {
        if (i < P) { flat_idx[i] = key_idx(keys[i]); score[i] = key_score(keys[i]); }
        else { flat_idx[i] = -1; score[i] = -INFINITY; }
    }
}

// One 1024-thread block per (window, image): pass 1 sum of squares over the gathered C*kh*kw
// values, pass 2 normalise (+shift) and write the row.
__global__ __launch_bounds__(1024) void region_gather_l2_kernel(const float* __restrict__ fmap_all, int C, int Hf, int Wf, int kh,
                                                                int kw, const int64_t* __restrict__ flat_idx, int Wp,
                                                                const float* __restrict__ shift, float eps,
                                                                float* __restrict__ rows) {
    __shared__ float red[16];
    const int khw = kh * kw;
    const int F = C * khw;
    const int64_t w = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;       // window id: image-major
    const float* fmap = fmap_all + (int64_t)blockIdx.y * C * Hf * Wf;
    float* out = rows + w * F;
    const int64_t fi = flat_idx[w];
    const int row = (int)(fi / Wp), col = (int)(fi % Wp);
    if (fi < 0 || row + kh > Hf || col + kw > Wf) {   // padding entry or out-of-range index: zero row, no gather
        for (int j = threadIdx.x; j < F; j += 1024) out[j] = 0.0f;
        return;
    }
    const float* base = fmap + (int64_t)row * Wf + col;
    float ss = 0.0f;
    for (int j = threadIdx.x; j < F; j += 1024) {
        const int c = j / khw, r = j - c * khw, a = r / kw, b = r - a * kw;
        const float v = base[((int64_t)c * Hf + a) * Wf + b];
        ss += v * v;
    }
    ss = block_sum<1024>(ss, red);
    const float n = sqrtf(ss + eps);
    for (int j = threadIdx.x; j < F; j += 1024) {
        const int c = j / khw, r = j - c * khw, a = r / kw, b = r - a * kw;
        const float v = base[((int64_t)c * Hf + a) * Wf + b];
        out[j] = v / n + (shift ? shift[j] : 0.0f);
    }
}

// ---- channels-last variants: cls (B,Hp,Wp,K), fmap (B,Hf,Wf,C) -- the layouts the NHWC trunk and the 1x1-convolution classifier produce --------
// class-max of location p by one wave: lanes stride the K contiguous scores
__device__ __forceinline__ float wave_class_max(const float* __restrict__ row, int K, int lane) {
    float m = -INFINITY;
    for (int k = lane; k < K; k += 64) { const float v = row[k]; m = v > m ? v : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(m, o, 64); m = w > m ? w : m; }
    return m;
}

// One 256-thread block per image; same selection, descriptor arithmetic and reduction order as best_location_desc_kernel.
__global__ __launch_bounds__(256) void best_location_desc_nhwc_kernel(const float* __restrict__ cls, int K, int Hp, int Wp, float eps,
                                                                      float* __restrict__ desc, int64_t* __restrict__ loc) {
    __shared__ uint64_t kred[4];
    __shared__ float red[4];
    const int P = Hp * Wp, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* c = cls + (int64_t)blockIdx.x * K * P;
    uint64_t best = 0;
    for (int p = wave; p < P; p += 4) {
        const float m = wave_class_max(c + (int64_t)p * K, K, lane);
        const uint64_t key = loc_key(m, p / Wp, p % Wp, Hp);
        best = key > best ? key : best;
    }
    if (lane == 0) kred[wave] = best;
    __syncthreads();
    best = kred[0];
    for (int i = 1; i < 4; ++i) best = kred[i] > best ? kred[i] : best;
    const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull);
    const int col = flat / Hp, row = flat % Hp;
    const float* v = c + (int64_t)(row * Wp + col) * K;
    float ss = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) ss += v[k] * v[k];
    ss = block_sum<256>(ss, red);
    const float n = sqrtf(ss + eps);
    float* d = desc + (int64_t)blockIdx.x * K;
    for (int k = threadIdx.x; k < K; k += 256) d[k] = v[k] / n;
    if (threadIdx.x == 0) { loc[blockIdx.x * 2] = row; loc[blockIdx.x * 2 + 1] = col; }
}

__global__ __launch_bounds__(256) void region_topk_nhwc_kernel(const float* __restrict__ cls_all, int K, int P, int PP2, int k,
                                                               int64_t* __restrict__ flat_idx_all, float* __restrict__ score_all) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cls = cls_all + (int64_t)blockIdx.x * K * P;
    int64_t* flat_idx = flat_idx_all + (int64_t)blockIdx.x * k;
    float* score = score_all + (int64_t)blockIdx.x * k;
    for (int p = wave; p < PP2; p += 4) {
        uint64_t key = 0;
        if (p < P) key = rank_key(wave_class_max(cls + (int64_t)p * K, K, lane), (uint32_t)p);
        if (lane == 0) keys[p] = key;
    }
    bitonic_sort_desc<256>(keys, PP2);
    for (int i = threadIdx.x; i < k; i += 256) {
        if (i < P) { flat_idx[i] = key_idx(keys[i]); score[i] = key_score(keys[i]); }
        else { flat_idx[i] = -1; score[i] = -INFINITY; }
    }
}

// One 1024-thread block per (window, image), channels-last map: the window's kh x kw pixels are kh runs of kw * C contiguous floats, the row
// keeps that (h, w, C) order (16-B loads and stores, no transpose; the caller's Shift vector and Linear weight columns are permuted to match).
__global__ __launch_bounds__(1024) void region_gather_l2_nhwc_kernel(const float* __restrict__ fmap_all, int C, int Hf, int Wf, int kh, int kw,
                                                                     const int64_t* __restrict__ flat_idx, int Wp, const float* __restrict__ shift_hwc,
                                                                     float eps, float* __restrict__ rows) {
    __shared__ float red[16];
    const int run4 = kw * C / 4;                               // float4 per window row
    const int F4 = kh * run4;
    const int64_t w = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const float* fmap = fmap_all + (int64_t)blockIdx.y * C * Hf * Wf;
    float4* out = reinterpret_cast<float4*>(rows + w * (int64_t)F4 * 4);
    const int64_t fi = flat_idx[w];
    const int row = (int)(fi / Wp), col = (int)(fi % Wp);
    if (fi < 0 || row + kh > Hf || col + kw > Wf) {
        for (int j = threadIdx.x; j < F4; j += 1024) out[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* base = fmap + ((int64_t)row * Wf + col) * C;
    float ss = 0.0f;
    for (int j = threadIdx.x; j < F4; j += 1024) {
        const int a = j / run4, r = j - a * run4;
        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)a * Wf * C + r * 4);
        ss += v.x * v.x; ss += v.y * v.y; ss += v.z * v.z; ss += v.w * v.w;
    }
    ss = block_sum<1024>(ss, red);
    const float n = sqrtf(ss + eps);
    const float4* sh = reinterpret_cast<const float4*>(shift_hwc);
    for (int j = threadIdx.x; j < F4; j += 1024) {
        const int a = j / run4, r = j - a * run4;
        const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)a * Wf * C + r * 4);
        const float4 s4 = shift_hwc ? sh[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        out[j] = make_float4(v.x / n + s4.x, v.y / n + s4.y, v.z / n + s4.z, v.w / n + s4.w);
    }
}

}  // namespace isx

using namespace isx;

ISX_API int isx_best_location_desc(const float* cls, int64_t B, int K, int Hp, int Wp, float eps, float* desc,
                                   int64_t* loc, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && K > 0 && Hp > 0 && Wp > 0 && B < (1ll << 31) && (int64_t)Hp * Wp < (1ll << 31),
                "isx_best_location_desc: bad shape B=%lld K=%d Hp=%d Wp=%d", (long long)B, K, Hp, Wp);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(cls && desc && loc, "isx_best_location_desc: null pointer");
    hipLaunchKernelGGL(best_location_desc_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, cls, K, Hp, Wp, eps, desc, loc);
    ISX_CHECK_LAUNCH("isx_best_location_desc");
    return ISX_OK;
}

ISX_API int isx_region_topk(const float* cls, int64_t B, int K, int Hp, int Wp, int k, int64_t* flat_idx, float* score,
                            isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && B < 65536 && K > 0 && Hp > 0 && Wp > 0 && k > 0, "isx_region_topk: bad shape B=%lld K=%d Hp=%d Wp=%d k=%d", (long long)B, K, Hp, Wp, k);
    if (B == 0) return ISX_OK;
    const int64_t P = (int64_t)Hp * Wp;
    ISX_REQUIRE(P <= 4096, "isx_region_topk: Hp*Wp=%lld exceeds 4096 locations", (long long)P);
    ISX_REQUIRE(cls && flat_idx && score, "isx_region_topk: null pointer");
    const int PP2 = next_pow2((int)P < 2 ? 2 : (int)P);
    hipLaunchKernelGGL(region_topk_kernel, dim3((unsigned)B), dim3(256), (size_t)PP2 * 8, (hipStream_t)stream, cls, K, (int)P, PP2, k, flat_idx, score);
    ISX_CHECK_LAUNCH("isx_region_topk");
    return ISX_OK;
}

ISX_API int isx_region_gather_l2(const float* fmap, int64_t B, int C, int Hf, int Wf, int kh, int kw, const int64_t* flat_idx, int k,
                                 int Wp, const float* shift, float eps, float* rows, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && B < 65536 && C > 0 && Hf > 0 && Wf > 0 && kh > 0 && kw > 0 && kh <= Hf && kw <= Wf && k >= 0 && Wp == Wf - kw + 1,
                "isx_region_gather_l2: bad shape C=%d Hf=%d Wf=%d k=%dx%d n=%d Wp=%d", C, Hf, Wf, kh, kw, k, Wp);
    ISX_REQUIRE((int64_t)C * kh * kw < (1ll << 31), "isx_region_gather_l2: window too large");
    ISX_REQUIRE(fmap && flat_idx && rows, "isx_region_gather_l2: null pointer");
    if (k == 0 || B == 0) return ISX_OK;
    hipLaunchKernelGGL(region_gather_l2_kernel, dim3((unsigned)k, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, fmap, C, Hf, Wf, kh, kw,
                       flat_idx, Wp, shift, eps, rows);
    ISX_CHECK_LAUNCH("isx_region_gather_l2");
    return ISX_OK;
}

// ---- channels-last entry points (same contracts, cls (B,Hp,Wp,K) / fmap (B,Hf,Wf,C)) -------------------------------------------------------
ISX_API int isx_best_location_desc_nhwc(const float* cls, int64_t B, int K, int Hp, int Wp, float eps, float* desc, int64_t* loc,
                                        isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && K > 0 && Hp > 0 && Wp > 0 && B < (1ll << 31) && (int64_t)Hp * Wp < (1ll << 31),
                "isx_best_location_desc_nhwc: bad shape B=%lld K=%d Hp=%d Wp=%d", (long long)B, K, Hp, Wp);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(cls && desc && loc, "isx_best_location_desc_nhwc: null pointer");
    hipLaunchKernelGGL(best_location_desc_nhwc_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, cls, K, Hp, Wp, eps, desc, loc);
    ISX_CHECK_LAUNCH("isx_best_location_desc_nhwc");
    return ISX_OK;
}

ISX_API int isx_region_topk_nhwc(const float* cls, int64_t B, int K, int Hp, int Wp, int k, int64_t* flat_idx, float* score,
                                 isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && B < 65536 && K > 0 && Hp > 0 && Wp > 0 && k > 0, "isx_region_topk_nhwc: bad shape B=%lld K=%d Hp=%d Wp=%d k=%d", (long long)B, K, Hp, Wp, k);
    if (B == 0) return ISX_OK;
    const int64_t P = (int64_t)Hp * Wp;
    ISX_REQUIRE(P <= 4096, "isx_region_topk_nhwc: Hp*Wp=%lld exceeds 4096 locations", (long long)P);
    ISX_REQUIRE(cls && flat_idx && score, "isx_region_topk_nhwc: null pointer");
    const int PP2 = next_pow2((int)P < 2 ? 2 : (int)P);
    hipLaunchKernelGGL(region_topk_nhwc_kernel, dim3((unsigned)B), dim3(256), (size_t)PP2 * 8, (hipStream_t)stream, cls, K, (int)P, PP2, k, flat_idx, score);
    ISX_CHECK_LAUNCH("isx_region_topk_nhwc");
    return ISX_OK;
}

ISX_API int isx_region_gather_l2_nhwc(const float* fmap, int64_t B, int C, int Hf, int Wf, int kh, int kw, const int64_t* flat_idx, int k,
                                      int Wp, const float* shift_hwc, float eps, float* rows, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && B < 65536 && C > 0 && Hf > 0 && Wf > 0 && kh > 0 && kw > 0 && kh <= Hf && kw <= Wf && k >= 0 && Wp == Wf - kw + 1,
                "isx_region_gather_l2_nhwc: bad shape C=%d Hf=%d Wf=%d k=%dx%d n=%d Wp=%d", C, Hf, Wf, kh, kw, k, Wp);
    ISX_REQUIRE((int64_t)C * kh * kw < (1ll << 31), "isx_region_gather_l2_nhwc: window too large");
    ISX_REQUIRE(C % 4 == 0 && (((uintptr_t)fmap | (uintptr_t)rows | (uintptr_t)shift_hwc) % 16) == 0,
                "isx_region_gather_l2_nhwc: C=%d must be a multiple of 4 and the pointers 16-B aligned", C);
    ISX_REQUIRE(fmap && flat_idx && rows, "isx_region_gather_l2_nhwc: null pointer");
    if (k == 0 || B == 0) return ISX_OK;
    hipLaunchKernelGGL(region_gather_l2_nhwc_kernel, dim3((unsigned)k, (unsigned)B), dim3(1024), 0, (hipStream_t)stream, fmap, C, Hf, Wf, kh, kw,
                       flat_idx, Wp, shift_hwc, eps, rows);
    ISX_CHECK_LAUNCH("isx_region_gather_l2_nhwc");
    return ISX_OK;
}
