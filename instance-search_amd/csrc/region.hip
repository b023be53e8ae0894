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

}  // namespace isx

using namespace isx;

ISX_API int isx_best_location_desc(const float* cls, int64_t B, int K, int Hp, int Wp, float eps, float* desc,
                                   int64_t* loc, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && K > 0 && Hp > 0 && Wp > 0 && B < (1ll << 31) && (int64_t)Hp * Wp < (1ll << 31),
                "isx_best_location_desc: bad shape B=%lld K=%d Hp=%d Wp=%d", (long long)B, K, Hp, Wp);
    ISX_REQUIRE(cls && desc && loc, "isx_best_location_desc: null pointer");
    if (B == 0) return ISX_OK;
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
