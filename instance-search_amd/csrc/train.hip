// train.hip -- kernels of the siamese triplet training step:
//   * hard / semi-hard negative mining over the epoch's similarity matrix
//     (reference train/siamese_descriptor.py:94-128, siamese_regions.py:94-135)
//   * TripletLoss forward + analytic backward (reference model/custom_modules.py:140-203)
#include "isx_common.hpp"

namespace isx {

// One 256-thread workgroup per positive couple (i1, i2): the negative is the most similar gallery
// item of anchor i1 that does not share its label and (semi-hard phase, FaceNet) is strictly less
// similar than the positive.  Excluded entries are treated as -2 exactly like the reference
// (`sims[ind_exl] = -2; sims.max(0)`), ties go to the smallest index; -1 = every item excluded
// (the caller falls back to a random negative, reference :100-107,129-131).
// sim holds rows [row_base, row_base + rows) of the N x N matrix (row_base = 0, rows = N: the whole matrix).
__global__ __launch_bounds__(256) void mine_negatives_kernel(const float* __restrict__ sim, int64_t N, int64_t row_base,
                                                             const int32_t* __restrict__ lab, const int64_t* __restrict__ i1,
                                                             const int64_t* __restrict__ i2, int semi_hard,
                                                             int64_t* __restrict__ neg) {
    __shared__ uint64_t red[4];
    const int64_t a = i1[blockIdx.x], p = i2[blockIdx.x];
    const float* row = sim + (a - row_base) * N;
    const float sim_pos = row[p];
    const int32_t la = lab[a];
    uint64_t best = 0;
    for (int64_t j = threadIdx.x; j < N; j += 256) {
        const float s = row[j];
        const bool excl = (lab[j] == la) || (semi_hard && s >= sim_pos);
        if (!excl) {
            const uint64_t key = rank_key(s, (uint32_t)j);
            best = key > best ? key : best;
        }
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) best = red[i] > best ? red[i] : best;
        neg[blockIdx.x] = best ? (int64_t)key_idx(best) : -1;
    }
}

// One wave per triplet row.  normalized: l = a.n - a.p + margin ; else l = (|a-p|^2 - |a-n|^2 + 2 margin) / 2;
// loss_rows[b] = max(l, 0)  (the reference zeroes l <= 0, custom_modules.py:166-167).
__global__ __launch_bounds__(256) void triplet_fwd_kernel(const float* __restrict__ A, const float* __restrict__ P,
                                                          const float* __restrict__ Ng, int64_t B, int D, float margin,
                                                          int normalized, float* __restrict__ loss_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *a = A + b * D, *p = P + b * D, *n = Ng + b * D;
    float s = 0.0f;
    for (int j = lane; j < D; j += 64) {
        if (normalized) s += a[j] * n[j] - a[j] * p[j];
        else { const float dp = a[j] - p[j], dn = a[j] - n[j]; s += dp * dp - dn * dn; }
    }
    s = wave_sum(s);
    float l = normalized ? s + margin : (s + 2.0f * margin) * 0.5f;
    if (lane == 0) loss_rows[b] = l > 0.0f ? l : 0.0f;
}

// Gradients (custom_modules.py:173-203), rows with zero loss get zero gradient, everything scaled by
// `scale` (= grad_output [/ B when size_average]).
__global__ __launch_bounds__(256) void triplet_bwd_kernel(const float* __restrict__ A, const float* __restrict__ P,
                                                          const float* __restrict__ Ng, const float* __restrict__ loss_rows,
                                                          int64_t B, int D, float scale, const float* __restrict__ scale_dev, int normalized,
                                                          float* __restrict__ gA, float* __restrict__ gP, float* __restrict__ gN) {
    if (scale_dev) scale = scale * scale_dev[0];          // grad_output left on the device: no host read-back in the backward pass
    const int64_t total = B * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / D;
        const bool on = loss_rows[b] > 0.0f;
        const float a = A[i], p = P[i], n = Ng[i];
        float ga = n - p, gp, gn;
        if (normalized) { gp = -a; gn = a; } else { gp = p - a; gn = a - n; }
        gA[i] = on ? ga * scale : 0.0f;
        gP[i] = on ? gp * scale : 0.0f;
        gN[i] = on ? gn * scale : 0.0f;
    }
}

// The triplet loss of ALL micro-batches ("leaves") of an optimizer step in one launch (round 6): d holds, leaf by leaf, the k anchor rows, the k
// positive rows and the k negative rows of the leaf (the layout the head engine produces: 3 k rows per leaf).  One workgroup per leaf, one wave per
// triplet row in turn: the row's loss exactly as triplet_fwd_kernel forms it, its three gradient rows exactly as triplet_bwd_kernel forms them
// (scale = scale_a * scale_b, the product the per-leaf path forms from 1 / k and autograd's grad_output), and the leaf's loss = the rows' losses
// added in row order.  Replaces, per leaf: forward launch, torch's sum, backward launch and four copies -- 8 x 36 us of a 11 ms step.
__global__ __launch_bounds__(256) void triplet_leaves_kernel(const float* __restrict__ d, int k, int D, float margin, int normalized, float scale_a,
                                                             float scale_b, float* __restrict__ loss_leaf, float* __restrict__ dd) {
    extern __shared__ float rows[];                                   // k row losses
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * 3 * k;
    const float scale = scale_a * scale_b;
    for (int r = wave; r < k; r += 4) {
        const float *a = d + (base + r) * D, *p = d + (base + k + r) * D, *n = d + (base + 2 * k + r) * D;
        float s = 0.0f;
        for (int j = lane; j < D; j += 64) {
            if (normalized) s += a[j] * n[j] - a[j] * p[j];
            else { const float dp = a[j] - p[j], dn = a[j] - n[j]; s += dp * dp - dn * dn; }
        }
        s = wave_sum(s);
        float l = normalized ? s + margin : (s + 2.0f * margin) * 0.5f;
        l = l > 0.0f ? l : 0.0f;
        if (lane == 0) rows[r] = l;
        const bool on = l > 0.0f;
        float *ga = dd + (base + r) * D, *gp = dd + (base + k + r) * D, *gn = dd + (base + 2 * k + r) * D;
        for (int j = lane; j < D; j += 64) {
            const float av = a[j], pv = p[j], nv = n[j];
            float x = nv - pv, y, z;
            if (normalized) { y = -av; z = av; } else { y = pv - av; z = av - nv; }
            ga[j] = on ? x * scale : 0.0f;
            gp[j] = on ? y * scale : 0.0f;
            gn[j] = on ? z * scale : 0.0f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int r = 0; r < k; ++r) t += rows[r];
        loss_leaf[blockIdx.x] = t;
    }
}

}  // namespace isx

using namespace isx;

ISX_API int isx_triplet_leaves(const float* d, int leaves, int k, int D, float margin, int normalized, float scale_a, float scale_b,
                               float* loss_leaf, float* dd, isx_stream_t stream) {
    ISX_REQUIRE(leaves >= 0 && k > 0 && k <= 8192 && D > 0, "isx_triplet_leaves: bad shape leaves=%d k=%d D=%d (k <= 8192)", leaves, k, D);
    if (leaves == 0) return ISX_OK;
    ISX_REQUIRE(d && loss_leaf && dd && dd != d, "isx_triplet_leaves: null pointer or dd aliases d");
    hipLaunchKernelGGL(triplet_leaves_kernel, dim3((unsigned)leaves), dim3(256), (size_t)k * sizeof(float), (hipStream_t)stream, d, k, D, margin, normalized,
                       scale_a, scale_b, loss_leaf, dd);
    ISX_CHECK_LAUNCH("isx_triplet_leaves");
    return ISX_OK;
}

ISX_API int isx_mine_negatives(const float* sim, int64_t N, const int32_t* labels, const int64_t* i1, const int64_t* i2,
                               int64_t n_couples, int semi_hard, int64_t* neg, isx_stream_t stream) {
    ISX_REQUIRE(N > 0 && N <= 0xFFFFFFFFll && n_couples >= 0 && n_couples < (1ll << 31), "isx_mine_negatives: bad shape N=%lld couples=%lld", (long long)N, (long long)n_couples);
    if (n_couples == 0) return ISX_OK;
    ISX_REQUIRE(sim && labels && i1 && i2 && neg, "isx_mine_negatives: null pointer");
    hipLaunchKernelGGL(mine_negatives_kernel, dim3((unsigned)n_couples), dim3(256), 0, (hipStream_t)stream, sim, N, (int64_t)0, labels, i1, i2, semi_hard, neg);
    ISX_CHECK_LAUNCH("isx_mine_negatives");
    return ISX_OK;
}

// The same mining on a BLOCK of anchor rows: sim_rows = rows [row_base, row_base + rows) of the N x N matrix, every i1[c] inside
// that range (the caller's contract; i1 / i2 stay absolute indices).  For similarity matrices that are never built whole.
ISX_API int isx_mine_negatives_rows(const float* sim_rows, int64_t N, int64_t row_base, int64_t rows, const int32_t* labels, const int64_t* i1,
                                    const int64_t* i2, int64_t n_couples, int semi_hard, int64_t* neg, isx_stream_t stream) {
    ISX_REQUIRE(N > 0 && N <= 0xFFFFFFFFll && n_couples >= 0 && n_couples < (1ll << 31) && row_base >= 0 && rows >= 0 && row_base + rows <= N,
                "isx_mine_negatives_rows: bad shape N=%lld row_base=%lld rows=%lld couples=%lld", (long long)N, (long long)row_base, (long long)rows, (long long)n_couples);
    if (n_couples == 0) return ISX_OK;
    ISX_REQUIRE(sim_rows && labels && i1 && i2 && neg && rows > 0, "isx_mine_negatives_rows: null pointer or empty block");
    hipLaunchKernelGGL(mine_negatives_kernel, dim3((unsigned)n_couples), dim3(256), 0, (hipStream_t)stream, sim_rows, N, row_base, labels, i1, i2, semi_hard, neg);
    ISX_CHECK_LAUNCH("isx_mine_negatives_rows");
    return ISX_OK;
}

ISX_API int isx_triplet_loss_fwd(const float* anchor, const float* pos, const float* neg, int64_t B, int D, float margin,
                                 int normalized, float* loss_rows, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D > 0 && B < (1ll << 31), "isx_triplet_loss_fwd: bad shape B=%lld D=%d", (long long)B, D);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(anchor && pos && neg && loss_rows, "isx_triplet_loss_fwd: null pointer");
    hipLaunchKernelGGL(triplet_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, anchor, pos, neg, B, D, margin, normalized, loss_rows);
    ISX_CHECK_LAUNCH("isx_triplet_loss_fwd");
    return ISX_OK;
}

ISX_API int isx_triplet_loss_bwd(const float* anchor, const float* pos, const float* neg, const float* loss_rows, int64_t B, int D,
                                 float scale, int normalized, float* g_anchor, float* g_pos, float* g_neg, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D > 0 && B < (1ll << 31), "isx_triplet_loss_bwd: bad shape B=%lld D=%d", (long long)B, D);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(anchor && pos && neg && loss_rows && g_anchor && g_pos && g_neg, "isx_triplet_loss_bwd: null pointer");
    const int64_t total = B * D;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, anchor, pos, neg, loss_rows, B, D, scale, (const float*)nullptr, normalized, g_anchor, g_pos, g_neg);
    ISX_CHECK_LAUNCH("isx_triplet_loss_bwd");
    return ISX_OK;
}

// The same with the incoming gradient as a DEVICE scalar: every gradient is multiplied by scale * scale_dev[0].  autograd hands grad_output over as
// a device tensor; reading it on the host costs a synchronisation per micro-batch of the training step.
ISX_API int isx_triplet_loss_bwd_dev(const float* anchor, const float* pos, const float* neg, const float* loss_rows, int64_t B, int D,
                                     float scale, const float* scale_dev, int normalized, float* g_anchor, float* g_pos, float* g_neg,
                                     isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && D > 0, "isx_triplet_loss_bwd_dev: bad shape B=%lld D=%d", (long long)B, D);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(anchor && pos && neg && loss_rows && scale_dev && g_anchor && g_pos && g_neg, "isx_triplet_loss_bwd_dev: null pointer");
    const int64_t total = B * D;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(triplet_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, anchor, pos, neg, loss_rows, B, D, scale, scale_dev, normalized, g_anchor, g_pos, g_neg);
    ISX_CHECK_LAUNCH("isx_triplet_loss_bwd_dev");
    return ISX_OK;
}
