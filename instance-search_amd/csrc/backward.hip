// backward.hip -- the backward pass of the TRAINABLE trunk suffix of siamese training (reference
// train/siamese_descriptor_p.py:14-17,48: layer4 of the ResNet is trained, model/nn_utils.py:5-23) on channels-last activations.
// The reference leaves these to torch autograd (cuDNN there; MIOpen here runs the 24-image micro-batches as per-image im2col +
// GEMM loops: ~150 launches per micro-batch).  Forward reuses the inference kernels with the BatchNorm folded into the
// convolution (conv.hip); this file adds
//
//   isx_conv1x1_dgrad_nhwc   dX = (dZ . W' (+ add)) (. [mask > 0])     the NT GEMM of cosine.hip, epilogue mode 3
//   isx_conv3x3_dgrad_nhwc   the same for a 3x3 convolution: conv3x3 of dZ with the flipped / transposed weight, mask epilogue
//   isx_conv_wgrad_nhwc      dW'[co][tap][ci] = sum_p dZ[p][co] * X[src(p, tap)][ci]   (1x1, strided 1x1 and 3x3: a TN GEMM over the
//                            pixels; both operands are K-major in memory, so tiles go to LDS without a transpose)
//                            -- split over the pixels into partials, with the bias gradient db[co] = sum_p dZ[p][co] as a by-product
//   isx_relu_grad            dZ = dY . [y > 0] at the output of the last block
//   isx_bn_fold_backward     sums the partials in order, then the chain rule of the fold  w' = w * s, b' = beta - mean * s,
//                            s = gamma / sqrt(var + eps):  dw (+)= dW' * s,  dgamma (+)= (<dW', w> - mean * db) / sqrt(var + eps),  dbeta (+)= db
//
// Every sum runs in a fixed order (k-ordered MFMA chains over the pixels, fixed-shape block reductions): the gradient of a
// micro-batch does not depend on what else is in flight, which the canonical gradient tree of isx/dp.py relies on.
#include <stdlib.h>

#include "wgrad_kernel.hpp"

namespace isx {

int launch_gemm_masked(const float* A, int64_t M, const float* Bt, int64_t N, int D, float* C, const float* mask, const float* add, hipStream_t st);

// Number of pixel splits of a weight-gradient launch -- a function of the SHAPE only, so the summation tree of a micro-batch is the same whatever
// runs around it (a rank with one micro-batch and a process with eight land on the same bits).
// Shapes the 128x128 tiles cover (round 5): the split is chosen for the launch the single-GPU step issues -- 8 micro-batches at once -- by a round
// model: blocks = tiles x 8 x S run in ceil(blocks / 768) rounds (three resident workgroups per CU), a block costs its nk / S k-tiles plus ~6
// k-tiles' worth of prologue and epilogue.  Round 4's rule (~1024 blocks per leaf) put every layer4 launch at 1.3-1.5 rounds: a quarter to a
// third of each launch ran half empty (wgrad 3.3 ms per step for 311 GFLOP = 0.60 of the fp32 peak).
static int wgrad_splits(int64_t K, int Cin, int Cout, int taps) {
    const int64_t nk = (K + 31) / 32;
    static const int64_t target = [] { const char* e = getenv("ISX_WGRAD_TARGET"); const long long v = e ? atoll(e) : 0; return (int64_t)(v >= 64 ? v : 0); }();   // A/B knob: > 0 = the rule of rounds 4-5a (blocks per leaf)
    if (target == 0 && Cout % 128 == 0 && Cin % 128 == 0 && nk >= 8) {
        const int64_t n128 = (int64_t)(Cout / 128) * (Cin / 128) * taps;
        int best = 1;
        double best_t = 1e300;
        for (int s = 1; s <= 8 && nk / s >= 4; ++s) {
            const int64_t rounds = (n128 * 8 * s + 767) / 768;
            const double t = (double)rounds * ((double)nk / s + 6.0);
            if (t < best_t) { best_t = t; best = s; }
        }
        return best;
    }
    const bool big = Cout % 128 == 0 && Cin % 128 == 0 && (int64_t)(Cout / 128) * (Cin / 128) * taps >= 512;
    const int64_t tiles = (big ? (int64_t)(Cout / 128) * (Cin / 128) : (int64_t)(Cout / 64) * (Cin / 64)) * taps;
    const int64_t tgt = target > 0 ? target : 512;
    int64_t s = (tgt + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 16) s = 16;
    return (int)(s < 1 ? 1 : s);
}

// ---- dZ = dY . [y > 0] ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_grad_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, int64_t n4, float4* __restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = dy[i], v = y[i];
    dz[i] = make_float4(v.x > 0.0f ? g.x : 0.0f, v.y > 0.0f ? g.y : 0.0f, v.z > 0.0f ? g.z : 0.0f, v.w > 0.0f ? g.w : 0.0f);
}

// ---- input gradient of a STRIDE-2 3x3 convolution from its per-tap columns --------------------------------------------------------
// dcol[p][tap][ci] = sum_co dz[p][co] * w'[co][tap][ci] (one NT GEMM over the OUTPUT pixels: a quarter of the pixels of the zero-upsampled
// form, 9 x Cin columns); input pixel (h, w) collects the taps that reach it: (h + 1 - kh) and (w + 1 - kw) even and inside the output grid
// -- one tap for (even, even), two for mixed parity, four for (odd, odd) -- in (kh, kw) order; then the ReLU mask.  One thread = 4 channels.
__global__ __launch_bounds__(256) void col2im_s2_kernel(const float4* __restrict__ dcol, int64_t B, int H, int W, int Ho, int Wo, int C4,
                                                        const float4* __restrict__ mask, float4* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = B * H * W * C4;
    if (i >= total) return;
    const int c = (int)(i % C4);
    const int64_t pix = i / C4;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int64_t b = pix / ((int64_t)W * H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hn = h + 1 - kh;
        if (hn < 0 || (hn & 1) || (hn >> 1) >= Ho) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wn = w + 1 - kw;
            if (wn < 0 || (wn & 1) || (wn >> 1) >= Wo) continue;
            const int64_t p = (b * Ho + (hn >> 1)) * Wo + (wn >> 1);
            const float4 v = dcol[(p * 9 + kh * 3 + kw) * C4 + c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (mask) {
        const float4 m = mask[i];
        acc.x = m.x > 0.0f ? acc.x : 0.0f; acc.y = m.y > 0.0f ? acc.y : 0.0f; acc.z = m.z > 0.0f ? acc.z : 0.0f; acc.w = m.w > 0.0f ? acc.w : 0.0f;
    }
    dx[i] = acc;
}

// ---- chain rule of the BatchNorm fold -----------------------------------------------------------------------------------------
// one block per (output channel, leaf).  dwp: leaves x splits partials of (Cout, taps, Cin) [the wgrad layout], a leaf's partials added in
// split order; db: leaves x splits partials of (Cout); w: (Cout, Cin, taps) [the nn.Conv2d parameter layout]; gw / ggamma / gbeta: leaf l
// writes at + l * leaf_stride floats (the leaf's own flat gradient buffer), same layouts as the parameters
// kFoldRow: floats of one output channel's partial row a block can stage in the LDS (3x3 convolutions of up to 512 input channels)
constexpr int kFoldRow = 9 * (512 + 1);

__global__ __launch_bounds__(256) void bn_fold_backward_kernel(const float* __restrict__ dwp, int splits, const float* __restrict__ w,
                                                               const float* __restrict__ scale, const float* __restrict__ mean,
                                                               const float* __restrict__ istd, const float* __restrict__ db, int Cout, int Cin, int taps,
                                                               int accumulate, int64_t leaf_stride, float* __restrict__ gw, float* __restrict__ ggamma,
                                                               float* __restrict__ gbeta) {
    __shared__ float red[4];
    __shared__ float row[kFoldRow];
    const int co = (int)blockIdx.x, leaf = (int)blockIdx.y;
    const int K = Cin * taps;
    const float s = scale[co];
    const int64_t part = (int64_t)Cout * K;
    dwp += (int64_t)leaf * splits * part;
    db += (int64_t)leaf * splits * Cout;
    gw += leaf * leaf_stride; ggamma += leaf * leaf_stride; gbeta += leaf * leaf_stride;
    float dot = 0.0f;
    if (taps > 1 && taps * (Cin + 1) <= kFoldRow) {
        // 3x3: the partials lie (tap, ci), the parameter (ci, tap).  Round 4 read the one and scattered into the other at a stride of 9 floats
        // (0.9 ms per training step for 15 M parameters x 8 micro-batches); here the row is summed into the LDS in partial order (coalesced reads,
        // tap rows padded by one float: conflict-free transposed reads) and leaves in parameter order (coalesced w reads and gw writes).
        for (int k = threadIdx.x; k < K; k += 256) {
            float d = dwp[(int64_t)co * K + k];
            for (int sp = 1; sp < splits; ++sp) d += dwp[sp * part + (int64_t)co * K + k];
            const int tap = k / Cin, ci = k - tap * Cin;
            row[tap * (Cin + 1) + ci] = d;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K; i += 256) {
            const int ci = i / taps, tap = i - ci * taps;
            const int64_t iw = (int64_t)co * K + i;
            const float d = row[tap * (Cin + 1) + ci];
            dot += d * w[iw];
            const float g = d * s;
            gw[iw] = accumulate ? gw[iw] + g : g;
        }
    } else {
        for (int k = threadIdx.x; k < K; k += 256) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int64_t iw = (int64_t)co * K + (int64_t)ci * taps + tap;
            float d = dwp[(int64_t)co * K + k];
            for (int sp = 1; sp < splits; ++sp) d += dwp[sp * part + (int64_t)co * K + k];
            dot += d * w[iw];
            const float g = d * s;
            gw[iw] = accumulate ? gw[iw] + g : g;
        }
    }
    dot = block_sum<256>(dot, red);
    if (threadIdx.x == 0) {
        float dbs = db[co];
        for (int sp = 1; sp < splits; ++sp) dbs += db[(int64_t)sp * Cout + co];
        const float ds = dot - mean[co] * dbs;
        const float gg = ds * istd[co];
        ggamma[co] = accumulate ? ggamma[co] + gg : gg;
        gbeta[co] = accumulate ? gbeta[co] + dbs : dbs;
    }
}

}  // namespace isx

using namespace isx;

// dX = (dZ . W (+ add)) . [mask > 0]:  dz (M, Cout), wt = W^T as (Cin, Cout) row-major, add / mask (M, Cin) or NULL.
// The gradient of a 1x1 convolution wrt its input, with the ReLU of the layer below (mask = that layer's output) and the sum with the
// identity-shortcut gradient (add) fused into the GEMM's epilogue.
ISX_API int isx_conv1x1_dgrad_nhwc(const float* dz, int64_t M, int Cout, const float* wt, int Cin, const float* add, const float* mask, float* dx,
                                   isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && Cin > 0 && Cout > 0 && Cin <= (1 << 20), "isx_conv1x1_dgrad_nhwc: bad shape M=%lld Cout=%d Cin=%d", (long long)M, Cout, Cin);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(dz && wt && dx, "isx_conv1x1_dgrad_nhwc: null pointer");
    ISX_REQUIRE(dx != dz && dx != add && dx != mask, "isx_conv1x1_dgrad_nhwc: dx must not alias an input");
    return launch_gemm_masked(dz, M, wt, Cin, Cout, dx, mask, add, (hipStream_t)stream);
}

// Number of pixel splits isx_conv_wgrad_nhwc uses for this shape (host arithmetic; sizes the partial buffers).
ISX_API int isx_conv_wgrad_splits(int64_t pixels, int Cin, int Cout, int taps) {
    if (pixels < 0 || Cin <= 0 || Cout <= 0 || (taps != 1 && taps != 9) || Cin % 64 != 0 || Cout % 64 != 0) return 0;
    return wgrad_splits(pixels, Cin, Cout, taps);
}

// Weight gradient of `leaves` micro-batches in one launch.  The B images are `leaves` consecutive groups of B / leaves; for leaf l and
// split s < S = isx_conv_wgrad_splits(pixels of ONE leaf, Cin, Cout, taps):
//   dw[l][s][co][tap][ci] = sum over the output pixels p of split s of leaf l of dz[p][co] * x[src(p, tap)][ci],  db[l][s][co] = sum of dz[p][co]
// -- a leaf's partials are the same bits whether it is launched alone or with its siblings; isx_bn_fold_backward adds them in split order.
// taps = 1: a 1x1 convolution with `stride` (x: (B,H,W,Cin), dz: (B,Ho,Wo,Cout)); taps = 9: 3x3, padding 1.  dw: (leaves, S, Cout, taps, Cin) -- per
// partial the OHWI layout of the forward kernels; db: (leaves, S, Cout) or NULL.  Cout % 64 == 0, Cin % 64 == 0.
ISX_API int isx_conv_wgrad_nhwc(const float* dz, const float* x, int64_t B, int leaves, int H, int W, int Cin, int Cout, int taps, int stride, float* dw,
                                float* db, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (taps == 1 || taps == 9) && (stride == 1 || stride == 2) && leaves >= 1 && leaves <= 4096,
                "isx_conv_wgrad_nhwc: bad shape B=%lld leaves=%d H=%d W=%d Cin=%d Cout=%d taps=%d stride=%d", (long long)B, leaves, H, W, Cin, Cout, taps, stride);
    ISX_REQUIRE(B % leaves == 0, "isx_conv_wgrad_nhwc: B=%lld is not a multiple of leaves=%d", (long long)B, leaves);
    ISX_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "isx_conv_wgrad_nhwc: Cin=%d and Cout=%d must be multiples of 64", Cin, Cout);
    ISX_REQUIRE(B * H * W < (1ll << 31), "isx_conv_wgrad_nhwc: too many pixels for 32-bit pixel indices");
    ISX_REQUIRE(dw, "isx_conv_wgrad_nhwc: null pointer");
    WgradGeom g;
    g.H = H; g.W = W; g.stride = stride;
    g.Ho = (H - 1) / stride + 1;
    g.Wo = (W - 1) / stride + 1;
    g.ident = (taps == 1 && stride == 1) ? 1 : 0;
    const int64_t K = (B / leaves) * g.Ho * g.Wo;               // pixels of one leaf
    ISX_REQUIRE(K == 0 || (dz && x), "isx_conv_wgrad_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)dz | (uintptr_t)x | (uintptr_t)dw) % 16) == 0, "isx_conv_wgrad_nhwc: dz, x and dw must be 16-B aligned");
    const int64_t ldc = (int64_t)taps * Cin;
    const int S = wgrad_splits(K, Cin, Cout, taps);
    const int kt_per = (int)((((K + 31) / 32) + S - 1) / S);
    ISX_REQUIRE((int64_t)leaves * S <= 65535, "isx_conv_wgrad_nhwc: leaves x splits exceeds the grid's z extent");
    hipStream_t st = (hipStream_t)stream;
    // 128x128 tiles once the LAUNCH fills the chip twice over, 64x64 below (layer4 of ResNet-50, 2048 x 512: 64 big tiles x 4 splits for one leaf,
    // x 8 leaves in the batched step).  The tile shape only groups output elements: every element is the same k-ordered chain over the same
    // pixels of its split in either shape, so -- unlike S and the k-tile boundaries -- it may depend on how many leaves share the launch.
    const int64_t big = (int64_t)(Cout / 128) * (Cin / 128) * taps * leaves * S;
    if (Cout % 128 == 0 && Cin % 128 == 0 && big >= 512) {
        hipLaunchKernelGGL((wgrad_gemm_kernel<2, 2>), dim3((unsigned)((Cout / 128) * (Cin / 128)), (unsigned)taps, (unsigned)(leaves * S)), dim3(256), 0, st, dz, K,
                           Cout, x, Cin, g, taps, dw, ldc, Cin / 128, kt_per, S, db);
    } else {
        hipLaunchKernelGGL((wgrad_gemm_kernel<1, 1>), dim3((unsigned)((Cout / 64) * (Cin / 64)), (unsigned)taps, (unsigned)(leaves * S)), dim3(256), 0, st, dz, K,
                           Cout, x, Cin, g, taps, dw, ldc, Cin / 64, kt_per, S, db);
    }
    ISX_CHECK_LAUNCH("isx_conv_wgrad_nhwc");
    return ISX_OK;
}

// Gradient of a STRIDE-2 3x3 convolution (padding 1) wrt its input from the per-tap columns dcol = dz . w' -- (B*Ho*Wo, 9, Cin), the output of
// isx_conv1x1_dgrad_nhwc(dz, ..., wt = w' as (9 * Cin, Cout), ...) -- with the ReLU of the layer below fused: dx = gather(dcol) . [mask > 0].
// mask / dx: (B,H,W,Cin); Ho = (H - 1) / 2 + 1, Wo likewise; Cin % 4 == 0; mask may be NULL.
ISX_API int isx_conv3x3_s2_col2im_nhwc(const float* dcol, int64_t B, int H, int W, int Cin, const float* mask, float* dx, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cin % 4 == 0, "isx_conv3x3_s2_col2im_nhwc: bad shape B=%lld H=%d W=%d Cin=%d (Cin %% 4 == 0)", (long long)B, H, W, Cin);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(dcol && dx, "isx_conv3x3_s2_col2im_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)dcol | (uintptr_t)dx | (uintptr_t)mask) % 16) == 0, "isx_conv3x3_s2_col2im_nhwc: pointers must be 16-B aligned");
    const int64_t total = B * H * W * (Cin / 4);
    ISX_REQUIRE((total + 255) / 256 < (1ll << 31), "isx_conv3x3_s2_col2im_nhwc: too many elements for one grid");
    hipLaunchKernelGGL(col2im_s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)dcol, B, H, W, (H - 1) / 2 + 1,
                       (W - 1) / 2 + 1, Cin / 4, (const float4*)mask, (float4*)dx);
    ISX_CHECK_LAUNCH("isx_conv3x3_s2_col2im_nhwc");
    return ISX_OK;
}

// dz = dy . [y > 0]  (backward of the ReLU whose output is y).  dy, y, dz: n floats, n % 4 == 0, 16-B aligned; dz == dy allowed.
ISX_API int isx_relu_grad(const float* dy, const float* y, int64_t n, float* dz, isx_stream_t stream) {
    ISX_REQUIRE(n >= 0 && n % 4 == 0, "isx_relu_grad: n=%lld must be a non-negative multiple of 4", (long long)n);
    if (n == 0) return ISX_OK;
    ISX_REQUIRE(dy && y && dz, "isx_relu_grad: null pointer");
    ISX_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dz) % 16) == 0, "isx_relu_grad: pointers must be 16-B aligned");
    hipLaunchKernelGGL(relu_grad_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)y, n / 4,
                       (float4*)dz);
    ISX_CHECK_LAUNCH("isx_relu_grad");
    return ISX_OK;
}

// Gradients of (conv weight, BN gamma, BN beta) of `leaves` micro-batches from the partial gradients of the FOLDED convolution (dwp, db) that
// isx_conv_wgrad_nhwc wrote: per leaf l, with d = sum_s dwp[l][s], b = sum_s db[l][s] (split order):
// gw_l (+)= d * scale, ggamma_l (+)= (<d, w> - mean * b) * istd, gbeta_l (+)= b;  scale = gamma * istd, istd = 1 / sqrt(var + eps).
// dwp: (leaves, splits, Cout, taps, Cin); db: (leaves, splits, Cout); w: (Cout, Cin, taps) (nn.Conv2d's OIHW); gw / ggamma / gbeta: the
// gradient tensors of leaf 0, leaf l at + l * leaf_stride floats (one flat gradient buffer per leaf); accumulate != 0 adds into them.
ISX_API int isx_bn_fold_backward(const float* dwp, const float* db, int leaves, int splits, const float* w, const float* scale, const float* mean,
                                 const float* istd, int Cout, int Cin, int taps, int accumulate, int64_t leaf_stride, float* gw, float* ggamma, float* gbeta,
                                 isx_stream_t stream) {
    ISX_REQUIRE(Cout > 0 && Cin > 0 && (taps == 1 || taps == 9) && splits >= 1 && leaves >= 1 && leaves <= 65535 && leaf_stride >= 0,
                "isx_bn_fold_backward: bad shape Cout=%d Cin=%d taps=%d splits=%d leaves=%d", Cout, Cin, taps, splits, leaves);
    ISX_REQUIRE(dwp && w && scale && mean && istd && db && gw && ggamma && gbeta, "isx_bn_fold_backward: null pointer");
    hipLaunchKernelGGL(bn_fold_backward_kernel, dim3((unsigned)Cout, (unsigned)leaves), dim3(256), 0, (hipStream_t)stream, dwp, splits, w, scale, mean, istd, db, Cout,
                       Cin, taps, accumulate ? 1 : 0, leaf_stride, gw, ggamma, gbeta);
    ISX_CHECK_LAUNCH("isx_bn_fold_backward");
    return ISX_OK;
}
