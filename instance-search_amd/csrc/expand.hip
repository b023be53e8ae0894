// expand.hip -- conv2 + conv3 (+ projection) of the 64-channel bottlenecks as ONE kernel (isx_conv3x3_expand_nhwc, isx_conv3x3_expand_dual_nhwc).
// A translation unit of its own: conv.hip and cosine.hip are compiled with the max-ILP scheduling strategy, which suits their plain tile loops
// (-0.3 ms each on the bench) and costs this kernel 4 % (Makefile).
// Reference call sites: the torchvision ResNet `features` trunk built by model/ModelDefinition.py, split by model/nn_utils.py:56-71 and run from
// model/siamese.py:20,107,151.
#include <stdlib.h>

#include "expand_kernel.hpp"

using namespace isx;

// 3x3 convolution (padding 1, stride 1 or 2) to 64 channels + ReLU, then the 1x1 expansion to 256 channels + bias (+ residual) (+ ReLU), one kernel
// (see conv3x3_expand_kernel).  w2_ohwi: (64,3,3,Cin); w3t: (64, 256) = the expansion weight TRANSPOSED; y / residual: (B,Ho,Wo,256).
ISX_API int isx_conv3x3_expand_nhwc(const float* x, int64_t B, int H, int W, int Cin, const float* w2_ohwi, const float* b2, int stride,
                                    const float* w3t, int Cout, const float* b3, const float* residual, int relu, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && (stride == 1 || stride == 2),
                "isx_conv3x3_expand_nhwc: bad shape B=%lld H=%d W=%d Cin=%d stride=%d", (long long)B, H, W, Cin, stride);
    ISX_REQUIRE(Cin % 32 == 0, "isx_conv3x3_expand_nhwc: Cin=%d must be a multiple of 32", Cin);
    ISX_REQUIRE(Cout == 256, "isx_conv3x3_expand_nhwc: Cout=%d (this kernel is built for 64 mid channels -> 256)", Cout);
    ISX_REQUIRE(H < 32767 && W < 32767 && B * H * W < (1ll << 31), "isx_conv3x3_expand_nhwc: input has too many pixels for 32-bit pixel indices");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(x && w2_ohwi && b2 && w3t && b3 && y, "isx_conv3x3_expand_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)x | (uintptr_t)w2_ohwi | (uintptr_t)w3t) % 16) == 0, "isx_conv3x3_expand_nhwc: x, w2 and w3t must be 16-B aligned");
    ISX_REQUIRE(y != x && y != residual, "isx_conv3x3_expand_nhwc: y must not alias x or residual");
    Conv3x3Geom g;
    g.H = H; g.W = W; g.Cin = Cin; g.stride = stride;
    g.Ho = (H - 1) / stride + 1;
    g.Wo = (W - 1) / stride + 1;
    const int64_t M = B * g.Ho * g.Wo;
    ISX_REQUIRE((M + 63) / 64 < (1ll << 31), "isx_conv3x3_expand_nhwc: too many tiles");
    // ISX_EXPAND_TILE=256: the 256-pixel variant (four independent accumulators per wave in the 3x3 loop, round 4).  Bit-identical results; measured
    // 5 % SLOWER on the two identity-shortcut launches of the bench step (10.38 vs 10.02 ms for the family): at two workgroups per CU the HBM-heavy
    // epilogue (7.4 GB per launch) finds too few waves to hide behind, and the family already sits at busy x clock = 0.84 x 2.17 / 2.4 = 0.76 of nominal.
    static const int forced = [] { const char* e = getenv("ISX_EXPAND_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 256) {
        hipLaunchKernelGGL((conv3x3_expand256_kernel<4>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M, w2_ohwi, g, b2, w3t, b3, residual,
                           relu ? 1 : 0, y);
        ISX_CHECK_LAUNCH("isx_conv3x3_expand_nhwc");
        return ISX_OK;
    }
    hipLaunchKernelGGL((conv3x3_expand_kernel<4, false>), dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, M, w2_ohwi, g, b2, w3t, b3, residual,
                       relu ? 1 : 0, y);
    ISX_CHECK_LAUNCH("isx_conv3x3_expand_nhwc");
    return ISX_OK;
}

// The same with a 1x1 PROJECTION shortcut of the 64-channel block input x2 (first block of the stage; stride 1 only):
//   y = act( [W3 | Wd] . [relu(conv3x3(t, W2) + b2) ; x2] + bias ).  wcat_t: (128, 256) = [W3 | Wd] transposed; x2: (B,H,W,64).
ISX_API int isx_conv3x3_expand_dual_nhwc(const float* t, int64_t B, int H, int W, int Cin, const float* w2_ohwi, const float* b2, const float* x2,
                                         const float* wcat_t, int Cout, const float* bias, int relu, float* y, isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0, "isx_conv3x3_expand_dual_nhwc: bad shape B=%lld H=%d W=%d Cin=%d", (long long)B, H, W, Cin);
    ISX_REQUIRE(Cin % 32 == 0, "isx_conv3x3_expand_dual_nhwc: Cin=%d must be a multiple of 32", Cin);
    ISX_REQUIRE(Cout == 256, "isx_conv3x3_expand_dual_nhwc: Cout=%d (this kernel is built for 64 mid channels -> 256)", Cout);
    ISX_REQUIRE(H < 32767 && W < 32767 && B * H * W < (1ll << 31), "isx_conv3x3_expand_dual_nhwc: input has too many pixels for 32-bit pixel indices");
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(t && w2_ohwi && b2 && x2 && wcat_t && bias && y, "isx_conv3x3_expand_dual_nhwc: null pointer");
    ISX_REQUIRE((((uintptr_t)t | (uintptr_t)w2_ohwi | (uintptr_t)wcat_t | (uintptr_t)x2) % 16) == 0, "isx_conv3x3_expand_dual_nhwc: t, x2, w2 and wcat_t must be 16-B aligned");
    ISX_REQUIRE(y != t && y != x2, "isx_conv3x3_expand_dual_nhwc: y must not alias an input");
    Conv3x3Geom g;
    g.H = H; g.W = W; g.Cin = Cin; g.stride = 1; g.Ho = H; g.Wo = W;
    const int64_t M = B * H * W;
    hipLaunchKernelGGL((conv3x3_expand_kernel<4, true>), dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, t, M, w2_ohwi, g, b2, wcat_t, bias, x2,
                       relu ? 1 : 0, y);
    ISX_CHECK_LAUNCH("isx_conv3x3_expand_dual_nhwc");
    return ISX_OK;
}
