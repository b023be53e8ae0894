"""The big-tile convolution kernels at the REAL layer shapes of ResNet-50 / ResNet-152 (the same 25 distinct (Cin, Cout, H, stride,
residual) shapes; ResNet-152 only repeats them: 8 / 36 blocks in stages 2 / 3), against the CPU oracle's fma chains bit for bit.

Round-2 VERDICT: the 128x128-tile + 64x64-tail dispatch was pinned against the oracle on synthetic shapes (Cin = 32 ...) and, at the
bench size, by self-consistency only.  Here every shape runs with a batch chosen so that the launch is one whole round of resident
128x128 workgroups plus a partial round -- the case in which the automatic pick takes 128x128 tiles and cuts the rows past the
round into 64x64 tiles -- and three schedules must return the same bits: the automatic pick, 128x128 forced (debug cfg 0), and the
automatic pick without tails (cfg 7).  The oracle then checks three whole images of the output: the first (first tiles), the one
whose pixels straddle the row where the 64x64 region starts, and the last (ragged end of the grid).
Reference: the torchvision ResNet trunk behind model/nn_utils.py:56-71 (extract_layers), model/siamese.py:20,107,151."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from isx import ops as o
    return o


@pytest.fixture()
def cfg():
    from isx._lib import lib
    conv, gemm = lib().isx_debug_set_conv_cfg, lib().isx_debug_set_gemm_cfg
    yield conv, gemm
    conv(-1)
    gemm(-1)


def _batch_for(hw_out, cout):
    """Images such that the launch is ~1.3 rounds of 1024 resident 128x128 workgroups (tile columns = ceil(Cout / 128)); the image
    that holds the first row of the 64x64 region."""
    tn = (cout + 127) // 128
    rows_per_round = (1024 // tn) * 128
    B = max(3, -(-int(1.3 * rows_per_round) // hw_out))
    return B, min(B - 1, rows_per_round // hw_out)


def _rand(shape, gen, scale=1.0, relu=False):
    t = torch.randn(shape, device="cuda", generator=gen) * scale
    return torch.relu_(t) if relu else t


def _nhwc(t_bhwc):
    """(B,H,W,C) contiguous CUDA tensor -> logical (B,C,H,W) view in channels-last memory (no copy)."""
    return t_bhwc.permute(0, 3, 1, 2)


def _same_bits(outs):
    ref = outs[0].view(torch.int32)
    for o in outs[1:]:
        assert torch.equal(ref, o.view(torch.int32))


# (Cin, Cout, H = W of the layer, residual)
CONV1X1 = [(64, 64, 56, False), (256, 64, 56, False), (64, 256, 56, True), (256, 128, 56, False), (512, 128, 28, False), (128, 512, 28, True),
           (512, 256, 28, False), (1024, 256, 14, False), (256, 1024, 14, True), (1024, 512, 14, False), (2048, 512, 7, False), (512, 2048, 7, True)]


@pytest.mark.parametrize("Cin,Cout,H,res", CONV1X1)
def test_conv1x1_real_layer_shapes(ops, cfg, Cin, Cout, H, res):
    conv_cfg, gemm_cfg = cfg
    B, b_split = _batch_for(H * H, Cout)
    gen = torch.Generator(device="cuda").manual_seed(Cin * 7 + Cout + H)
    x = _rand((B, H, H, Cin), gen, relu=True)
    w = _rand((Cout, Cin), gen, scale=Cin ** -0.5)
    b = _rand((Cout,), gen)
    r = _rand((B, H, H, Cout), gen) if res else None
    outs = []
    for c_conv, c_gemm in ((-1, -1), (-1, 0), (7, -1), (9, -1)):          # automatic | 128x128 forced | no tails | general path instead of the streaming kernel
        conv_cfg(c_conv); gemm_cfg(c_gemm)
        outs.append(ops.conv1x1_nhwc(_nhwc(x), w, b, _nhwc(r) if res else None, True).permute(0, 2, 3, 1).contiguous())
    conv_cfg(-1); gemm_cfg(-1)
    _same_bits(outs)
    y = outs[0]
    for i in sorted({0, b_split, B - 1}):
        want = O.conv1x1_nhwc(host(x[i]).reshape(-1, Cin), host(w), host(b), host(r[i]).reshape(-1, Cout) if res else None, True)
        np.testing.assert_array_equal(host(y[i]).reshape(-1, Cout), want)


# (Cin = Cout, H of the INPUT, stride)
CONV3X3 = [(64, 56, 1), (128, 56, 2), (128, 28, 1), (256, 28, 2), (256, 14, 1), (512, 14, 2), (512, 7, 1)]


@pytest.mark.parametrize("C,H,stride", CONV3X3)
def test_conv3x3_real_layer_shapes(ops, cfg, C, H, stride):
    conv_cfg, _ = cfg
    Ho = (H - 1) // stride + 1
    B, b_split = _batch_for(Ho * Ho, C)
    gen = torch.Generator(device="cuda").manual_seed(C + H + stride)
    x = _rand((B, H, H, C), gen, relu=True)
    w = _rand((C, 3, 3, C), gen, scale=(9 * C) ** -0.5)
    b = _rand((C,), gen)
    outs = []
    for c in (-1, 0, 7, 3):                                                # automatic | 128x128 (+ tail) forced | no tails | 64x64 everywhere
        conv_cfg(c)
        outs.append(ops.conv3x3_nhwc(_nhwc(x), w, b, stride, None, True).permute(0, 2, 3, 1).contiguous())
    conv_cfg(-1)
    _same_bits(outs)
    y = outs[0]
    for i in sorted({0, b_split, B - 1}):
        want = O.conv3x3_nhwc(host(x[i:i + 1]), host(w), host(b), stride, None, True)
        np.testing.assert_array_equal(host(y[i:i + 1]), want)


# (K1 = mid channels, K2 = block input channels, Cout, H of the block input, stride)
DUAL = [(64, 64, 256, 56, 1), (128, 256, 512, 56, 2), (256, 512, 1024, 28, 2), (512, 1024, 2048, 14, 2)]


@pytest.mark.parametrize("K1,K2,Cout,H,stride", DUAL)
def test_conv1x1_dual_real_layer_shapes(ops, cfg, K1, K2, Cout, H, stride):
    conv_cfg, _ = cfg
    Ho = (H - 1) // stride + 1
    B, b_split = _batch_for(Ho * Ho, Cout)
    gen = torch.Generator(device="cuda").manual_seed(K1 + K2 + H)
    t = _rand((B, Ho, Ho, K1), gen, relu=True)
    x = _rand((B, H, H, K2), gen, relu=True)
    w = _rand((Cout, K1 + K2), gen, scale=(K1 + K2) ** -0.5)
    b = _rand((Cout,), gen)
    outs = []
    for c in (-1, 0, 7):
        conv_cfg(c)
        outs.append(ops.conv1x1_dual_nhwc(_nhwc(t), _nhwc(x), w, b, stride, True).permute(0, 2, 3, 1).contiguous())
    conv_cfg(-1)
    _same_bits(outs)
    y = outs[0]
    for i in sorted({0, b_split, B - 1}):
        want = O.conv1x1_dual_nhwc(host(t[i:i + 1]), host(x[i:i + 1]), host(w), host(b), stride, True)
        np.testing.assert_array_equal(host(y[i:i + 1]), want)


@pytest.mark.parametrize("dual", [False, True])
def test_conv3x3_expand_real_layer_shapes(ops, dual):
    """Stage-1 bottlenecks at 56 x 56: conv2 (64 -> 64, 3x3) + conv3 (64 -> 256) + identity / projection shortcut + ReLU as ONE kernel ==
    the oracle's conv3x3 -> conv1x1 (-> dual) chain on the first, a middle and the last image of a chip-filling batch."""
    B, H = 40, 56
    gen = torch.Generator(device="cuda").manual_seed(11 + dual)
    t = _rand((B, H, H, 64), gen, relu=True)
    w2 = _rand((64, 3, 3, 64), gen, scale=(9 * 64) ** -0.5)
    b2 = _rand((64,), gen)
    b3 = _rand((256,), gen)
    if dual:
        x2 = _rand((B, H, H, 64), gen, relu=True)
        wcat = _rand((256, 128), gen, scale=128 ** -0.5)
        y = ops.conv3x3_expand_dual_nhwc(_nhwc(t), w2, b2, _nhwc(x2), wcat.t().contiguous(), b3, True).permute(0, 2, 3, 1).contiguous()
    else:
        r = _rand((B, H, H, 256), gen, relu=True)
        w3 = _rand((256, 64), gen, scale=64 ** -0.5)
        y = ops.conv3x3_expand_nhwc(_nhwc(t), w2, b2, 1, w3.t().contiguous(), b3, _nhwc(r), True).permute(0, 2, 3, 1).contiguous()
    for i in (0, B // 2, B - 1):
        mid = O.conv3x3_nhwc(host(t[i:i + 1]), host(w2), host(b2), 1, None, True)
        if dual:
            want = O.conv1x1_dual_nhwc(mid, host(x2[i:i + 1]), host(wcat), host(b3), 1, True)
        else:
            want = O.conv1x1_nhwc(mid.reshape(-1, 64), host(w3), host(b3), host(r[i]).reshape(-1, 256), True).reshape(1, H, H, 256)
        np.testing.assert_array_equal(host(y[i:i + 1]), want)
