"""The trainable trunk suffix on the hand-written kernels (isx/suffix.py, csrc/backward.hip) against torch autograd on the plain
modules (fp32, MIOpen): outputs and every parameter gradient.  Floating-point kernels: the reference here is torch fp32, the tolerance
is stated per check (summation order differs: k-ordered fp32 MFMA chains vs MIOpen's blocked sums)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _randomise_bn(mods, seed):
    g = torch.Generator().manual_seed(seed)
    for m in mods.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))


def _blocks(which):
    from isx import backbones
    net = backbones.resnet50(pretrained=True, seed=0)
    if which == "layer4":
        blocks, cin, hw = list(net.layer4), 1024, 14
    else:                                                    # last block of layer3 + the projection block of layer4: a projection shortcut above the first block
        blocks, cin, hw = [net.layer3[-1], net.layer4[0]], 1024, 14
    seq = nn.Sequential(*blocks).cuda()
    _randomise_bn(seq, 3)
    seq.train()
    for m in seq.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    return seq, cin, hw


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def _ref64_with_masks(seq64, x64, masks):
    """The bottleneck stack in float64 with every ReLU replaced by a multiplication with the GIVEN 0/1 mask (the engine's own activation
    pattern): a gradient comparison must not depend on which side of 0 a pre-activation of size 1e-7 falls after rounding -- one flipped
    unit moves a bias gradient by percents."""
    F = torch.nn.functional

    def cb(x, conv, bn):
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
        s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        return y * s.view(1, -1, 1, 1) + (bn.bias - bn.running_mean * s).view(1, -1, 1, 1)

    x = x64
    for blk, (m1, m2, m3) in zip(seq64, masks):
        t1 = cb(x, blk.conv1, blk.bn1) * m1
        t2 = cb(t1, blk.conv2, blk.bn2) * m2
        idt = x if blk.downsample is None else cb(x, blk.downsample[0], blk.downsample[1])
        x = (cb(t2, blk.conv3, blk.bn3) + idt) * m3
    return x


@pytest.mark.parametrize("which,B", [("layer4", 3), ("layer4", 24), ("layer3+4", 8)])
def test_suffix_engine_matches_float64_autograd(which, B):
    """Arbiter: torch autograd in FLOAT64 on the plain convolution + BatchNorm formulas, with the engine's own ReLU pattern (see
    _ref64_with_masks).  Output and every parameter gradient of the fp32 engine within 1e-5 of it (relative to the tensor's largest entry;
    the sums are k-ordered fp32 chains over up to 4704 pixels / 4608 channels)."""
    from isx.suffix import SuffixEngine
    seq, cin, hw = _blocks(which)
    ref64 = copy.deepcopy(seq).double()
    assert SuffixEngine.applicable(list(seq))
    eng = SuffixEngine(list(seq))
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.relu(torch.randn(B, cin, hw, hw, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    y = eng(x)
    assert y.is_contiguous(memory_format=torch.channels_last)
    _, saved = eng.forward(x)                                  # (x, t1, t2, y) per block, contiguous (B,H,W,C)
    masks = [tuple((t > 0).permute(0, 3, 1, 2).double() for t in (t1, t2, yb)) for _, t1, t2, yb in saved]
    y64 = _ref64_with_masks(ref64, x.double(), masks)
    assert y.shape == y64.shape and _rel(y.double(), y64) <= 1e-5
    # and against the plain modules in fp32 (MIOpen), forward only (the ReLU pattern may differ in a handful of units)
    with torch.no_grad():
        assert _rel(y, seq(x)) <= 2e-5
    r = torch.randn(y.shape, device="cuda", generator=g)
    (y * r).sum().backward()
    (y64 * r.double()).sum().backward()
    worst = 0.0
    for (n, p), (_, q) in zip(seq.named_parameters(), ref64.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, n
        e = _rel(p.grad.double(), q.grad)
        worst = max(worst, e)
        assert e <= 1e-5, (n, e)
    print("suffix engine %s B=%d: max relative deviation from float64 -- gradients %.2e, output %.2e" % (which, B, worst, _rel(y.double(), y64)))
    # a second micro-batch ACCUMULATES in place into the existing .grad tensors (no new tensors, no autograd add pass)
    ptrs = [p.grad.data_ptr() for p in seq.parameters()]
    first = [p.grad.clone() for p in seq.parameters()]
    (eng(x) * r).sum().backward()
    for p, g0, ptr in zip(seq.parameters(), first, ptrs):
        assert p.grad.data_ptr() == ptr
        assert _rel(p.grad, 2 * g0) <= 1e-6


def test_post_accumulate_hooks_fire_for_gradients_accumulated_in_place():
    """A parameter that already holds a .grad gets its new gradient added in place by isx_bn_fold_backward and autograd is handed None for it;
    the engine still visits the parameter's AccumulateGrad node, so the hooks registered behind it (dp.GradAllReducer's bucket counters in the
    all-reduce gradient exchange) run exactly once per backward either way (round-4 ADVICE assumed they did not)."""
    from isx.suffix import SuffixEngine
    seq, cin, hw = _blocks("layer4")
    eng = SuffixEngine(list(seq))
    x = torch.relu(torch.randn(2, cin, hw, hw, device="cuda")).contiguous(memory_format=torch.channels_last)
    seen = []
    params = [p for p in seq.parameters() if p.requires_grad]
    for p in params:
        p.register_post_accumulate_grad_hook(lambda q: seen.append(id(q)))
    eng(x).sum().backward()                                  # first backward: fresh tensors handed to autograd, its own AccumulateGrad calls the hooks
    assert sorted(seen) == sorted(id(p) for p in params)
    del seen[:]
    eng(x).sum().backward()                                  # second: in place
    assert sorted(seen) == sorted(id(p) for p in params)


def test_suffix_engine_follows_the_optimizer():
    """The folded weights are derived per step from the parameters' version counters: after an optimizer step the engine computes with the new
    weights (same output as the plain modules), and a BatchNorm put into training mode makes it inapplicable."""
    from isx.suffix import SuffixEngine
    seq, cin, hw = _blocks("layer4")
    eng = SuffixEngine(list(seq))
    x = torch.relu(torch.randn(2, cin, hw, hw, device="cuda")).contiguous(memory_format=torch.channels_last)
    opt = torch.optim.SGD(seq.parameters(), lr=0.05)
    y0 = eng(x).detach().clone()
    eng(x).square().mean().backward()
    opt.step()
    y1 = eng(x)
    assert not torch.equal(y0, y1)
    with torch.no_grad():
        assert _rel(y1, seq(x)) <= 2e-5
    seq[0].bn1.train()
    assert not SuffixEngine.applicable(list(seq))


def test_backward_kernels_vs_torch():
    """The individual entry points of csrc/backward.hip on small ragged shapes against torch fp32."""
    from isx._lib import check, lib
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    # relu_grad
    M, C = 1177, 200
    dy, y = rn(M, C), rn(M, C)
    dz = torch.empty_like(dy)
    check(L.isx_relu_grad(dy.data_ptr(), y.data_ptr(), M * C, dz.data_ptr(), st), "x")
    assert torch.equal(dz, dy * (y > 0))
    # 1x1 dgrad with add + mask
    M, Co, Ci = 333, 192, 128
    dzz, w, add, mask = rn(M, Co), rn(Co, Ci), rn(M, Ci), rn(M, Ci)
    dx = torch.empty(M, Ci, device="cuda")
    check(L.isx_conv1x1_dgrad_nhwc(dzz.data_ptr(), M, Co, w.t().contiguous().data_ptr(), Ci, add.data_ptr(), mask.data_ptr(), dx.data_ptr(), st), "x")
    want = (dzz.double() @ w.double() + add.double()) * (mask > 0)
    np.testing.assert_allclose(dx.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-3)
    check(L.isx_conv1x1_dgrad_nhwc(dzz.data_ptr(), M, Co, w.t().contiguous().data_ptr(), Ci, None, None, dx.data_ptr(), st), "x")
    np.testing.assert_allclose(dx.cpu().numpy(), (dzz.double() @ w.double()).float().cpu().numpy(), rtol=1e-4, atol=1e-3)
    # wgrad (+ bias gradient): 1x1, strided 1x1, 3x3 stride 1 and 2 against torch's convolution_backward; pixel counts that give 1 and several splits
    for taps, stride, B, H, W, Ci, Co in ((1, 1, 3, 5, 7, 64, 128), (1, 2, 2, 7, 6, 128, 64), (9, 1, 2, 6, 5, 64, 64), (9, 2, 3, 7, 7, 128, 64),
                                          (1, 1, 24, 7, 7, 128, 64), (9, 1, 8, 14, 14, 64, 64), (9, 2, 2, 14, 14, 64, 128)):
        k = 3 if taps == 9 else 1
        x = rn(B, Ci, H, W).contiguous(memory_format=torch.channels_last)
        wt = rn(Co, Ci, k, k)
        out = torch.nn.functional.conv2d(x, wt, None, stride, k // 2)
        dz_ = rn(*out.shape).contiguous(memory_format=torch.channels_last)
        _, gw, _ = torch.ops.aten.convolution_backward(dz_, x, wt, None, [stride, stride], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, False])
        S = L.isx_conv_wgrad_splits(out.shape[0] * out.shape[2] * out.shape[3], Ci, Co, taps)
        assert S >= 1
        dw, dbp = torch.empty(S, Co, taps, Ci, device="cuda"), torch.empty(S, Co, device="cuda")
        dzc = dz_.permute(0, 2, 3, 1).contiguous()
        xc = x.permute(0, 2, 3, 1).contiguous()
        check(L.isx_conv_wgrad_nhwc(dzc.data_ptr(), xc.data_ptr(), B, 1, H, W, Ci, Co, taps, stride, dw.data_ptr(), dbp.data_ptr(), st), "x")
        got = dw.sum(0).view(Co, k, k, Ci).permute(0, 3, 1, 2)
        if B % 2 == 0:                                           # two leaves in one launch: each leaf's partials are the bits of a launch of its own
            S2 = L.isx_conv_wgrad_splits(out.shape[0] // 2 * out.shape[2] * out.shape[3], Ci, Co, taps)
            both, dbb2 = torch.empty(2, S2, Co, taps, Ci, device="cuda"), torch.empty(2, S2, Co, device="cuda")
            check(L.isx_conv_wgrad_nhwc(dzc.data_ptr(), xc.data_ptr(), B, 2, H, W, Ci, Co, taps, stride, both.data_ptr(), dbb2.data_ptr(), st), "x")
            for l in range(2):
                one, db1 = torch.empty(1, S2, Co, taps, Ci, device="cuda"), torch.empty(1, S2, Co, device="cuda")
                h = B // 2
                check(L.isx_conv_wgrad_nhwc(dzc[l * h:(l + 1) * h].contiguous().data_ptr(), xc[l * h:(l + 1) * h].contiguous().data_ptr(), h, 1, H, W, Ci, Co, taps,
                                            stride, one.data_ptr(), db1.data_ptr(), st), "x")
                assert torch.equal(one[0], both[l]) and torch.equal(db1[0], dbb2[l]), (taps, stride, l)
            assert _rel(both.sum((0, 1)).view(Co, k, k, Ci).permute(0, 3, 1, 2), gw) <= 2e-5
        assert _rel(got, gw) <= 2e-5, (taps, stride, S, _rel(got, gw))
        assert _rel(dbp.sum(0).double(), dzc.double().sum((0, 1, 2))) <= 1e-5, (taps, stride, S)
        if taps == 9:                                           # 3x3 dgrad (with mask) against torch
            gx, _, _ = torch.ops.aten.convolution_backward(dz_, x, wt, None, [stride, stride], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
            d = dz_.permute(0, 2, 3, 1).contiguous()
            if stride == 2:
                up = torch.zeros(B, H, W, Co, device="cuda")
                up[:, ::2, ::2] = d
                d = up
            wd = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()
            m = rn(B, H, W, Ci)
            dxx = torch.empty(B, H, W, Ci, device="cuda")
            check(L.isx_conv3x3_dgrad_nhwc(d.data_ptr(), B, H, W, Co, wd.data_ptr(), Ci, m.data_ptr(), dxx.data_ptr(), st), "x")
            want = gx.permute(0, 2, 3, 1) * (m > 0)
            assert _rel(dxx, want) <= 2e-5, (stride, _rel(dxx, want))
            if stride == 2:                                     # the product path of a stride-2 layer: per-tap columns (one GEMM over the output pixels) + tap gather
                Mo = out.shape[0] * out.shape[2] * out.shape[3]
                wcol = wt.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).t().contiguous()          # (9 Ci, Co)
                dcol = torch.empty(Mo, 9 * Ci, device="cuda")
                check(L.isx_conv1x1_dgrad_nhwc(dz_.permute(0, 2, 3, 1).contiguous().data_ptr(), Mo, Co, wcol.data_ptr(), 9 * Ci, None, None, dcol.data_ptr(), st), "x")
                dx2 = torch.empty(B, H, W, Ci, device="cuda")
                check(L.isx_conv3x3_s2_col2im_nhwc(dcol.data_ptr(), B, H, W, Ci, m.data_ptr(), dx2.data_ptr(), st), "x")
                assert _rel(dx2, want) <= 2e-5, _rel(dx2, want)
                check(L.isx_conv3x3_s2_col2im_nhwc(dcol.data_ptr(), B, H, W, Ci, None, dx2.data_ptr(), st), "x")
                assert _rel(dx2, gx.permute(0, 2, 3, 1)) <= 2e-5
    # chain rule of the fold against autograd through the fold itself
    Co, Ci, taps = 64, 128, 9
    w = rn(Co, Ci, 3, 3).requires_grad_()
    gam, bet = (0.5 + torch.rand(Co, device="cuda", generator=g)).requires_grad_(), rn(Co).requires_grad_()
    mean, var = rn(Co), 0.5 + torch.rand(Co, device="cuda", generator=g)
    istd = torch.rsqrt(var + 1e-5)
    s = gam * istd
    wf, bf = w * s.view(-1, 1, 1, 1), bet - mean * s
    dwp, dbb = rn(Co, 3, 3, Ci), rn(Co)                          # gradient of the folded weight in OHWI, of the folded bias
    ((wf.permute(0, 2, 3, 1) * dwp).sum() + (bf * dbb).sum()).backward()
    gw, gg, gb = torch.ones_like(w), torch.ones_like(gam), torch.ones_like(bet)      # accumulate into ones
    parts = torch.stack([0.25 * dwp, 0.5 * dwp, 0.25 * dwp]).contiguous()             # three partials that add up to dwp (exactly: powers of two)
    dparts = torch.stack([0.5 * dbb, 0.25 * dbb, 0.25 * dbb]).contiguous()
    check(L.isx_bn_fold_backward(parts.data_ptr(), dparts.data_ptr(), 1, 3, w.detach().data_ptr(), s.detach().contiguous().data_ptr(), mean.data_ptr(),
                                 istd.data_ptr(), Co, Ci, taps, 1, 0, gw.data_ptr(), gg.data_ptr(), gb.data_ptr(), st), "x")
    assert _rel(gw - 1, w.grad) <= 1e-5 and _rel(gg - 1, gam.grad) <= 1e-4 and _rel(gb - 1, bet.grad) <= 1e-6


@pytest.mark.parametrize("B,D", [(24, 100352), (24, 2048), (3, 17), (1, 5000), (70, 1025)])
def test_l2norm_rows_bwd_vs_float64_autograd(B, D):
    """isx_l2norm_rows_bwd (reference model/custom_modules.py:59-67) against torch autograd of the same formula in float64; and the
    NormalizeL2 module routes its backward through it on the GPU."""
    from isx import ops
    from model.custom_modules import NormalizeL2
    g = torch.Generator(device="cuda").manual_seed(B * 7 + D)
    x = torch.randn(B, D, device="cuda", generator=g)
    dy = torch.randn(B, D, device="cuda", generator=g)
    x64 = x.double().requires_grad_()
    (x64 / (x64.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()).backward(dy.double())
    got = ops.l2norm_rows_bwd(x, dy, 1e-10)
    assert _rel(got.double(), x64.grad) <= 2e-6
    xm = x.clone().requires_grad_()
    NormalizeL2()(xm).backward(dy)
    assert torch.equal(xm.grad, got)
