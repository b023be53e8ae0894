"""Round 3: the region path (BASELINE configs[2], reference train/classif_regions.py:107-132, model/siamese.py:64-89, 185-223) on the
channels-last trunk without a transpose and without MIOpen: the fused stem on images wider than 224 (column bands inside the workgroup),
the NHWC box pooling, the 1x1-convolution classifier as a libisx GEMM, best-location / top-k selection and the window gather on
channels-last maps.  Everything against the CPU oracle on the same inputs (selection, indices, pooled sums, convolution chains: bit-exact)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-6, atol=2e-7)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def cl(a_nchw):
    """numpy (B,C,H,W) -> channels-last CUDA tensor with the same logical shape."""
    return dev(a_nchw).contiguous(memory_format=torch.channels_last)


@pytest.fixture(scope="module")
def ops():
    from isx import ops as o
    return o


@pytest.mark.parametrize("B,H,W", [(2, 448, 448), (1, 448, 300), (3, 384, 512), (1, 40, 228), (2, 64, 452), (1, 33, 896), (1, 100, 676), (260, 16, 232),
                                   (1, 448, 224), (2, 21, 448)])
def test_stem7x7_pool_wide(ops, B, H, W):
    """Images wider than one 224-column band: units (row step, column band) inside one workgroup, carry rows per band in registers, the
    right-most pooled partial column handed to the next band through LDS -- bit-exact against the oracle; W = 228 / 452 / 676 leave a band
    of ONE convolution column pair, 896 is the widest supported, 224-wide goes through the single-band instantiation."""
    rng = np.random.default_rng(H * 1000 + W + B)
    x = rng.standard_normal((B, H, W, 3), dtype=np.float32)
    w = rng.standard_normal((64, 7, 7, 3), dtype=np.float32) * np.float32(147 ** -0.5)
    b = rng.standard_normal(64, dtype=np.float32)
    xt = dev(x).permute(0, 3, 1, 2)
    got = host(ops.stem7x7_pool(xt, dev(w), dev(b)).permute(0, 2, 3, 1))
    want = O.stem7x7_pool_nhwc(x, w, b)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)


def test_stem_applicable_up_to_896(ops):
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3)
    for W, ok in ((448, True), (896, True), (900, False), (450, False)):
        x = torch.empty(1, 3, 8, W, device="cuda").contiguous(memory_format=torch.channels_last)
        assert ops.stem7x7_pool_applicable(x, conv) == ok


@pytest.mark.parametrize("B,C,H,W,kh,kw", [(2, 2048, 14, 14, 7, 7), (1, 16, 7, 5, 3, 3), (1, 256, 13, 13, 6, 6), (3, 4, 7, 7, 7, 7), (2, 72, 20, 30, 7, 7),
                                           (1, 2048, 12, 16, 7, 7)])
def test_boxpool_s1_nhwc(ops, B, C, H, W, kh, kw):
    rng = np.random.default_rng(H * W + C)
    f = rng.standard_normal((B, C, H, W), dtype=np.float32)
    got = ops.boxpool_s1_nhwc(cl(f), kh, kw)
    assert got.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_array_equal(host(got), O.boxpool_s1(f, kh, kw))          # same in-order window sums as the NCHW kernel and torch CPU
    np.testing.assert_array_equal(host(got), host(ops.boxpool_s1(dev(f), kh, kw)))


def test_best_location_nhwc(ops, golden):
    g = golden("best_location.npz")
    for t in list(range(4)) + ["_r"]:
        m = g["map%s" % t] if t != "_r" else g["map_r"]
        d, loc = ops.best_location_desc(cl(m[None]))
        want_loc = g["locs"][t] if t != "_r" else g["loc_r"]
        assert tuple(host(loc)[0]) == tuple(want_loc)
        np.testing.assert_allclose(host(d)[0], g["desc%s" % t] if t != "_r" else g["desc_r"], **TOL)
    rng = np.random.default_rng(1)
    cls = rng.standard_normal((6, 464, 8, 8), dtype=np.float32)
    cls[2, :, 5, 2] = cls[2, :, 1, 6]            # tie between two locations: smallest column wins
    cls[2, 7, 5, 2] = cls[2, 7, 1, 6] = 50.0
    cls[4, :, 3, 3] = cls[4, :, 6, 3]            # tie inside one column: smallest row wins
    cls[4, 9, 3, 3] = cls[4, 9, 6, 3] = 60.0
    d, loc = ops.best_location_desc(cl(cls))
    d0, loc0 = ops.best_location_desc(dev(cls))
    assert torch.equal(loc, loc0) and torch.equal(d, d0)                        # same reduction order as the NCHW kernel
    for b in range(6):
        od, ol = O.best_location_desc(cls[b])
        assert tuple(host(loc)[b]) == tuple(ol)
        np.testing.assert_allclose(host(d)[b], od, **TOL)
    assert tuple(host(loc)[2]) == (5, 2) and tuple(host(loc)[4]) == (3, 3)


@pytest.mark.parametrize("B,K,Hp,Wp,k", [(3, 9, 5, 3, 3), (2, 9, 5, 3, 40), (4, 464, 8, 8, 6), (1, 17, 1, 2, 6), (2, 5, 60, 60, 10), (2, 70, 9, 4, 5)])
def test_region_topk_and_gather_nhwc(ops, B, K, Hp, Wp, k):
    rng = np.random.default_rng(K + Hp)
    cls = rng.standard_normal((B, K, Hp, Wp), dtype=np.float32)
    if Hp * Wp > 4:
        cls[:, :, 1, 1] = cls[:, :, 0, 0]        # tie: the smaller flat index ranks first
    idx, sc = ops.region_topk(cl(cls), k)
    C, fs = 12, 3
    fmap = rng.standard_normal((B, C, Hp + fs - 1, Wp + fs - 1), dtype=np.float32)
    sh = rng.standard_normal((C * fs * fs,), dtype=np.float32) * 0.05
    sh_hwc = np.ascontiguousarray(sh.reshape(C, fs, fs).transpose(1, 2, 0)).reshape(-1)
    rows = host(ops.region_gather_l2_nhwc(cl(fmap), fs, fs, idx, Wp, dev(sh_hwc)))
    for b in range(B):
        oi, osc = O.region_topk(cls[b], k)
        n = len(oi)
        np.testing.assert_array_equal(host(idx)[b, :n], oi)
        np.testing.assert_array_equal(host(sc)[b, :n], osc)
        assert (host(idx)[b, n:] == -1).all()
        want = O.region_gather_l2(fmap[b], fs, fs, oi, Wp, sh)                 # (n, C*fs*fs) in (C,h,w) order
        want_hwc = want.reshape(n, C, fs, fs).transpose(0, 2, 3, 1).reshape(n, -1)
        np.testing.assert_allclose(rows[b, :n], want_hwc, rtol=2e-6, atol=1e-7)
        assert (rows[b, n:] == 0).all()


def _sub_net(n_cls=464):
    from isx import backbones
    from model.nn_utils import fold_batch_norm
    from model.siamese import TuneClassifSub
    torch.manual_seed(0)
    net = TuneClassifSub(backbones.resnet50(pretrained=True, seed=0), n_cls, (7, 7)).eval()
    net.features = fold_batch_norm(net.features)
    return net.cuda().to(memory_format=torch.channels_last)


def test_classif_regions_tail_is_channels_last_and_exact(ops):
    """TuneClassifSub behind the NHWC trunk at 448 x 448: box pooling, the 1x1 classifier (libisx GEMM) and the best-location descriptor never
    leave channels-last memory, and every stage equals the oracle on the trunk's own feature map (pooled sums and the classifier's fma chains
    bit for bit, descriptor within the L2 summation slack)."""
    from train import classif_regions as cr
    net = _sub_net(120)
    x = torch.randn(3, 3, 448, 448, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        fmap = net.features(x)
        pooled = net.feature_reduc(fmap)
        score = net.classifier(pooled)
        desc = cr._best_location_descriptors(score)
    for t in (fmap, pooled, score):
        assert t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
    assert tuple(score.shape) == (3, 120, 8, 8)
    f = host(fmap)
    want_pooled = O.boxpool_s1(f, 7, 7)
    np.testing.assert_array_equal(host(pooled), want_pooled)
    conv = net.classifier[0]
    w, b = host(conv.weight).reshape(120, -1), host(conv.bias)
    px = np.ascontiguousarray(want_pooled.transpose(0, 2, 3, 1)).reshape(-1, 2048)
    want_score = O.conv1x1_nhwc(px, w, b, None, False).reshape(3, 8, 8, 120).transpose(0, 3, 1, 2)
    np.testing.assert_array_equal(host(score), want_score)
    for i in range(3):
        od, _ = O.best_location_desc(want_score[i])
        np.testing.assert_allclose(host(desc)[i], od, **TOL)


def test_region_path_448_is_batch_independent(ops):
    """B = 128 images of 448 x 448 in one launch == 8 launches of 16, bit for bit (fused wide stem with its row/column bands, tile picks and
    64x64 tails of every convolution at 4x the pixels of a 224 batch, NHWC region tail): the descriptor of an image does not depend on the
    batch it rides in."""
    from train import classif_regions as cr
    net = _sub_net(464)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(128, 3, 448, 448, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        big = cr._best_location_descriptors(net(x)[0])
        small = torch.cat([cr._best_location_descriptors(net(x[i:i + 16])[0]) for i in range(0, 128, 16)], 0)
    assert torch.equal(big.view(torch.int32), small.view(torch.int32))
    assert torch.isfinite(big).all() and float((big.norm(dim=1) - 1).abs().max()) < 1e-5


def test_region_descriptor_net_channels_last_matches_nchw_path(ops):
    """RegionDescriptorNet on the channels-last trunk (windows gathered as (h,w,C) runs against the permuted Shift / Linear weight) against
    the same net fed through the NCHW kernels: identical windows, descriptors equal within fp32 summation order (1e-5)."""
    from isx import backbones
    from model.nn_utils import fold_batch_norm
    from model.siamese import RegionDescriptorNet
    torch.manual_seed(0)
    net = RegionDescriptorNet(backbones.resnet18(pretrained=True, seed=0), 4, 96, (7, 7)).eval()
    with torch.no_grad():
        net.feature_reduc1[1].param.normal_(0, 0.01)
    net.features = fold_batch_norm(net.features)
    net = net.cuda().to(memory_format=torch.channels_last)
    x = torch.randn(5, 3, 320, 352, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        fmap = net.features(x)
        c = net.classifier(net.feature_reduc(fmap))
        assert ops._is_nhwc(fmap) and ops._is_nhwc(c)
        d_cl, cls_cl = net._batched_gpu(fmap, c)
        d_nc, cls_nc = net._batched_gpu(fmap.contiguous(), c.contiguous())
    assert torch.equal(cls_cl, cls_nc)                                            # same windows picked
    assert float((d_cl - d_nc).abs().max()) < 1e-5
    # a weight written in place after the first forward is picked up (the permuted copies are keyed on the version counters)
    with torch.no_grad():
        net.feature_reduc1[2].weight.mul_(0.5)
        net.feature_reduc1[2].bias.zero_()
        d2, _ = net._batched_gpu(fmap, c)
        d2_nc, _ = net._batched_gpu(fmap.contiguous(), c.contiguous())
    assert float((d2 - d2_nc).abs().max()) < 1e-5
