"""The descriptor head for all micro-batches of a training step at once (isx/head.py, csrc/head.hip): floating-point kernels against
float64 torch, and the property the canonical gradient tree needs -- a row's result does not depend on how many rows share the pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


def _lib():
    from isx._lib import check, lib
    return lib(), check, torch.cuda.current_stream().cuda_stream


def _fwd(x, w, b):
    L, check, st = _lib()
    M, K = x.shape
    N = w.shape[0]
    Mp = (M + 63) // 64 * 64
    xT = x.new_zeros((K, Mp))
    xT[:, :M] = x.t()
    S = L.isx_head_linear_splits(K)
    ws = torch.empty(S * Mp * N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    check(L.isx_head_linear_fwd(xT.data_ptr(), M, Mp, K, w.data_ptr(), N, b.data_ptr() if b is not None else None, y.data_ptr(), ws.data_ptr(), ws.numel() * 4, st), "x")
    return y


def _dgrad(dy, w):
    L, check, st = _lib()
    M, N = dy.shape
    K = w.shape[1]
    Mp = (M + 63) // 64 * 64
    dyT = dy.new_zeros((N, Mp))
    dyT[:, :M] = dy.t()
    dx = torch.empty(Mp, K, device="cuda")
    check(L.isx_head_linear_dgrad(dyT.data_ptr(), Mp, N, w.data_ptr(), K, dx.data_ptr(), st), "x")
    return dx[:M]


@pytest.mark.parametrize("K,N", [(100352, 2048), (4096, 64), (640, 192)])
def test_head_linear_forward_and_input_gradient(K, N):
    g = torch.Generator(device="cuda").manual_seed(K + N)
    M = 192
    x = torch.randn(M, K, device="cuda", generator=g) * 0.01
    w = torch.randn(N, K, device="cuda", generator=g) * 0.01
    b = torch.randn(N, device="cuda", generator=g)
    y = _fwd(x, w, b)
    assert _rel(y, x.double() @ w.double().t() + b.double()) <= 2e-6
    assert _rel(_fwd(x, w, None), x.double() @ w.double().t()) <= 2e-6
    dy = torch.randn(M, N, device="cuda", generator=g)
    dx = _dgrad(dy, w)
    assert _rel(dx, dy.double() @ w.double()) <= 5e-6                 # one k-ordered fp32 chain over N = 2048 terms per output
    # the rows of ONE micro-batch (24 rows, padded to 64 columns) and of other row counts: the same bits as inside the 192-row pass
    for lo, hi in ((0, 24), (24, 48), (168, 192), (0, 64), (48, 176), (7, 8)):
        assert torch.equal(_fwd(x[lo:hi].contiguous(), w, b), y[lo:hi]), (lo, hi)
        assert torch.equal(_dgrad(dy[lo:hi].contiguous(), w), dx[lo:hi]), (lo, hi)


def test_colsum_leaves():
    L, check, st = _lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    for leaves, R, C in ((8, 24, 100352), (1, 24, 2048), (3, 5, 7), (2, 1, 300)):
        x = torch.randn(leaves * R, C, device="cuda", generator=g)
        out = torch.empty(leaves, C, device="cuda")
        check(L.isx_colsum_leaves(x.data_ptr(), leaves, R, C, out.data_ptr(), st), "x")
        assert _rel(out, x.double().view(leaves, R, C).sum(1)) <= 2e-6
        one = torch.empty(1, C, device="cuda")                    # a leaf alone: the same bits
        check(L.isx_colsum_leaves(x[R * (leaves - 1):].contiguous().data_ptr(), 1, R, C, one.data_ptr(), st), "x")
        assert torch.equal(one[0], out[leaves - 1])


def test_head_engine_matches_float64_autograd_and_is_row_count_invariant():
    """HeadEngine forward + backward on 4 micro-batches of 6 rows against torch autograd of the plain head modules in float64: descriptors,
    gradient wrt the trunk output, per-micro-batch bias / Shift gradients, and the (x, dy) rows handed to the RowSink; then the last
    micro-batch alone: bit-identical slices."""
    from isx import backbones, dp
    from isx.head import HeadEngine
    from model.siamese import DescriptorNet, TuneClassif
    torch.manual_seed(0)
    net = DescriptorNet(TuneClassif(backbones.resnet50(pretrained=True, seed=0), 5), 128, (7, 7), untrained=-1).cuda()
    net.feature_reduc1[1].param.data.normal_(0, 0.002)
    net.train()
    assert HeadEngine.applicable(net)
    eng = HeadEngine(net)
    lin, shift = net.feature_reduc1[2], net.feature_reduc1[1]
    Lv, R = 4, 6
    g = torch.Generator(device="cuda").manual_seed(1)
    f = torch.relu(torch.randn(Lv * R, 2048, 7, 7, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    dd = torch.randn(Lv * R, 128, device="cuda", generator=g)
    flat = dp.FlatGrads([shift.param, lin.bias])
    flat_all = torch.zeros(Lv, flat.flat.numel(), device="cuda")
    sink = dp.RowSink([lin.weight])
    d, ctx = eng.forward(f)
    df = eng.backward(ctx, dd, Lv, sink, flat_all, flat.slices)
    # float64 reference through the plain formulas
    l2 = lambda v: v / (v.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()
    f64 = f.double().requires_grad_()
    sp, w64, b64 = shift.param.detach().double().requires_grad_(), lin.weight.detach().double().requires_grad_(), lin.bias.detach().double().requires_grad_()
    x1 = l2(f64.reshape(Lv * R, -1)) + sp.view(1, -1)
    y = x1 @ w64.t() + b64
    d64 = l2(y)
    assert _rel(d, d64) <= 2e-6
    per_leaf = []
    for l in range(Lv):                                       # per-micro-batch gradients of the small parameters: one backward per leaf
        gs, gb = torch.autograd.grad((d64[l * R:(l + 1) * R] * dd[l * R:(l + 1) * R].double()).sum(), (sp, b64), retain_graph=True)
        per_leaf.append((gs, gb))
    gf, gw = torch.autograd.grad((d64 * dd.double()).sum(), (f64, w64))
    assert _rel(df, gf) <= 5e-6
    lo_s, hi_s = flat.slices[shift.param]
    lo_b, hi_b = flat.slices[lin.bias]
    for l in range(Lv):
        assert _rel(flat_all[l, lo_s:hi_s], per_leaf[l][0]) <= 5e-6 and _rel(flat_all[l, lo_b:hi_b], per_leaf[l][1]) <= 5e-6
    sink.finish()
    assert _rel(lin.weight.grad, gw) <= 5e-6
    # the last micro-batch alone
    flat1 = torch.zeros(1, flat.flat.numel(), device="cuda")
    sink1 = dp.RowSink([lin.weight])
    d1, ctx1 = eng.forward(f[(Lv - 1) * R:])
    df1 = eng.backward(ctx1, dd[(Lv - 1) * R:], 1, sink1, flat1, flat.slices)
    assert torch.equal(d1, d[(Lv - 1) * R:]) and torch.equal(df1, df[(Lv - 1) * R:]) and torch.equal(flat1[0], flat_all[Lv - 1])
    assert torch.equal(sink1.x[0][0], ctx[1][(Lv - 1) * R:])          # the rows handed to the sink are the rows of the big pass


def test_weight_gradient_from_rows_on_the_tn_kernel():
    """isx/dp.weight_gradient_from_rows on the GPU = the TN GEMM over the rows: against float64, and one chain per output in row order (a row
    of zeros appended changes nothing, bit for bit; the hipBLASLt result only agrees to rounding)."""
    from isx import dp
    g = torch.Generator(device="cuda").manual_seed(3)
    for R, n_out, n_in in ((192, 2048, 100352), (24, 128, 100352), (77, 64, 640)):
        x = torch.randn(R, n_in, device="cuda", generator=g)
        dy = torch.randn(R, n_out, device="cuda", generator=g)
        got = dp.weight_gradient_from_rows(dy, x)
        assert got.shape == (n_out, n_in) and _rel(got, dy.double().t() @ x.double()) <= 2e-6
        pad = dp.weight_gradient_from_rows(torch.cat([dy, torch.zeros(5, n_out, device="cuda")]), torch.cat([x, torch.zeros(5, n_in, device="cuda")]))
        assert torch.equal(pad, got)
        assert _rel(got, dy.t().mm(x)) <= 1e-5


@pytest.mark.parametrize("nesterov", [False, True])
def test_fused_weight_gradient_and_sgd_step(nesterov):
    """isx_head_sgd_step through isx/dp.fused_sgd_from_rows: the weight gradient is bit for bit the TN kernel's (one chain over the rows), and three
    optimizer steps -- the first creates the momentum buffer -- track torch.optim.SGD stepping on that gradient tensor (the arithmetic differs only by
    fma contraction inside torch's kernel: 1e-6 of the weight scale); the optimizer's own step() then leaves the weight alone (no .grad)."""
    from isx import dp
    g = torch.Generator(device="cuda").manual_seed(11)
    for R, n_out, n_in in ((192, 256, 1280), (77, 128, 100352)):
        w0 = torch.randn(n_out, n_in, device="cuda", generator=g) * 0.01
        w_f = torch.nn.Parameter(w0.clone())
        w_t = torch.nn.Parameter(w0.clone())
        kw = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4, nesterov=nesterov)
        opt_f, opt_t = torch.optim.SGD([w_f], **kw), torch.optim.SGD([w_t], **kw)
        for step in range(3):
            x = torch.randn(R, n_in, device="cuda", generator=g)
            dy = torch.randn(R, n_out, device="cuda", generator=g)
            assert dp.fused_sgd_from_rows(opt_f, w_f, dy, x)
            assert w_f.grad is None
            opt_f.step()                                              # must not touch w_f
            w_t.grad = dp.weight_gradient_from_rows(dy, x)
            opt_t.step()
            scale = float(w_t.detach().abs().max())
            assert float((w_f - w_t).detach().abs().max()) <= 1e-6 * scale, (step, float((w_f - w_t).detach().abs().max()), scale)
            bf, bt = opt_f.state[w_f]["momentum_buffer"], opt_t.state[w_t]["momentum_buffer"]
            assert float((bf - bt).abs().max()) <= 1e-6 * float(bt.abs().max())
        # the update is a function of the rows only: a second weight stepped with the same rows lands on the same bits
        w_g = torch.nn.Parameter(w0.clone())
        opt_g = torch.optim.SGD([w_g], **kw)
        g2 = torch.Generator(device="cuda").manual_seed(5)
        xs = [(torch.randn(R, n_in, device="cuda", generator=g2), torch.randn(R, n_out, device="cuda", generator=g2)) for _ in range(2)]
        w_h = torch.nn.Parameter(w0.clone())
        opt_h = torch.optim.SGD([w_h], **kw)
        for x, dy in xs:
            dp.fused_sgd_from_rows(opt_g, w_g, dy, x)
            dp.fused_sgd_from_rows(opt_h, w_h, torch.cat([dy, torch.zeros(3, n_out, device="cuda")]), torch.cat([x, torch.zeros(3, n_in, device="cuda")]))
        assert torch.equal(w_g, w_h)                                   # zero rows appended: the same chain, the same bits


def test_fused_sgd_declines_what_the_kernel_does_not_cover():
    from isx import dp
    w = torch.nn.Parameter(torch.zeros(100, 64, device="cuda"))         # widths that are not multiples of 128
    assert not dp.fused_sgd_from_rows(torch.optim.SGD([w], lr=0.1), w, torch.zeros(4, 100, device="cuda"), torch.zeros(4, 64, device="cuda"))
    w2 = torch.nn.Parameter(torch.zeros(128, 128, device="cuda"))
    assert not dp.fused_sgd_from_rows(torch.optim.Adam([w2], lr=0.1), w2, torch.zeros(4, 128, device="cuda"), torch.zeros(4, 128, device="cuda"))


@pytest.mark.parametrize("L", [1, 2, 3, 5, 8, 13, 16])
def test_tree_sum_rows_is_the_canonical_tree(L):
    """isx_tree_sum_rows == isx/dp.tree_sum on the same rows, bit for bit (float4 and scalar launches, strided rows, in place into row 0)."""
    from isx import dp, ops
    g = torch.Generator().manual_seed(L)
    for n, pad in ((4096 + 8, 0), (1001, 3)):
        buf = (torch.randn(L, n + pad, generator=g) * torch.logspace(-3, 3, L).view(L, 1)).cuda()
        rows = buf[:, :n]
        want = dp.tree_sum(0, L, lambda i: rows[i].clone())
        got = ops.tree_sum_rows(rows) if pad == 0 else torch.empty(n, device="cuda")
        if pad:
            from isx._lib import check, lib
            check(lib().isx_tree_sum_rows(rows.data_ptr(), L, rows.stride(0), n, got.data_ptr(), torch.cuda.current_stream().cuda_stream), "x")
        assert torch.equal(got, want)
        cpu = dp.tree_sum(0, L, lambda i: rows[i].cpu().clone())                   # and the CPU's fp32 adds in the same order
        assert torch.equal(got.cpu(), cpu)
    rows = torch.randn(L, 512, generator=g).cuda()
    want = dp.tree_sum(0, L, lambda i: rows[i].clone())
    ops.tree_sum_rows(rows, out=rows[0])
    assert torch.equal(rows[0], want)
