"""Forward + backward of the TRAINABLE trunk suffix of siamese training on the hand-written kernels.

The reference trains layer4 of its ResNet (train/siamese_descriptor_p.py:14-17,48 -> model/nn_utils.py:5-23 freeze everything below)
with BatchNorm in eval mode (micro-batch 8 < 16: model/nn_utils.py:160-163, train/siamese_descriptor_p.py:89-93), and leaves forward and
backward of those three bottleneck blocks to torch (`loss.backward()`, utils/train_general.py:51-61).  Here they run as ONE
autograd node over libisx:

  forward   the inference kernels on the FOLDED convolutions w' = w * s, b' = beta - mean * s, s = gamma / sqrt(var + eps)
            (isx_conv1x1_nhwc, isx_conv3x3_nhwc, isx_conv1x1_dual_nhwc: bias / shortcut / ReLU fused), activations kept channels-last
  backward  per convolution one dgrad GEMM with the ReLU mask of the layer below and the shortcut gradient fused into its epilogue
            (isx_conv1x1_dgrad_nhwc / isx_conv3x3_dgrad_nhwc), one weight-gradient GEMM over the pixels (isx_conv_wgrad_nhwc), the
            bias gradient as its by-product, both split over the pixels into partials) and the chain rule of the fold
            (isx_bn_fold_backward: adds the partials in order and ACCUMULATES straight into the parameters' .grad) -- ~45 launches per
            micro-batch where MIOpen's per-image im2col + GEMM loops took ~150.

The folded weights and their re-layouts are derived once per optimizer step (version counters of the parameters) -- the weights do not
change between the micro-batches of a step.

Applicability (otherwise the caller keeps the plain modules + torch autograd): every suffix module is a Bottleneck (1x1 -> 3x3 ->
1x1, optional 1x1 projection), BatchNorm in eval mode, channels multiples of 64, fp32 GPU tensors.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Folded(object):
    """Per-step derived tensors of one convolution + eval-mode BatchNorm."""
    __slots__ = ("conv", "bn", "key", "taps", "cin", "cout", "stride", "scale", "istd", "mean", "bias", "w_fwd", "w_dgrad", "w_col", "cat_key", "w_cat", "bias_cat")

    def __init__(self, conv, bn):
        self.conv, self.bn, self.key, self.cat_key = conv, bn, None, None
        self.taps = conv.kernel_size[0] * conv.kernel_size[1]
        self.cin, self.cout, self.stride = conv.in_channels, conv.out_channels, conv.stride[0]

    def refresh(self):
        c, b = self.conv, self.bn
        src = (c.weight, b.weight, b.bias, b.running_mean, b.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src)
        if key == self.key:
            return self
        if not c.weight.is_contiguous():
            raise _lib.IsxError("suffix engine: convolution weights must be contiguous (OIHW), not channels-last")
        with torch.no_grad():
            self.istd = torch.rsqrt(b.running_var + b.eps)
            self.scale = (b.weight * self.istd).contiguous()
            self.mean = b.running_mean
            self.bias = (b.bias - b.running_mean * self.scale).contiguous()
            wf = c.weight * self.scale.view(-1, 1, 1, 1)                               # (Cout, Cin, kh, kw)
            self.w_fwd = wf.permute(0, 2, 3, 1).contiguous()                            # OHWI: (Cout, kh, kw, Cin)
            if self.taps == 1:
                self.w_dgrad = wf.view(self.cout, self.cin).t().contiguous()            # W'^T: (Cin, Cout)
            elif self.stride == 2:
                self.w_dgrad = None
                self.w_col = self.w_fwd.view(self.cout, self.taps * self.cin).t().contiguous()   # (9 Cin, Cout): per-tap columns of the input gradient
            else:
                self.w_dgrad = wf.flip(2, 3).permute(1, 2, 3, 0).contiguous()           # (Cin, kh, kw, Cout), taps flipped
        self.key = key
        return self


def _bottleneck_parts(block):
    from . import backbones
    if not isinstance(block, backbones.Bottleneck):
        return None
    convs = [(block.conv1, block.bn1), (block.conv2, block.bn2), (block.conv3, block.bn3)]
    down = None
    if block.downsample is not None:
        if len(block.downsample) != 2:
            return None
        down = (block.downsample[0], block.downsample[1])
    (c1, _), (c2, _), (c3, _) = convs
    ok = (c1.kernel_size == (1, 1) and c1.stride == (1, 1) and c1.padding == (0, 0) and c1.bias is None
          and c2.kernel_size == (3, 3) and c2.stride in ((1, 1), (2, 2)) and c2.padding == (1, 1) and c2.dilation == (1, 1) and c2.groups == 1 and c2.bias is None
          and c3.kernel_size == (1, 1) and c3.stride == (1, 1) and c3.padding == (0, 0) and c3.bias is None
          and all(c.in_channels % 64 == 0 and c.out_channels % 64 == 0 and c.groups == 1 for c, _ in convs))
    if down is not None:
        d = down[0]
        ok = ok and (d.kernel_size == (1, 1) and d.padding == (0, 0) and d.stride == c2.stride and d.bias is None and d.in_channels % 64 == 0
                     and d.out_channels == c3.out_channels and isinstance(down[1], torch.nn.BatchNorm2d))
    else:
        ok = ok and c2.stride == (1, 1) and c1.in_channels == c3.out_channels
    return (convs, down) if ok else None


class SuffixEngine(object):
    """The bottleneck blocks `blocks` (modules of net.features from the first trainable one on) as one autograd node."""

    def __init__(self, blocks):
        self.blocks = list(blocks)
        self.layers = []                      # per block: ([_Folded x 3], _Folded | None)
        for b in self.blocks:
            convs, down = _bottleneck_parts(b)
            self.layers.append(([_Folded(c, n) for c, n in convs], _Folded(*down) if down is not None else None))
        self.params = []
        for convs, down in self.layers:
            for f in convs + ([down] if down is not None else []):
                self.params += [f.conv.weight, f.bn.weight, f.bn.bias]

    @staticmethod
    def applicable(blocks):
        blocks = list(blocks)
        if not blocks or any(_bottleneck_parts(b) is None for b in blocks):
            return False
        for b in blocks:
            for m in b.modules():
                if isinstance(m, torch.nn.BatchNorm2d) and (m.training or not m.affine or not m.track_running_stats):
                    return False
        return True

    def __call__(self, x):
        """x: (B, C, H, W) fp32 GPU feature map of the frozen prefix (no graph) -> (B, C', H', W') channels-last."""
        if x.requires_grad:
            raise _lib.IsxError("suffix engine: the input must not require a gradient (the trunk prefix is frozen)")
        return _SuffixFn.apply(x, self, *self.params)

    # ---- kernels on contiguous (B, H, W, C) tensors ------------------------------------------------------------------------------
    @staticmethod
    def _conv1x1(x, f, residual, relu):
        B, H, W, _ = x.shape
        y = torch.empty((B, H, W, f.cout), device=x.device, dtype=torch.float32)
        check(lib().isx_conv1x1_nhwc(x.data_ptr(), B * H * W, f.cin, f.w_fwd.data_ptr(), f.cout, f.bias.data_ptr(),
                                     residual.data_ptr() if residual is not None else None, 1 if relu else 0, y.data_ptr(), _stream()), "isx_conv1x1_nhwc")
        return y

    @staticmethod
    def _conv3x3(x, f, relu):
        B, H, W, _ = x.shape
        Ho, Wo = (H - 1) // f.stride + 1, (W - 1) // f.stride + 1
        y = torch.empty((B, Ho, Wo, f.cout), device=x.device, dtype=torch.float32)
        check(lib().isx_conv3x3_nhwc(x.data_ptr(), B, H, W, f.cin, f.w_fwd.data_ptr(), f.cout, f.stride, f.bias.data_ptr(), None, 1 if relu else 0,
                                     y.data_ptr(), _stream()), "isx_conv3x3_nhwc")
        return y

    @staticmethod
    def _conv1x1_dual(t, x, f3, fd, w_cat, bias):
        B, H, W, _ = x.shape
        Ho, Wo = t.shape[1], t.shape[2]
        y = torch.empty((B, Ho, Wo, f3.cout), device=x.device, dtype=torch.float32)
        check(lib().isx_conv1x1_dual_nhwc(t.data_ptr(), f3.cin, x.data_ptr(), B, H, W, fd.cin, fd.stride, w_cat.data_ptr(), f3.cout, bias.data_ptr(), 1,
                                          y.data_ptr(), _stream()), "isx_conv1x1_dual_nhwc")
        return y

    @staticmethod
    def _wgrad(dz, x, f, leaves=1):
        """Partial gradients of the FOLDED weight and bias per leaf: ((leaves, S, Cout, taps, Cin), (leaves, S, Cout)); dz (B,Ho,Wo,Cout), x (B,H,W,Cin)."""
        B, H, W, _ = x.shape
        S = lib().isx_conv_wgrad_splits(dz.numel() // f.cout // leaves, f.cin, f.cout, f.taps)
        dw = torch.empty((leaves, S, f.cout, f.taps, f.cin), device=x.device, dtype=torch.float32)
        db = torch.empty((leaves, S, f.cout), device=x.device, dtype=torch.float32)
        check(lib().isx_conv_wgrad_nhwc(dz.data_ptr(), x.data_ptr(), B, leaves, H, W, f.cin, f.cout, f.taps, f.stride, dw.data_ptr(), db.data_ptr(), _stream()),
              "isx_conv_wgrad_nhwc")
        return dw, db

    @staticmethod
    def _fold_backward(f, dwp, db, grads, leaf_grads=None):
        """Gradients of (conv.weight, bn.weight, bn.bias) from the partials of the folded convolution.
        leaf_grads = (flat_all (L, total), slices): leaf l's gradients are WRITTEN into row l of flat_all at the parameters' slices (the
        training step's per-leaf flat gradient buffers).  Otherwise (one leaf): parameters that already hold a .grad are accumulated IN PLACE
        (no autograd add pass); the others get a fresh tensor handed back to autograd."""
        params = (f.conv.weight, f.bn.weight, f.bn.bias)
        leaves, S = dwp.shape[0], dwp.shape[1]
        if leaf_grads is not None:
            flat_all, slices = leaf_grads
            base = flat_all.data_ptr()
            ptrs = [base + 4 * slices[p][0] for p in params]
            grads += [None, None, None]
            check(lib().isx_bn_fold_backward(dwp.data_ptr(), db.data_ptr(), leaves, S, f.conv.weight.data_ptr(), f.scale.data_ptr(), f.mean.data_ptr(),
                                             f.istd.data_ptr(), f.cout, f.cin, f.taps, 0, flat_all.stride(0), ptrs[0], ptrs[1], ptrs[2], _stream()),
                  "isx_bn_fold_backward")
            return
        outs = []
        for p in params:
            g = p.grad
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                outs.append(g)
                grads.append(None)
            else:
                g = torch.zeros_like(p, memory_format=torch.contiguous_format)
                outs.append(g)
                grads.append(g)
        check(lib().isx_bn_fold_backward(dwp.data_ptr(), db.data_ptr(), 1, S, f.conv.weight.data_ptr(), f.scale.data_ptr(), f.mean.data_ptr(),
                                         f.istd.data_ptr(), f.cout, f.cin, f.taps, 1, 0, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), _stream()),
              "isx_bn_fold_backward")

    # ---- forward / backward of the whole suffix -------------------------------------------------------------------------------------
    def forward(self, x_nchw):
        x = x_nchw.permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        saved = []
        for convs, down in self.layers:
            f1, f2, f3 = (f.refresh() for f in convs)
            t1 = self._conv1x1(x, f1, None, True)
            t2 = self._conv3x3(t1, f2, True)
            if down is not None:
                fd = down.refresh()
                if fd.cat_key != (f3.key, fd.key):           # [W3' | Wd'] and b3' + bd' of the fused last-conv + projection GEMM, once per step
                    with torch.no_grad():
                        fd.w_cat = torch.cat([f3.w_fwd.view(f3.cout, f3.cin), fd.w_fwd.view(fd.cout, fd.cin)], 1).contiguous()
                        fd.bias_cat = f3.bias + fd.bias
                    fd.cat_key = (f3.key, fd.key)
                y = self._conv1x1_dual(t2, x, f3, fd, fd.w_cat, fd.bias_cat)
            else:
                y = self._conv1x1(t2, f3, x, True)
            saved.append((x, t1, t2, y))
            x = y
        return x.permute(0, 3, 1, 2), saved

    def backward(self, saved, dy_nchw, leaves=1, leaf_grads=None):
        """Backward of forward().  leaves > 1: the batch is `leaves` consecutive micro-batches of equal size whose parameter gradients are
        kept APART (leaf_grads = (flat_all, slices), see _fold_backward) -- forward and dgrad kernels compute every row as they would in a
        launch of its own, the weight-gradient kernel restarts its pixel sum at every leaf, so leaf l's gradient is bit for bit what a
        launch of leaf l alone produces."""
        if leaves > 1 and leaf_grads is None:
            raise _lib.IsxError("suffix engine: per-leaf gradients need leaf_grads")
        dy = dy_nchw.permute(0, 2, 3, 1)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        L = lib()
        st = _stream()
        grads_rev = []                                        # per block (last first): grads of [c1 x3, c2 x3, c3 x3, (down x3)]
        dS = None
        for bi in range(len(self.layers) - 1, -1, -1):
            (f1, f2, f3), fd = self.layers[bi]
            x, t1, t2, y = saved[bi]
            B, H, W, _ = x.shape
            Ho, Wo = t2.shape[1], t2.shape[2]
            M2 = B * Ho * Wo
            if dS is None:                                    # the last block: backward of its output ReLU
                dS = torch.empty_like(y)
                check(L.isx_relu_grad(dy.data_ptr(), y.data_ptr(), y.numel(), dS.data_ptr(), st), "isx_relu_grad")
            # (further down the mask is fused into the dgrad of the block above)
            g1, g2, g3, gd = [], [], [], []
            # conv3 (+ projection): weight gradients (the bias gradient = column sums of dS comes with them), then the gradient wrt t2 with t2's ReLU fused
            self._fold_backward(f3, *self._wgrad(dS, t2, f3, leaves), g3, leaf_grads)
            if fd is not None:
                self._fold_backward(fd, *self._wgrad(dS, x, fd, leaves), gd, leaf_grads)
            dT2 = torch.empty_like(t2)
            check(L.isx_conv1x1_dgrad_nhwc(dS.data_ptr(), M2, f3.cout, f3.w_dgrad.data_ptr(), f3.cin, None, t2.data_ptr(), dT2.data_ptr(), st),
                  "isx_conv1x1_dgrad_nhwc")
            # conv2 (3x3)
            self._fold_backward(f2, *self._wgrad(dT2, t1, f2, leaves), g2, leaf_grads)
            dT1 = torch.empty_like(t1)
            if f2.stride == 2:                                # per-tap columns over the OUTPUT pixels (one GEMM), then each input pixel gathers its 1 / 2 / 4 taps
                dcol = torch.empty((M2, 9 * f2.cin), device=x.device, dtype=torch.float32)
                check(L.isx_conv1x1_dgrad_nhwc(dT2.data_ptr(), M2, f2.cout, f2.w_col.data_ptr(), 9 * f2.cin, None, None, dcol.data_ptr(), st), "isx_conv1x1_dgrad_nhwc")
                check(L.isx_conv3x3_s2_col2im_nhwc(dcol.data_ptr(), B, H, W, f2.cin, t1.data_ptr(), dT1.data_ptr(), st), "isx_conv3x3_s2_col2im_nhwc")
            else:
                check(L.isx_conv3x3_dgrad_nhwc(dT2.data_ptr(), B, H, W, f2.cout, f2.w_dgrad.data_ptr(), f2.cin, t1.data_ptr(), dT1.data_ptr(), st),
                      "isx_conv3x3_dgrad_nhwc")
            # conv1
            self._fold_backward(f1, *self._wgrad(dT1, x, f1, leaves), g1, leaf_grads)
            grads_rev.append(g1 + g2 + g3 + gd)
            if bi == 0:
                break                                         # the prefix below is frozen and carries no graph: no gradient wrt x
            # gradient wrt the block input, with the ReLU of the block below (whose output IS x) fused: this is that block's dS
            if fd is None:
                add = dS
            else:                                             # projection shortcut inside the trainable suffix (rare: layer3 + layer4 trained)
                dd = torch.empty((B, Ho, Wo, fd.cin), device=x.device, dtype=torch.float32)
                check(L.isx_conv1x1_dgrad_nhwc(dS.data_ptr(), M2, fd.cout, fd.w_dgrad.data_ptr(), fd.cin, None, None, dd.data_ptr(), st), "isx_conv1x1_dgrad_nhwc")
                add = torch.zeros_like(x)
                add[:, ::fd.stride, ::fd.stride] = dd
            dX = torch.empty_like(x)
            check(L.isx_conv1x1_dgrad_nhwc(dT1.data_ptr(), B * H * W, f1.cout, f1.w_dgrad.data_ptr(), f1.cin, add.data_ptr(), x.data_ptr(), dX.data_ptr(), st),
                  "isx_conv1x1_dgrad_nhwc")
            dS = dX
        out = []
        for g in reversed(grads_rev):
            out += g
        return out


class _SuffixFn(Function):
    @staticmethod
    def forward(ctx, x, engine, *params):
        y, saved = engine.forward(x.detach())
        ctx.engine, ctx.saved = engine, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        grads = ctx.engine.backward(ctx.saved, dy)
        ctx.saved = None
        return (None, None) + tuple(grads)
