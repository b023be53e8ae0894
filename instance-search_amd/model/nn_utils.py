"""Backbone surgery helpers -- same names and behaviour as the reference's
model/nn_utils.py (extract_layers :56-71, convolutionalize :26-39, get_feature_size :42-53,
set_untrained_blocks :6-23, set_net_train :160-163), written against isx.backbones instead
of torchvision.  The BN copy/replace helpers (:74-134) are training-time surgery and are
out of scope."""
import os

import torch
import torch.nn as nn

from isx import backbones as models


def set_untrained_blocks(containers, n):
    """n < 0 freezes everything; otherwise the first n parameterised modules (counted across
    the containers in order) are frozen and the rest left trainable."""
    params_of = lambda m: list(m.parameters())
    for seq in containers:
        for m in seq:
            for p in params_of(m):
                p.requires_grad = n >= 0
    frozen = 0
    for seq in containers:
        for m in seq:
            if frozen >= n:
                break
            ps = params_of(m)
            if not ps:
                continue            # parameter-free modules do not count
            for p in ps:
                p.requires_grad = False
            frozen += 1


def convolutionalize(fc, in_size2d):
    """Linear(C*h*w -> out) as Conv2d(C -> out, kernel (h, w)): weight rows are viewed
    (C, h, w), i.e. channel-major then h then w."""
    kh, kw = in_size2d
    if fc.in_features % (kh * kw) != 0:
        raise ValueError('FC in_feature size {0} is not divisible by in_size2d {1}'.format(fc.in_features, in_size2d))
    cin = fc.in_features // (kh * kw)
    conv = nn.Conv2d(cin, fc.out_features, (kh, kw), bias=fc.bias is not None)
    with torch.no_grad():
        conv.weight.copy_(fc.weight.reshape(fc.out_features, cin, kh, kw))
        if fc.bias is not None:
            conv.bias.copy_(fc.bias)
    return conv.to(fc.weight.device)


def get_feature_size(seq, factor=1, default=-1):
    """Output width of the last conv-like / linear module of `seq` (conv widths times `factor`)."""
    size = default
    for m in seq:
        if isinstance(m, models.Bottleneck):
            size = m.conv3.out_channels * factor
        elif isinstance(m, models.BasicBlock):
            size = m.conv2.out_channels * factor
        elif isinstance(m, nn.Conv2d):
            size = m.out_channels * factor
        elif isinstance(m, nn.Linear):
            size = m.out_features
    return size


def extract_layers(net):
    """(features, feature_reduc, classifier) of a backbone."""
    if all(hasattr(net, a) for a in ('features', 'feature_reduc', 'classifier')):
        return net.features, net.feature_reduc, net.classifier
    if isinstance(net, models.ResNet):
        stem = [net.conv1, net.bn1, net.relu, net.maxpool]
        blocks = [b for layer in (net.layer1, net.layer2, net.layer3, net.layer4) for b in layer]
        return nn.Sequential(*(stem + blocks)), nn.Sequential(net.avgpool), nn.Sequential(net.fc)
    return net.features, nn.Sequential(), net.classifier


# A/B switches (environment).  ISX_CONV1X1=0 sends the 1x1 trunk convolutions back to MIOpen.  ISX_CONV3X3: "auto"
# (default) runs the implicit-GEMM kernel where it wins on MI355X -- every 3x3 convolution of a BN-folded residual block (round 2,
# with the buffer-instruction epilogue: 1.93-1.98 ms against MIOpen's igemm + zero fill + separate epilogue pass 2.05-2.15 ms at
# Cin >= 256, B = 1024; ResNet-50 extraction +2-3 % at B = 256 and at 448x448) and stand-alone convolutions with Cin <= 128; the
# 13x13 AlexNet layers (192->384, 384->256, 256->256) stay with MIOpen (-5 % otherwise) -- "1" everywhere, "0" never, an integer
# N > 1: Cin <= N.
_CONV3X3_MODE = os.environ.get("ISX_CONV3X3", "auto")
_IMPLICIT_GEMM_3X3 = _CONV3X3_MODE != "0"
_GEMM_1X1 = os.environ.get("ISX_CONV1X1", "1") != "0"
_FUSE_EXPAND = os.environ.get("ISX_FUSE_EXPAND", "1") != "0"     # 0: conv2 and conv3 of the 64-channel bottlenecks as two kernels again
_FUSED_STEM = os.environ.get("ISX_STEM", "1") != "0"       # 0: stem convolution back to MIOpen + the separate bias/ReLU/maxpool pass
_FUSE_PROJECTION = os.environ.get("ISX_FUSE_PROJECTION", "1") != "0"     # last 1x1 conv + projection shortcut as one GEMM
if not (_GEMM_1X1 and _FUSED_STEM and _FUSE_EXPAND and _FUSE_PROJECTION and _CONV3X3_MODE == "auto"):
    import warnings
    warnings.warn("libisx A/B switches are set (ISX_CONV1X1 / ISX_CONV3X3 / ISX_STEM / ISX_FUSE_EXPAND / ISX_FUSE_PROJECTION): some trunk "
                  "convolutions run on MIOpen or unfused in this process -- a measurement aid, not the product path")

# Convolutions that ran on torch's conv2d (MIOpen) on a GPU tensor in this process, as {(kernel, stride, Cin, Cout): calls}: the ResNet
# trunks leave it empty (every layer is a libisx kernel); AlexNet's 11x11 / 5x5 layers and its 3x3 layers at 13x13 with Cin > 128, a 3x3
# with Cin % 32 != 0, grouped / dilated convolutions and stand-alone strided 1x1 layers land here ("what still runs on MIOpen").
TORCH_CONV_CALLS = {}


def _note_torch_conv(c):
    key = (tuple(c.kernel_size), tuple(c.stride), c.in_channels, c.out_channels)
    TORCH_CONV_CALLS[key] = TORCH_CONV_CALLS.get(key, 0) + 1


def _derived(owner, slot, sources, build):
    """A tensor derived from weights (`sources`), cached on `owner` under `slot` and rebuilt when a source was written in place
    (load_state_dict copies into the parameters: the version counter moves), replaced, or moved to another device.  Writes through
    `.data` bypass the version counter: call `invalidate_derived_weights(module)` after such surgery."""
    key = tuple((t.data_ptr(), t._version, str(t.device)) for t in sources)
    hit = owner.__dict__.get(slot)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        owner.__dict__[slot] = hit
    return hit[1]


def invalidate_derived_weights(module):
    """Drop every cached re-layout of a weight (OHWI copies, concatenated / transposed projection weights) under `module`."""
    for m in module.modules():
        for slot in ('_c_w_ohwi', '_c_w_cat', '_c_w3t', '_hwc', '_c_pad64'):
            if slot in m.__dict__:
                m.__dict__[slot] = None


class _ConvBiasAct(nn.Module):
    """Bias-free convolution + fused `y = act(y + bias (+ residual))` epilogue (libisx `isx_bias_act_inplace` on
    the GPU, plain torch otherwise)."""

    def __init__(self, conv, relu):
        super().__init__()
        bias = conv.bias.detach().clone() if conv.bias is not None else torch.zeros(conv.out_channels)
        self.conv = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                              conv.groups, bias=False)
        self.conv.weight = nn.Parameter(conv.weight.detach().clone(), requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False)
        self.relu = relu
        self.plain = False            # True: no epilogue of its own (projection shortcut, bias merged elsewhere)
        self.in_block = False         # True: a convolution of a BN-folded residual block (_FusedBlock)

    def w_ohwi(self):
        """(Cout,kh,kw,Cin) copy of the weight for the implicit-GEMM / stem kernels; follows the weight (see _derived)."""
        w = self.conv.weight
        return _derived(self, '_c_w_ohwi', (w,), lambda: w.detach().permute(0, 2, 3, 1).contiguous())

    def _pointwise(self):
        c = self.conv
        return _GEMM_1X1 and c.kernel_size == (1, 1) and c.padding == (0, 0) and c.groups == 1 and c.dilation == (1, 1) and c.stride == (1, 1)

    def _three_by_three(self):
        c = self.conv
        return (_IMPLICIT_GEMM_3X3 and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.groups == 1 and c.dilation == (1, 1)
                and c.stride in ((1, 1), (2, 2)) and c.in_channels % 32 == 0
                and (_CONV3X3_MODE == "1" or (_CONV3X3_MODE == "auto" and self.in_block)
                     or c.in_channels <= (int(_CONV3X3_MODE) if _CONV3X3_MODE.isdigit() and int(_CONV3X3_MODE) > 1 else 128)))

    def forward(self, x, residual=None):
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and self._three_by_three()
                and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
            # 3x3 convolution on channels-last activations: implicit GEMM on the fp32 matrix cores, epilogue fused
            from isx import ops
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            return ops.conv3x3_nhwc(x, self.w_ohwi(), self.bias, self.conv.stride[0], residual, self.relu)
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and self._pointwise()
                and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
            # 1x1 convolution on channels-last activations: one fp32-MFMA GEMM over the pixels, epilogue fused
            from isx import ops
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            return ops.conv1x1_nhwc(x, self.conv.weight, self.bias, residual, self.relu)
        y = self.conv(x)
        if y.is_cuda:
            _note_torch_conv(self.conv)
        if self.plain:
            return y
        if y.is_cuda and y.dtype == torch.float32 and not torch.is_grad_enabled():
            from isx import ops
            if residual is not None and residual.stride() != y.stride():
                residual = residual.contiguous(memory_format=torch.channels_last if not y.is_contiguous() else torch.contiguous_format)
            return ops.bias_act_(y, self.bias, residual, self.relu)
        if not torch.is_grad_enabled():
            # CPU inference (the parity side of the GPU tests, CPU-only users): the epilogue in place on the convolution's own output -- the same
            # fp32 adds in the same order, without three more passes' worth of fresh 100 MB tensors per layer
            y.add_(self.bias.view(1, -1, 1, 1).to(y.dtype))
            if residual is not None:
                y.add_(residual)
            return y.relu_() if self.relu else y
        y = y + self.bias.view(1, -1, 1, 1).to(y.dtype)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if self.relu else y


class _StemConvPool(nn.Module):
    """conv + folded BN + ReLU + MaxPool2d(3, 2, 1) of the ResNet stem: the bias / ReLU epilogue and the pooling run
    as ONE pass over the convolution output (libisx `isx_bias_relu_maxpool_nhwc`) on channels-last GPU tensors."""

    def __init__(self, conv, pool):
        super().__init__()
        self.cba = _ConvBiasAct(conv, relu=True)
        self.pool = pool

    def forward(self, x):
        c = self.cba
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and c.conv.out_channels % 4 == 0):
            from isx import ops
            if _FUSED_STEM and ops.stem7x7_pool_applicable(x, c.conv):
                # images up to 896 wide: convolution + bias + ReLU + pooling as ONE kernel, nothing in between touches memory
                return ops.stem7x7_pool(x, c.w_ohwi(), c.bias)
            y = c.conv(x)
            _note_torch_conv(c.conv)
            if y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous():
                return ops.bias_relu_maxpool(y, c.bias)
            return self.pool(ops_bias_act(y, c.bias, c.relu))
        return self.pool(c(x))


def ops_bias_act(y, bias, relu):
    from isx import ops
    return ops.bias_act_(y, bias, None, relu)


def _is_stem_pool(m):
    return (isinstance(m, nn.MaxPool2d) and m.kernel_size in (3, (3, 3)) and m.stride in (2, (2, 2)) and m.padding in (1, (1, 1))
            and m.dilation in (1, (1, 1)) and not m.ceil_mode)


class _FusedBlock(nn.Module):
    """Inference form of a (BN-folded) BasicBlock / Bottleneck: relu(convN(...) + bias + identity) with every
    bias / residual / ReLU fused into one in-place pass per convolution.  The projection shortcut keeps no
    epilogue of its own: its bias is added to the last convolution's bias."""

    def __init__(self, convs, downsample):
        super().__init__()
        self.convs = nn.ModuleList([_ConvBiasAct(c, relu=True) for c in convs])
        for c in self.convs:
            c.in_block = True
        self.downsample = None
        if downsample is not None:
            d = _ConvBiasAct(downsample, relu=False)
            d.bias = nn.Parameter(torch.zeros_like(d.bias), requires_grad=False)
            d.plain = True
            self.convs[-1].bias = nn.Parameter(self.convs[-1].bias + _ConvBiasAct(downsample, relu=False).bias, requires_grad=False)
            self.downsample = d                                             # projection (its bias lives in the last conv's epilogue)

    def w_cat(self):
        """[W_last | W_projection] (Cout, K1 + K2) for the fused last-conv + shortcut GEMM; follows both weights (see _derived)."""
        wl, wd = self.convs[-1].conv.weight, self.downsample.conv.weight
        co = wl.shape[0]
        return _derived(self, '_c_w_cat', (wl, wd), lambda: torch.cat([wl.detach().reshape(co, -1), wd.detach().reshape(co, -1)], 1).contiguous())

    def w3t(self):
        """Transposed expansion weight of the fused conv2 + conv3 kernels: [W3 | Wd]^T with a projection shortcut, W3^T without."""
        wl = self.convs[-1].conv.weight
        if self.downsample is not None:
            wd = self.downsample.conv.weight
            return _derived(self, '_c_w3t', (wl, wd), lambda: self.w_cat().t().contiguous())
        return _derived(self, '_c_w3t', (wl,), lambda: wl.detach().reshape(wl.shape[0], -1).t().contiguous())

    def _fusable_projection(self, x):
        last, d = self.convs[-1], self.downsample
        return (_GEMM_1X1 and _FUSE_PROJECTION and d is not None and last._pointwise() and d.conv.kernel_size == (1, 1) and d.conv.padding == (0, 0)
                and d.conv.groups == 1 and d.conv.stride in ((1, 1), (2, 2)) and last.conv.in_channels % 32 == 0 and d.conv.in_channels % 32 == 0
                and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())

    def _fusable_expand_dual(self):
        if not (_FUSE_EXPAND and _IMPLICIT_GEMM_3X3 and len(self.convs) == 3):
            return False
        c2, c3, d = self.convs[1].conv, self.convs[2].conv, self.downsample.conv
        return (c2.kernel_size == (3, 3) and c2.padding == (1, 1) and c2.groups == 1 and c2.dilation == (1, 1) and c2.stride == (1, 1)
                and c2.out_channels == 64 and c2.in_channels % 32 == 0 and self.convs[1].relu and c3.in_channels == 64 and c3.out_channels == 256
                and d.in_channels == 64 and d.stride == (1, 1))

    def _fusable_expand(self, x):
        if not (_FUSE_EXPAND and _GEMM_1X1 and _IMPLICIT_GEMM_3X3 and self.downsample is None and len(self.convs) == 3):
            return False
        c2, c3 = self.convs[1].conv, self.convs[2].conv
        return (c2.kernel_size == (3, 3) and c2.padding == (1, 1) and c2.groups == 1 and c2.dilation == (1, 1) and c2.stride in ((1, 1), (2, 2))
                and c2.out_channels == 64 and c2.in_channels % 32 == 0 and self.convs[1].relu and self.convs[2]._pointwise() and c3.in_channels == 64
                and c3.out_channels == 256 and x.shape[1] == 256 and c2.stride == (1, 1)
                and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous())

    def forward(self, x):
        if self._fusable_projection(x):
            # relu(conv_last(t) + projection(x) + bias) as ONE GEMM over [t ; x_strided] (libisx isx_conv1x1_dual_nhwc):
            # the shortcut tensor is never materialised
            from isx import ops
            last, d = self.convs[-1], self.downsample
            if self._fusable_expand_dual():
                # first block of the 64-channel stage: conv2 + conv3 + projection + ReLU as ONE kernel (isx_conv3x3_expand_dual_nhwc)
                c2 = self.convs[1]
                t = self.convs[0](x)
                if not t.is_contiguous(memory_format=torch.channels_last):
                    t = t.contiguous(memory_format=torch.channels_last)
                return ops.conv3x3_expand_dual_nhwc(t, c2.w_ohwi(), c2.bias, x, self.w3t(), last.bias, last.relu)
            t = x
            for c in self.convs[:-1]:
                t = c(t)
            if not t.is_contiguous(memory_format=torch.channels_last):
                t = t.contiguous(memory_format=torch.channels_last)
            return ops.conv1x1_dual_nhwc(t, x, self.w_cat(), last.bias, d.conv.stride[0], last.relu)
        if self._fusable_expand(x):
            # Bottleneck with 64 mid channels and an identity shortcut (ResNet stage 1): conv2 + conv3 + residual + ReLU as ONE kernel
            # (libisx isx_conv3x3_expand_nhwc): the mid activation never reaches memory
            from isx import ops
            c2, c3 = self.convs[1], self.convs[2]
            t = self.convs[0](x)
            if not t.is_contiguous(memory_format=torch.channels_last):
                t = t.contiguous(memory_format=torch.channels_last)
            return ops.conv3x3_expand_nhwc(t, c2.w_ohwi(), c2.bias, c2.conv.stride[0], self.w3t(), c3.bias, x, c3.relu)
        idt = x if self.downsample is None else self.downsample(x)
        y = x
        for c in self.convs[:-1]:
            y = c(y)
        return self.convs[-1](y, idt)


class _ChannelsLastEntry(nn.Module):
    """First module of a folded trunk: a GPU fp32 image batch enters in channels-last memory, so that every activation
    behind it is the row-major (pixels, channels) matrix the hand-written convolution kernels (and MIOpen's NHWC kernels)
    consume.  Same values, another memory format; the 3-channel input is the only tensor ever converted."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and not torch.is_grad_enabled():
            return x.contiguous(memory_format=torch.channels_last)
        return x


def fold_batch_norm(features, fuse_epilogues=True):
    """Inference-only copy of a `features` trunk with every eval-mode BatchNorm2d folded into the
    convolution in front of it (w' = w * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)).
    Same function up to fp32 rounding (max relative deviation ~2e-6 on the ResNet-50 map).  With
    `fuse_epilogues` the bias-add / residual-add / ReLU after each convolution run as ONE in-place HIP kernel
    (`isx_bias_act_inplace`) instead of 4-7 separate passes over the activation.  Not part of the reference
    (its BN helpers, model/nn_utils.py:74-155, only copy / re-create / freeze BN layers)."""
    import copy
    from torch.nn.utils.fusion import fuse_conv_bn_eval

    def fold_block(b):
        convs = [fuse_conv_bn_eval(copy.deepcopy(getattr(b, c)).eval(), copy.deepcopy(getattr(b, n)).eval())
                 for c, n in (('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3')) if hasattr(b, c)]
        down = None
        if b.downsample is not None:
            down = fuse_conv_bn_eval(copy.deepcopy(b.downsample[0]).eval(), copy.deepcopy(b.downsample[1]).eval())
        if fuse_epilogues:
            return _FusedBlock(convs, down)
        b = copy.deepcopy(b)
        for i, (c, n) in enumerate((('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3'))):
            if hasattr(b, c):
                setattr(b, c, convs[i])
                setattr(b, n, nn.Identity())
        if down is not None:
            b.downsample = nn.Sequential(down)
        return b

    mods, out, i = list(features), [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            conv = fuse_conv_bn_eval(copy.deepcopy(m).eval(), copy.deepcopy(mods[i + 1]).eval())
            i += 2
            if fuse_epilogues and i + 1 < len(mods) and isinstance(mods[i], nn.ReLU) and _is_stem_pool(mods[i + 1]):
                out.append(_StemConvPool(conv, copy.deepcopy(mods[i + 1])))     # stem: epilogue + pooling in one pass
                i += 2
            elif fuse_epilogues and i < len(mods) and isinstance(mods[i], nn.ReLU):
                out.append(_ConvBiasAct(conv, relu=True))      # conv + BN + ReLU -> conv, one fused epilogue
                i += 1
            else:
                out.append(conv)
            continue
        out.append(fold_block(m) if isinstance(m, (models.Bottleneck, models.BasicBlock)) else copy.deepcopy(m))
        i += 1
    if fuse_epilogues:
        out.insert(0, _ChannelsLastEntry())
    return nn.Sequential(*out).eval()


def set_batch_norm_train(seq, train):
    for m in seq.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.train(mode=train)


def set_net_train(net, train, bn_train=False):
    """train/eval switch; in train mode BatchNorm (only net.features has any) stays frozen
    unless bn_train."""
    net.train(mode=train)
    if train and not bn_train:
        set_batch_norm_train(net.features, False)
