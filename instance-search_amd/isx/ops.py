"""Thin torch-tensor wrappers over the C ABI (include/isx.h).  One function per entry.

All tensors must live on the GPU; work is enqueued on torch's current stream.  These are
the only places the product path touches ctypes."""
import torch

from . import _lib
from ._lib import check, lib

EPS = 1e-10  # model/custom_modules.py:48 NormalizeL2Fun(eps=1e-10)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _on_current_device(t, name):
    """Launches go to the CURRENT device's current stream: a tensor of another GPU would be touched from the wrong
    device / stream, so it is refused (select the device with torch.cuda.device(...) around the call)."""
    if t.device.index != torch.cuda.current_device():
        raise _lib.IsxError("%s lives on %s but the current device is cuda:%d -- wrap the call in torch.cuda.device(%s)"
                            % (name, t.device, torch.cuda.current_device(), t.device.index))


def _f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.IsxError("%s must be a CUDA tensor (libisx has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise _lib.IsxError("%s must be float32, got %s" % (name, t.dtype))
    _on_current_device(t, name)
    return t.contiguous()


def _typed(t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.IsxError("%s must be a CUDA tensor" % name)
    _on_current_device(t, name)
    return t.to(dtype).contiguous()


def l2norm_rows(x, eps=EPS, out=None):
    x = _f32(x, "x")
    B, D = x.shape
    y = torch.empty_like(x) if out is None else out
    check(lib().isx_l2norm_rows(x.data_ptr(), B, D, eps, y.data_ptr(), _stream()), "isx_l2norm_rows")
    return y


def l2norm_rows_bwd(x, dy, eps=EPS):
    """Gradient of l2norm_rows wrt x: (n2 dy - x <x, dy>) / (n2 sqrt(n2)), n2 = sum x^2 + eps, per row."""
    x = _f32(x, "x")
    dy = _f32(dy, "dy")
    if x.shape != dy.shape:
        raise _lib.IsxError("l2norm_rows_bwd: x and dy must have the same shape")
    B, D = x.shape
    dx = torch.empty_like(x)
    check(lib().isx_l2norm_rows_bwd(x.data_ptr(), dy.data_ptr(), B, D, eps, dx.data_ptr(), _stream()), "isx_l2norm_rows_bwd")
    return dx


def l2norm_shift_rows(x, shift=None, eps=EPS):
    x = _f32(x, "x")
    B, F = x.shape
    sp = 0
    if shift is not None:
        shift = _f32(shift, "shift")
        assert shift.numel() == F
        sp = shift.data_ptr()
    y = torch.empty_like(x)
    check(lib().isx_l2norm_shift_rows(x.data_ptr(), sp, B, F, eps, y.data_ptr(), _stream()), "isx_l2norm_shift_rows")
    return y


_HEAD_WS = {}


def head_linear_applicable(rows, weight, any_width=False):
    """any_width: N need not be a multiple of 64 nor K of 32 (head_linear_any pads the weight with zeros)."""
    K = weight.size(1) if weight.dim() == 2 else 0
    Kp = (K + 31) // 32 * 32 if any_width else K
    return (rows.is_cuda and rows.dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_contiguous() and rows.dim() == 2
            and weight.dim() == 2 and Kp % 32 == 0 and Kp > 0 and (any_width or weight.size(0) % 64 == 0) and 192 * Kp * 4 < 2 ** 32
            and rows.size(0) < 2 ** 24)


def pad_rows_to_64(weight, bias=None):
    """(weight, bias) padded with zeros to the GEMM's granules: rows (output features) to a multiple of 64 (head_linear's tile height) and
    -- for a weight whose K is not a multiple of 32 -- columns to a multiple of 32 (the k-tile)."""
    N, K = weight.shape
    Np, Kp = (N + 63) // 64 * 64, (K + 31) // 32 * 32
    wp = weight.new_zeros((Np, Kp))
    wp[:N, :K].copy_(weight.detach())
    bp = None
    if bias is not None:
        bp = bias.new_zeros((Np,))
        bp[:N].copy_(bias.detach())
    return wp, bp


def head_linear_any(rows, weight, bias=None, padded=None):
    """head_linear for any number of output features and any K: zeros pad the weight (and the bias) to a multiple of 64 outputs and of 32
    inputs, the rows are padded with zero columns to the same K, the padding outputs are dropped.  Every output element is its own set of
    k-ordered chains, and a zero product leaves a chain's value untouched (fma(0, 0, acc) == acc), so the padding changes no value: the
    result is what the kernel would compute with a zero-filled k tail, still independent of the batch (no torch / hipBLASLt GEMM, whose
    kernel choice depends on M).  `padded`: the (weight, bias) pair of pad_rows_to_64, kept by the caller next to its weight
    (model/siamese.RowsLinear caches it per version of the parameter); None: padded here, per call.
    For the classifier layers of TuneClassif (2048 -> 464 class scores, reference model/siamese.py:28-32): a row's scores do not depend on the
    batch it is computed in, as with every other descriptor of the path."""
    N, K = weight.shape
    if N % 64 == 0 and K % 32 == 0:
        return head_linear(rows, weight, bias)
    wp, bp = padded if padded is not None else pad_rows_to_64(weight, bias)
    if wp.size(1) != K:
        rows = torch.nn.functional.pad(rows, (0, wp.size(1) - K))
    y = head_linear(rows, wp, bp)
    return y if N == wp.size(0) else y[:, :N].contiguous()


def head_linear(rows, weight, bias=None):
    """y = rows . weight^T + bias on libisx's split-K GEMM (isx_head_linear_fwd_rows): a row's result does not depend on how many rows ride along,
    and it is the same kernel the training step runs -- one implementation of DescriptorNet's Linear (reference model/siamese.py:104-122).
    rows: (M, K) fp32, weight: (N, K) as nn.Linear stores it, K % 32 == 0, N % 64 == 0."""
    rows, weight = _f32(rows, "rows"), _f32(weight, "weight")
    if bias is not None:
        bias = _f32(bias, "bias")
    M, K = rows.shape
    if weight.dim() != 2 or weight.size(1) != K or K % 32 != 0 or weight.size(0) % 64 != 0:
        raise _lib.IsxError("head_linear: rows (%d, %d) against weight %s (K %% 32 == 0, N %% 64 == 0)" % (M, K, tuple(weight.shape)))
    N = weight.size(0)
    y = torch.empty((M, N), dtype=torch.float32, device=rows.device)
    if M == 0:
        return y
    need = lib().isx_head_linear_rows_workspace(M, K, N) // 4
    key = rows.device.index
    ws = _HEAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _HEAD_WS[key] = torch.empty(need, dtype=torch.float32, device=rows.device)
    check(lib().isx_head_linear_fwd_rows(rows.data_ptr(), M, K, weight.data_ptr(), N, bias.data_ptr() if bias is not None else None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 4, _stream()), "isx_head_linear_fwd_rows")
    return y


def gap_l2(fmap, eps=EPS, out=None):
    """Global average pool + L2 of a logical (B,C,H,W) feature map.  A channels-last tensor (the
    layout NHWC convolutions produce) is consumed in place by the NHWC kernel -- no transpose."""
    if not (isinstance(fmap, torch.Tensor) and fmap.is_cuda):
        raise _lib.IsxError("fmap must be a CUDA tensor (libisx has no CPU path)")
    if fmap.dtype != torch.float32:
        raise _lib.IsxError("fmap must be float32, got %s" % (fmap.dtype,))
    _on_current_device(fmap, "fmap")
    B, Cc, H, W = fmap.shape
    y = torch.empty((B, Cc), device=fmap.device, dtype=torch.float32) if out is None else out
    assert y.is_contiguous() and y.shape == (B, Cc)
    nhwc = (H * W > 1 and Cc > 1) and not fmap.is_contiguous() and fmap.is_contiguous(memory_format=torch.channels_last)
    if nhwc:
        check(lib().isx_gap_l2_nhwc(fmap.data_ptr(), B, Cc, H, W, eps, y.data_ptr(), _stream()), "isx_gap_l2_nhwc")
    else:
        fmap = fmap.contiguous()
        check(lib().isx_gap_l2(fmap.data_ptr(), B, Cc, H, W, eps, y.data_ptr(), _stream()), "isx_gap_l2")
    return y


def bias_act_(y, bias, residual=None, relu=True):
    """In place y = act(y + bias[channel] (+ residual)) on a dense (B,C,H,W) tensor in NCHW or channels-last
    memory (the residual must share y's memory format)."""
    if not (y.is_cuda and y.dtype == torch.float32):
        raise _lib.IsxError("y must be a float32 CUDA tensor")
    _on_current_device(y, "y")
    B, Cc, H, W = y.shape
    if y.is_contiguous():
        inner = H * W
    elif y.is_contiguous(memory_format=torch.channels_last):
        inner = 1
    else:
        raise _lib.IsxError("y must be dense NCHW or channels-last")
    rp = 0
    if residual is not None:
        if residual.shape != y.shape or residual.stride() != y.stride() or residual.dtype != torch.float32:
            raise _lib.IsxError("residual must match y's shape, dtype and memory format")
        rp = residual.data_ptr()
    check(lib().isx_bias_act_inplace(y.data_ptr(), _f32(bias, "bias").data_ptr(), rp, y.numel(), Cc, inner, 1 if relu else 0, _stream()),
          "isx_bias_act_inplace")
    return y


def images_u8_to_f32(img_u8, mean, std, channels_last=True):
    """ToTensor + Normalize on the GPU: (B,H,W,3) uint8 RGB -> (B,3,H,W) fp32 (channels-last memory by default)."""
    if not (img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[3] == 3 and img_u8.is_contiguous()):
        raise _lib.IsxError("img_u8 must be a contiguous (B,H,W,3) uint8 CUDA tensor")
    _on_current_device(img_u8, "img_u8")
    B, H, W, _ = img_u8.shape
    out = torch.empty((B, 3, H, W), device=img_u8.device, dtype=torch.float32,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    m, s_ = [float(v) for v in mean], [float(v) for v in std]
    check(lib().isx_images_u8_to_f32(img_u8.data_ptr(), B, H, W, m[0], m[1], m[2], s_[0], s_[1], s_[2], 1 if channels_last else 0,
                                     out.data_ptr(), _stream()), "isx_images_u8_to_f32")
    return out


# Optional per-launch timing of the trunk kernels (bench.py): a list that receives
# (kernel, algorithmic FLOP, algorithmic bytes, start event, end event) per call; None = off.
KERNEL_TIMER = None


def _timed(name, flop, nbytes, fn):
    if KERNEL_TIMER is None:
        return fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    KERNEL_TIMER.append((name, flop, nbytes, a, b))
    return out


def bias_relu_maxpool(y, bias):
    """relu(y + bias[c]) -> MaxPool2d(3, 2, 1) of a channels-last (B,C,H,W) fp32 tensor in one pass."""
    if not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.IsxError("y must be a channels-last float32 CUDA tensor (B,C,H,W)")
    _on_current_device(y, "y")
    B, Cc, H, W = y.shape
    out = torch.empty((B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=y.device, dtype=torch.float32, memory_format=torch.channels_last)
    check(lib().isx_bias_relu_maxpool_nhwc(y.data_ptr(), _f32(bias, "bias").data_ptr(), B, H, W, Cc, out.data_ptr(), _stream()),
          "isx_bias_relu_maxpool_nhwc")
    return out


STEM_MAX_W = 896      # csrc/stem.hip: up to four column bands of 224 input columns


def stem7x7_pool_applicable(x, conv):
    """Can `stem7x7_pool` run this convolution (the ResNet stem: 3 -> 64, 7x7, stride 2, padding 3) on x?"""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.shape[3] % 4 == 0 and x.shape[3] <= STEM_MAX_W
            and x.is_contiguous(memory_format=torch.channels_last) and conv.in_channels == 3 and conv.out_channels == 64
            and tuple(conv.kernel_size) == (7, 7) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and x.data_ptr() % 16 == 0)


def stem7x7_pool(x, w_ohwi, bias):
    """relu(conv7x7 / stride 2 / padding 3 (x) + bias) -> MaxPool2d(3, 2, 1) of a channels-last (B,3,H,W) fp32 batch as ONE kernel.
    w_ohwi: (64,7,7,3) contiguous (= conv.weight.permute(0,2,3,1)).  Returns channels-last (B,64,Hp,Wp)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.IsxError("x must be a channels-last float32 CUDA tensor (B,3,H,W)")
    _on_current_device(x, "x")
    B, _, H, W = x.shape
    w = _f32(w_ohwi, "w_ohwi")
    if tuple(w.shape) != (64, 7, 7, 3):
        raise _lib.IsxError("w_ohwi must be (64, 7, 7, 3)")
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((B, 64, Hp, Wp), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    bp = _f32(bias, "bias").data_ptr()
    _timed("isx_stem7x7_pool_nhwc", 2.0 * B * Hc * Wc * 147 * 64, 4.0 * (B * H * W * 3 + B * Hp * Wp * 64 + 147 * 64),
           lambda: check(lib().isx_stem7x7_pool_nhwc(x.data_ptr(), B, H, W, w.data_ptr(), bp, out.data_ptr(), _stream()), "isx_stem7x7_pool_nhwc"))
    return out


def conv1x1_nhwc(x, weight, bias, residual=None, relu=True):
    """1x1 stride-1 convolution of a channels-last (B,Cin,H,W) fp32 tensor with the epilogue fused:
    act(conv(x, weight) + bias (+ residual)) as ONE fp32-MFMA GEMM over the B*H*W pixels.  weight: (Cout,Cin[,1,1]).
    Returns a channels-last (B,Cout,H,W) tensor."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.IsxError("x must be a channels-last float32 CUDA tensor (B,C,H,W)")
    _on_current_device(x, "x")
    B, Cin, H, W = x.shape
    w = _f32(weight.reshape(weight.shape[0], -1), "weight")
    Cout = w.shape[0]
    if w.shape[1] != Cin:
        raise _lib.IsxError("weight must be (Cout, Cin)")
    y = torch.empty((B, Cout, H, W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    rp = 0
    if residual is not None:
        if residual.shape != y.shape or residual.dtype != torch.float32 or not residual.is_contiguous(memory_format=torch.channels_last):
            raise _lib.IsxError("residual must be a channels-last float32 tensor of the output's shape")
        rp = residual.data_ptr()
    bp, px = _f32(bias, "bias").data_ptr(), B * H * W
    _timed("isx_conv1x1_nhwc", 2.0 * px * Cin * Cout, 4.0 * (px * Cin + px * Cout * (2 if rp else 1) + Cin * Cout),
           lambda: check(lib().isx_conv1x1_nhwc(x.data_ptr(), px, Cin, w.data_ptr(), Cout, bp, rp, 1 if relu else 0, y.data_ptr(), _stream()),
                         "isx_conv1x1_nhwc"))
    return y


def conv1x1_dual_nhwc(t, x, w_cat, bias, stride=1, relu=True):
    """act(conv1x1(t, w_cat[:, :K1]) + conv1x1(x[:, :, ::stride, ::stride], w_cat[:, K1:]) + bias) in one GEMM: the last
    convolution of a bottleneck block fused with its projection shortcut.  t: (B,K1,Ho,Wo), x: (B,K2,H,W), both channels-last."""
    for a, n in ((t, "t"), (x, "x")):
        if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last)):
            raise _lib.IsxError(n + " must be a channels-last float32 CUDA tensor (B,C,H,W)")
        _on_current_device(a, n)
    B, K2, H, W = x.shape
    K1 = t.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if tuple(t.shape) != (B, K1, Ho, Wo):
        raise _lib.IsxError("t must be (B, K1, Ho, Wo) for x's shape and the stride")
    w = _f32(w_cat, "w_cat")
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, K1 + K2):
        raise _lib.IsxError("w_cat must be (Cout, K1 + K2)")
    y = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    bp, px = _f32(bias, "bias").data_ptr(), B * Ho * Wo
    _timed("isx_conv1x1_dual_nhwc", 2.0 * px * (K1 + K2) * Cout, 4.0 * (px * (K1 + K2) + px * Cout + (K1 + K2) * Cout),
           lambda: check(lib().isx_conv1x1_dual_nhwc(t.data_ptr(), K1, x.data_ptr(), B, H, W, K2, stride, w.data_ptr(), Cout, bp,
                                                     1 if relu else 0, y.data_ptr(), _stream()), "isx_conv1x1_dual_nhwc"))
    return y


def conv3x3_nhwc(x, w_ohwi, bias, stride=1, residual=None, relu=True):
    """3x3 convolution (padding 1, stride 1|2) of a channels-last (B,Cin,H,W) fp32 tensor, epilogue fused.
    w_ohwi: (Cout,3,3,Cin) contiguous (= conv.weight.permute(0,2,3,1)).  Returns channels-last (B,Cout,Ho,Wo)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.IsxError("x must be a channels-last float32 CUDA tensor (B,C,H,W)")
    _on_current_device(x, "x")
    B, Cin, H, W = x.shape
    w = _f32(w_ohwi, "w_ohwi")
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, 3, 3, Cin):
        raise _lib.IsxError("w_ohwi must be (Cout, 3, 3, Cin)")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    rp = 0
    if residual is not None:
        if residual.shape != y.shape or residual.dtype != torch.float32 or not residual.is_contiguous(memory_format=torch.channels_last):
            raise _lib.IsxError("residual must be a channels-last float32 tensor of the output's shape")
        rp = residual.data_ptr()
    bp = _f32(bias, "bias").data_ptr()
    _timed("isx_conv3x3_nhwc", 18.0 * B * Ho * Wo * Cin * Cout, 4.0 * (B * H * W * Cin + B * Ho * Wo * Cout * (2 if rp else 1) + 9 * Cin * Cout),
           lambda: check(lib().isx_conv3x3_nhwc(x.data_ptr(), B, H, W, Cin, w.data_ptr(), Cout, stride, bp, rp, 1 if relu else 0, y.data_ptr(),
                                                _stream()), "isx_conv3x3_nhwc"))
    return y


def conv3x3_expand_nhwc(x, w2_ohwi, b2, stride, w3t, b3, residual=None, relu=True):
    """conv2 + conv3 of a Bottleneck with 64 mid channels as ONE kernel: act(W3 . relu(conv3x3(x, W2) + b2) + b3 (+ residual)).
    x: channels-last (B,Cin,H,W); w2_ohwi: (64,3,3,Cin); w3t: (64,256) = conv3.weight.view(256,64).t().contiguous().
    Returns channels-last (B,256,Ho,Wo)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.IsxError("x must be a channels-last float32 CUDA tensor (B,C,H,W)")
    _on_current_device(x, "x")
    B, Cin, H, W = x.shape
    w2 = _f32(w2_ohwi, "w2_ohwi")
    w3 = _f32(w3t, "w3t")
    if tuple(w2.shape) != (64, 3, 3, Cin) or w3.dim() != 2 or w3.shape[0] != 64:
        raise _lib.IsxError("w2_ohwi must be (64, 3, 3, Cin) and w3t (64, Cout)")
    Cout = w3.shape[1]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    rp = 0
    if residual is not None:
        if residual.shape != y.shape or residual.dtype != torch.float32 or not residual.is_contiguous(memory_format=torch.channels_last):
            raise _lib.IsxError("residual must be a channels-last float32 tensor of the output's shape")
        rp = residual.data_ptr()
    b2p, b3p = _f32(b2, "b2").data_ptr(), _f32(b3, "b3").data_ptr()
    _timed("isx_conv3x3_expand_nhwc", 2.0 * B * Ho * Wo * (9 * Cin * 64 + 64 * Cout),
           4.0 * (B * H * W * Cin + B * Ho * Wo * Cout * (2 if rp else 1) + 9 * Cin * 64 + 64 * Cout),
           lambda: check(lib().isx_conv3x3_expand_nhwc(x.data_ptr(), B, H, W, Cin, w2.data_ptr(), b2p, stride, w3.data_ptr(), Cout, b3p, rp,
                                                       1 if relu else 0, y.data_ptr(), _stream()), "isx_conv3x3_expand_nhwc"))
    return y


def conv3x3_expand_dual_nhwc(t, w2_ohwi, b2, x2, wcat_t, bias, relu=True):
    """First block of a 64-mid-channel stage as ONE kernel: act([W3 | Wd] . [relu(conv3x3(t, W2) + b2) ; x2] + bias), stride 1.
    t: channels-last (B,Cin,H,W); x2: channels-last (B,64,H,W); wcat_t: (128,256) = cat([W3, Wd], 1).t().contiguous()."""
    for a, n in ((t, "t"), (x2, "x2")):
        if not (a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last)):
            raise _lib.IsxError("%s must be a channels-last float32 CUDA tensor (B,C,H,W)" % n)
        _on_current_device(a, n)
    B, Cin, H, W = t.shape
    w2 = _f32(w2_ohwi, "w2_ohwi")
    wc = _f32(wcat_t, "wcat_t")
    if tuple(w2.shape) != (64, 3, 3, Cin) or tuple(x2.shape) != (B, 64, H, W) or wc.dim() != 2 or wc.shape[0] != 128:
        raise _lib.IsxError("w2_ohwi must be (64, 3, 3, Cin), x2 (B, 64, H, W) and wcat_t (128, Cout)")
    Cout = wc.shape[1]
    y = torch.empty((B, Cout, H, W), device=t.device, dtype=torch.float32, memory_format=torch.channels_last)
    b2p, bp = _f32(b2, "b2").data_ptr(), _f32(bias, "bias").data_ptr()
    _timed("isx_conv3x3_expand_nhwc", 2.0 * B * H * W * (9 * Cin * 64 + 128 * Cout),
           4.0 * (B * H * W * (Cin + 64 + Cout) + 9 * Cin * 64 + 128 * Cout),
           lambda: check(lib().isx_conv3x3_expand_dual_nhwc(t.data_ptr(), B, H, W, Cin, w2.data_ptr(), b2p, x2.data_ptr(), wc.data_ptr(), Cout, bp,
                                                            1 if relu else 0, y.data_ptr(), _stream()), "isx_conv3x3_expand_dual_nhwc"))
    return y


def boxpool_s1(fmap, kh, kw):
    fmap = _f32(fmap, "fmap")
    B, Cc, H, W = fmap.shape
    out = torch.empty((B, Cc, H - kh + 1, W - kw + 1), device=fmap.device, dtype=torch.float32)
    check(lib().isx_boxpool_s1(fmap.data_ptr(), B, Cc, H, W, kh, kw, out.data_ptr(), _stream()), "isx_boxpool_s1")
    return out


def _is_nhwc(t):
    """A 4-d tensor whose MEMORY is channels-last and not also plain contiguous (sizes where both hold are treated as NCHW)."""
    return t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last)


def boxpool_s1_applicable_nhwc(fmap):
    return (fmap.is_cuda and fmap.dtype == torch.float32 and _is_nhwc(fmap) and fmap.shape[1] % 4 == 0
            and fmap.shape[2] * fmap.shape[3] * 16 <= 64 * 1024 and fmap.data_ptr() % 16 == 0)


def boxpool_s1_nhwc(fmap, kh, kw):
    """AvgPool2d((kh,kw), stride 1) of a channels-last (B,C,H,W) map, result channels-last: no transpose on the way in or out."""
    if not boxpool_s1_applicable_nhwc(fmap):
        raise _lib.IsxError("fmap must be a channels-last float32 CUDA tensor (B,C,H,W) with C % 4 == 0 and H*W <= 4096")
    _on_current_device(fmap, "fmap")
    B, Cc, H, W = fmap.shape
    out = torch.empty((B, Cc, H - kh + 1, W - kw + 1), device=fmap.device, dtype=torch.float32, memory_format=torch.channels_last)
    check(lib().isx_boxpool_s1_nhwc(fmap.data_ptr(), B, Cc, H, W, kh, kw, out.data_ptr(), _stream()), "isx_boxpool_s1_nhwc")
    return out


def best_location_desc(cls, eps=EPS):
    """cls (B,K,Hp,Wp) class-score maps -> (desc (B,K), loc (B,2)); a channels-last tensor is consumed in place."""
    if not (isinstance(cls, torch.Tensor) and cls.is_cuda and cls.dtype == torch.float32):
        raise _lib.IsxError("cls must be a float32 CUDA tensor (libisx has no CPU path)")
    _on_current_device(cls, "cls")
    B, K, Hp, Wp = cls.shape
    desc = torch.empty((B, K), device=cls.device, dtype=torch.float32)
    loc = torch.empty((B, 2), device=cls.device, dtype=torch.int64)
    if _is_nhwc(cls):
        check(lib().isx_best_location_desc_nhwc(cls.data_ptr(), B, K, Hp, Wp, eps, desc.data_ptr(), loc.data_ptr(), _stream()),
              "isx_best_location_desc_nhwc")
        return desc, loc
    cls = cls.contiguous()
    check(lib().isx_best_location_desc(cls.data_ptr(), B, K, Hp, Wp, eps, desc.data_ptr(), loc.data_ptr(), _stream()),
          "isx_best_location_desc")
    return desc, loc


def region_topk(cls, k):
    """Canonical top-k locations of the class-max map.  cls (K,Hp,Wp) -> (idx (k), score (k));
    cls (B,K,Hp,Wp) -> (idx (B,k), score (B,k))."""
    if not (isinstance(cls, torch.Tensor) and cls.is_cuda and cls.dtype == torch.float32):
        raise _lib.IsxError("cls must be a float32 CUDA tensor (libisx has no CPU path)")
    _on_current_device(cls, "cls")
    single = cls.dim() == 3
    if single:
        cls = cls.unsqueeze(0)
    B, K, Hp, Wp = cls.shape
    idx = torch.empty((B, k), device=cls.device, dtype=torch.int64)
    sc = torch.empty((B, k), device=cls.device, dtype=torch.float32)
    if _is_nhwc(cls):                  # channels-last score map (1x1-convolution classifier on the NHWC trunk): consumed in place
        check(lib().isx_region_topk_nhwc(cls.data_ptr(), B, K, Hp, Wp, k, idx.data_ptr(), sc.data_ptr(), _stream()), "isx_region_topk_nhwc")
    else:
        cls = cls.contiguous()
        check(lib().isx_region_topk(cls.data_ptr(), B, K, Hp, Wp, k, idx.data_ptr(), sc.data_ptr(), _stream()), "isx_region_topk")
    return (idx[0], sc[0]) if single else (idx, sc)


def region_gather_l2_nhwc(fmap, kh, kw, flat_idx, Wp, shift_hwc=None, eps=EPS):
    """Window gather + L2 + Shift on a channels-last (B,C,Hf,Wf) map; rows (B,k,kh*kw*C) keep the window in (h,w,C) order
    (row[(a*kw + b)*C + c]); shift_hwc: the Shift parameter in that order."""
    if not (fmap.is_cuda and fmap.dtype == torch.float32 and _is_nhwc(fmap) and fmap.shape[1] % 4 == 0):
        raise _lib.IsxError("fmap must be a channels-last float32 CUDA tensor (B,C,Hf,Wf) with C % 4 == 0")
    _on_current_device(fmap, "fmap")
    flat_idx = _typed(flat_idx, torch.int64, "flat_idx")
    B, Cc, Hf, Wf = fmap.shape
    k = flat_idx.size(1)
    assert flat_idx.shape == (B, k)
    rows = torch.empty((B, k, Cc * kh * kw), device=fmap.device, dtype=torch.float32)
    sp = 0
    if shift_hwc is not None:
        shift_hwc = _f32(shift_hwc, "shift_hwc")
        assert shift_hwc.numel() == Cc * kh * kw
        sp = shift_hwc.data_ptr()
    check(lib().isx_region_gather_l2_nhwc(fmap.data_ptr(), B, Cc, Hf, Wf, kh, kw, flat_idx.data_ptr(), k, Wp, sp, eps, rows.data_ptr(), _stream()),
          "isx_region_gather_l2_nhwc")
    return rows


def region_gather_l2(fmap, kh, kw, flat_idx, Wp, shift=None, eps=EPS):
    """Window gather + L2 + Shift.  fmap (C,Hf,Wf), flat_idx (k) -> rows (k,F); batched: fmap (B,C,Hf,Wf),
    flat_idx (B,k) -> rows (B,k,F)."""
    fmap = _f32(fmap, "fmap")
    flat_idx = _typed(flat_idx, torch.int64, "flat_idx")
    single = fmap.dim() == 3
    if single:
        fmap, flat_idx = fmap.unsqueeze(0), flat_idx.reshape(1, -1)
    B, Cc, Hf, Wf = fmap.shape
    k = flat_idx.size(1)
    assert flat_idx.shape == (B, k)
    rows = torch.empty((B, k, Cc * kh * kw), device=fmap.device, dtype=torch.float32)
    sp = 0
    if shift is not None:
        shift = _f32(shift, "shift")
        sp = shift.data_ptr()
    check(lib().isx_region_gather_l2(fmap.data_ptr(), B, Cc, Hf, Wf, kh, kw, flat_idx.data_ptr(), k, Wp, sp, eps,
                                     rows.data_ptr(), _stream()), "isx_region_gather_l2")
    return rows[0] if single else rows


def cosine_sim(Q, G, out=None):
    Q, G = _f32(Q, "Q"), _f32(G, "G")
    M, D = Q.shape
    N = G.shape[0]
    assert G.shape[1] == D
    sim = torch.empty((M, N), device=Q.device, dtype=torch.float32) if out is None else out
    check(lib().isx_cosine_sim(Q.data_ptr(), M, G.data_ptr(), N, D, sim.data_ptr(), _stream()), "isx_cosine_sim")
    return sim


def cosine_topk_workspace(M, N, D, k):
    return lib().isx_cosine_topk_workspace(M, N, D, k)


def cosine_topk(Q, G, k, idx_base=0, ws=None, out=None):
    """(top_score (M,k) f32, top_idx (M,k) i64), canonical order.  `ws`: optional uint8 CUDA
    tensor reused across calls (sized by cosine_topk_workspace or larger/smaller)."""
    Q, G = _f32(Q, "Q"), _f32(G, "G")
    M, D = Q.shape
    N = G.shape[0]
    assert G.shape[1] == D
    if ws is None:
        ws = torch.empty((cosine_topk_workspace(M, N, D, k),), device=Q.device, dtype=torch.uint8)
    if out is None:
        ts = torch.empty((M, k), device=Q.device, dtype=torch.float32)
        ti = torch.empty((M, k), device=Q.device, dtype=torch.int64)
    else:
        ts, ti = out
    check(lib().isx_cosine_topk(Q.data_ptr(), M, G.data_ptr(), N, D, k, idx_base, ts.data_ptr(), ti.data_ptr(),
                                ws.data_ptr(), ws.numel(), _stream()), "isx_cosine_topk")
    return ts, ti


def gallery_to_f16(G):
    """(Gh (N,D) fp16 scaled image, gstats (4,) f32: max |g|^2, max |G|, max conversion loss^2, 0) -- the cached gallery side of cosine_topk_fast."""
    G = _f32(G, "G")
    N, D = G.shape
    Gh = torch.empty((N, D), device=G.device, dtype=torch.float16)
    gstats = torch.empty((4,), device=G.device, dtype=torch.float32)
    check(lib().isx_gallery_to_f16(G.data_ptr(), N, D, Gh.data_ptr(), gstats.data_ptr(), _stream()), "isx_gallery_to_f16")
    return Gh, gstats


def cosine_topk_fast_workspace(M, N, D, k, have_gallery_f16=False):
    return lib().isx_cosine_topk_fast_workspace(M, N, D, k, 1 if have_gallery_f16 else 0)


def cosine_topk_fast_fallback_counter(ws, M, N, D, k, have_gallery_f16=False):
    """int32 view (1 element, on the device) of the fallback-row counter inside a cosine_topk_fast workspace, or None when a
    call of this shape runs the fp32 search as a whole.  Valid once the search has completed on its stream."""
    off = lib().isx_cosine_topk_fast_fallback_offset(M, N, D, k, 1 if have_gallery_f16 else 0)
    if off == (1 << 64) - 1 or off + 4 > ws.numel():
        return None
    return ws[off:off + 4].view(torch.int32)


def cosine_topk_fast(Q, G, k, idx_base=0, gallery_f16=None, ws=None, out=None):
    """Same result as cosine_topk, bit for bit; fp16-MFMA filter + exact fp32 re-scoring.
    gallery_f16: optional (Gh, gstats) from gallery_to_f16(G)."""
    Q, G = _f32(Q, "Q"), _f32(G, "G")
    M, D = Q.shape
    N = G.shape[0]
    assert G.shape[1] == D
    gh = gs = 0
    if gallery_f16 is not None:
        Gh, gstats = gallery_f16
        assert Gh.dtype == torch.float16 and Gh.shape == G.shape and Gh.is_contiguous() and gstats.numel() == 4
        gh, gs = Gh.data_ptr(), gstats.data_ptr()
    if ws is None:
        ws = torch.empty((cosine_topk_fast_workspace(M, N, D, k, gallery_f16 is not None),), device=Q.device, dtype=torch.uint8)
    if out is None:
        ts = torch.empty((M, k), device=Q.device, dtype=torch.float32)
        ti = torch.empty((M, k), device=Q.device, dtype=torch.int64)
    else:
        ts, ti = out
    check(lib().isx_cosine_topk_fast(Q.data_ptr(), M, G.data_ptr(), N, D, k, idx_base, gh, gs, ts.data_ptr(), ti.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream()), "isx_cosine_topk_fast")
    return ts, ti


def _row_segments(M, N, k):
    """Segments per row for a FEW-ROW top-k (0 = one launch over whole rows).  `isx_topk_rows` gives every row one workgroup: with fewer rows
    than the chip holds workgroups (1 000 queries x 100 000 gallery rows: 0.43 ms, 0.9 TB/s) the launch is latency-bound.  A row-major
    (M, N) matrix IS an (M S, N / S) matrix when S divides N, so the same kernel selects per segment and `isx_topk_merge` (canonical
    comparator on the restored column indices) merges the S lists of a row: same result, bit for bit."""
    if M >= 4096 or N < 16384 or k > 256:
        return 0
    target = max(2, 8192 // max(M, 1))               # one wave per segment: ~8 waves per SIMD fill the chip; more only adds merge work
    best = 0
    for S in range(2, min(64, 4096 // max(k, 1)) + 1):
        if N % S == 0 and N // S >= max(k, 1024) and (best == 0 or abs(S - target) < abs(best - target)):
            best = S
    return best


def topk_rows(sim, k, idx_base=0):
    sim = _f32(sim, "sim")
    M, N = sim.shape
    S = _row_segments(M, N, k)
    if S:
        L = N // S
        ts = torch.empty((M * S, k), device=sim.device, dtype=torch.float32)
        ti = torch.empty((M * S, k), device=sim.device, dtype=torch.int64)
        check(lib().isx_topk_rows(sim.data_ptr(), M * S, L, k, 0, ts.data_ptr(), ti.data_ptr(), _stream()), "isx_topk_rows")
        ti = ti.view(M, S, k) + (torch.arange(S, device=sim.device, dtype=torch.int64) * L + idx_base).view(1, S, 1)
        return topk_merge(ts.view(M, S, k).permute(1, 0, 2).contiguous(), ti.permute(1, 0, 2).contiguous())
    ts = torch.empty((M, k), device=sim.device, dtype=torch.float32)
    ti = torch.empty((M, k), device=sim.device, dtype=torch.int64)
    check(lib().isx_topk_rows(sim.data_ptr(), M, N, k, idx_base, ts.data_ptr(), ti.data_ptr(), _stream()), "isx_topk_rows")
    return ts, ti


def rank_full(sim):
    sim = _f32(sim, "sim")
    M, N = sim.shape
    ranked = torch.empty((M, N), device=sim.device, dtype=torch.int64)
    nb = lib().isx_rank_full_workspace(M, N)
    ws = torch.empty((nb,), device=sim.device, dtype=torch.uint8)
    check(lib().isx_rank_full(sim.data_ptr(), M, N, ranked.data_ptr(), ws.data_ptr(), nb, _stream()), "isx_rank_full")
    return ranked


def average_precision(ranked, qlab, glab, kth=1):
    ranked = _typed(ranked, torch.int64, "ranked")
    M, N = ranked.shape
    qlab = _typed(qlab, torch.int32, "qlab")
    glab = _typed(glab, torch.int32, "glab")
    ap = torch.empty((M,), device=ranked.device, dtype=torch.float64)
    check(lib().isx_average_precision(ranked.data_ptr(), M, N, qlab.data_ptr(), glab.data_ptr(), kth, ap.data_ptr(),
                                      _stream()), "isx_average_precision")
    return ap


def average_precision_sim(sim, qlab, glab, kth=1):
    """AP per query straight from the score matrix (no sort).  Rows with more than 2048 positives are
    recomputed through rank_full + average_precision, so the result always equals the sorted path."""
    sim = _f32(sim, "sim")
    M, N = sim.shape
    qlab = _typed(qlab, torch.int32, "qlab")
    glab = _typed(glab, torch.int32, "glab")
    ap = torch.empty((M,), device=sim.device, dtype=torch.float64)
    check(lib().isx_average_precision_sim(sim.data_ptr(), M, N, qlab.data_ptr(), glab.data_ptr(), kth, ap.data_ptr(),
                                          _stream()), "isx_average_precision_sim")
    heavy = (ap == -1.0).nonzero().flatten()
    if heavy.numel():
        ap[heavy] = average_precision(rank_full(sim[heavy]), qlab[heavy], glab, kth)
    return ap


def ap_shard_max_positives():
    return int(lib().isx_ap_shard_max_positives())


def ap_shard_positives(sim, idx_base, qlab, glab, cap=None):
    """Step 1 of the sharded average precision (include/isx.h): (keys (M, cap) int64 bit patterns of the canonical uint64 keys, 0 = empty;
    count (M,) int32) of this shard's positives per query.  sim: (M, Ns) scores against the shard's rows, idx_base: the shard's first global row."""
    sim = _f32(sim, "sim")
    M, N = sim.shape
    cap = ap_shard_max_positives() if cap is None else int(cap)
    qlab, glab = _typed(qlab, torch.int32, "qlab"), _typed(glab, torch.int32, "glab")
    keys = torch.empty((M, cap), dtype=torch.int64, device=sim.device)
    count = torch.empty((M,), dtype=torch.int32, device=sim.device)
    check(lib().isx_ap_shard_positives(sim.data_ptr(), M, N, int(idx_base), qlab.data_ptr(), glab.data_ptr(), cap, keys.data_ptr(), count.data_ptr(), _stream()),
          "isx_ap_shard_positives")
    return keys, count


def ap_shard_hist(sim, idx_base, keys_all):
    """Step 2: this shard's bucket counts (M, ap_shard_max_positives()) int32 against the gathered keys of all shards (M, W)."""
    sim = _f32(sim, "sim")
    M, N = sim.shape
    keys_all = _typed(keys_all, torch.int64, "keys_all")
    assert keys_all.dim() == 2 and keys_all.size(0) == M
    hist = torch.empty((M, ap_shard_max_positives()), dtype=torch.int32, device=sim.device)
    check(lib().isx_ap_shard_hist(sim.data_ptr(), M, N, int(idx_base), keys_all.data_ptr(), int(keys_all.size(1)), hist.data_ptr(), _stream()), "isx_ap_shard_hist")
    return hist


def ap_from_hist(hist, n_lab, kth=1):
    """Step 3: float64 AP per query from the histogram summed over the shards (M, ld) and the positives per query over all shards (M,)."""
    hist, n_lab = _typed(hist, torch.int32, "hist"), _typed(n_lab, torch.int32, "n_lab")
    M = hist.size(0)
    ap = torch.empty((M,), dtype=torch.float64, device=hist.device)
    check(lib().isx_ap_from_hist(hist.data_ptr(), int(hist.size(1)), n_lab.data_ptr(), M, int(kth), ap.data_ptr(), _stream()), "isx_ap_from_hist")
    return ap


def masked_sums(sim, qlab, glab):
    sim = _f32(sim, "sim")
    M, N = sim.shape
    qlab = _typed(qlab, torch.int32, "qlab")
    glab = _typed(glab, torch.int32, "glab")
    out = torch.empty((M, 2), device=sim.device, dtype=torch.float64)
    check(lib().isx_masked_sums(sim.data_ptr(), M, N, qlab.data_ptr(), glab.data_ptr(), out.data_ptr(), _stream()),
          "isx_masked_sums")
    return out


def tree_sum_rows(rows, out=None):
    """Sum of the rows of a (L, n) fp32 device tensor in the canonical tree order of isx/dp.py (tree_sum), one kernel (isx_tree_sum_rows);
    L <= 16.  `out`: an (n,) tensor to write (may be rows[0])."""
    rows = _f32(rows, "rows")
    L, n = rows.shape
    if rows.stride(1) != 1:
        raise _lib.IsxError("tree_sum_rows: rows must be contiguous along the last dimension")
    if out is None:
        out = torch.empty((n,), device=rows.device, dtype=torch.float32)
    check(lib().isx_tree_sum_rows(rows.data_ptr(), int(L), int(rows.stride(0)) if L > 1 else int(n), int(n), out.data_ptr(), _stream()), "isx_tree_sum_rows")
    return out


DBA_MAX_GROUP = 1024


def dba_groups(emb, order, grp_begin, grp_size, max_group, k=-1):
    """DBA (reference test/instance_avg.py:7-33) over same-instance groups of at most DBA_MAX_GROUP items; see include/isx.h."""
    emb = _f32(emb, "emb")
    N, D = emb.shape
    order, grp_begin, grp_size = (_typed(t, torch.int32, n) for t, n in ((order, "order"), (grp_begin, "grp_begin"), (grp_size, "grp_size")))
    assert order.numel() == N and grp_begin.numel() == N and grp_size.numel() == N
    out = torch.empty_like(emb)
    check(lib().isx_dba_groups(emb.data_ptr(), N, D, order.data_ptr(), grp_begin.data_ptr(), grp_size.data_ptr(), int(max_group), int(k),
                               out.data_ptr(), _stream()), "isx_dba_groups")
    return out


def topk_merge(scores, idx):
    scores = _f32(scores, "scores")
    idx = _typed(idx, torch.int64, "idx")
    P, M, k = scores.shape
    os_ = torch.empty((M, k), device=scores.device, dtype=torch.float32)
    oi = torch.empty((M, k), device=scores.device, dtype=torch.int64)
    check(lib().isx_topk_merge(scores.data_ptr(), idx.data_ptr(), P, M, k, os_.data_ptr(), oi.data_ptr(), _stream()),
          "isx_topk_merge")
    return os_, oi


# ---- native RCCL exchange of per-shard top-k lists (lazy-bound librccl; see csrc/comm.cpp) -------------
def comm_unique_id():
    """bytes of an ncclUniqueId (create on rank 0, ship to the other ranks by any means)."""
    import ctypes
    n = lib().isx_comm_unique_id_bytes()
    buf = ctypes.create_string_buffer(n)
    check(lib().isx_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "isx_comm_unique_id")
    return bytes(buf.raw)


def comm_init_rank(nranks, rank, unique_id):
    """ncclComm_t handle (an int) of this rank; the current CUDA device must already be set."""
    import ctypes
    handle = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(unique_id, len(unique_id))
    check(lib().isx_comm_init_rank(ctypes.byref(handle), nranks, rank, ctypes.cast(buf, ctypes.c_void_p)), "isx_comm_init_rank")
    return handle.value


def comm_destroy(comm):
    check(lib().isx_comm_destroy(comm), "isx_comm_destroy")


def shard_topk_allgather(comm, nranks, s_local, i_local):
    """(P,M,k) scores and indices gathered from every rank (rank-major), on the current stream."""
    s_local = _f32(s_local, "s_local")
    i_local = _typed(i_local, torch.int64, "i_local")
    M, k = s_local.shape
    all_s = torch.empty((nranks, M, k), device=s_local.device, dtype=torch.float32)
    all_i = torch.empty((nranks, M, k), device=s_local.device, dtype=torch.int64)
    check(lib().isx_shard_topk_allgather(comm, s_local.data_ptr(), i_local.data_ptr(), M, k, all_s.data_ptr(), all_i.data_ptr(),
                                         _stream()), "isx_shard_topk_allgather")
    return all_s, all_i


def comm_allgather_rows(comm, nranks, rows_local, out=None):
    """(P * rows, D) descriptor rows of every rank (rank-major) on libisx's communicator, on the current stream."""
    rows_local = _f32(rows_local, "rows_local")
    R, D = rows_local.shape
    if out is None:
        out = torch.empty((nranks * R, D), device=rows_local.device, dtype=torch.float32)
    elif tuple(out.shape) != (nranks * R, D) or out.dtype != torch.float32 or not out.is_contiguous():
        raise _lib.IsxError("comm_allgather_rows: out must be a contiguous fp32 (%d, %d)" % (nranks * R, D))
    check(lib().isx_comm_allgather_rows(comm, rows_local.data_ptr(), R, D, out.data_ptr(), _stream()), "isx_comm_allgather_rows")
    return out


# ---- training step -----------------------------------------------------------------------
def mine_negatives(sim, labels, i1, i2, semi_hard, row_base=0):
    """neg index per positive couple (int64, -1 = none available).  sim: the N x N matrix, or rows
    [row_base, row_base + sim.size(0)) of it (every anchor i1 inside that range; i1 / i2 are absolute indices)."""
    sim = _f32(sim, "sim")
    rows, N = sim.shape
    labels = _typed(labels, torch.int32, "labels")
    i1 = _typed(i1, torch.int64, "i1")
    i2 = _typed(i2, torch.int64, "i2")
    neg = torch.empty_like(i1)
    if rows == N and row_base == 0:
        check(lib().isx_mine_negatives(sim.data_ptr(), N, labels.data_ptr(), i1.data_ptr(), i2.data_ptr(), i1.numel(),
                                       1 if semi_hard else 0, neg.data_ptr(), _stream()), "isx_mine_negatives")
        return neg
    if i1.numel() and not (int(i1.min()) >= row_base and int(i1.max()) < row_base + rows):
        raise _lib.IsxError("mine_negatives: anchors outside the row block [%d, %d)" % (row_base, row_base + rows))
    check(lib().isx_mine_negatives_rows(sim.data_ptr(), N, row_base, rows, labels.data_ptr(), i1.data_ptr(), i2.data_ptr(), i1.numel(),
                                        1 if semi_hard else 0, neg.data_ptr(), _stream()), "isx_mine_negatives_rows")
    return neg


def triplet_loss_rows(anchor, pos, neg, margin, normalized=True):
    anchor, pos, neg = _f32(anchor, "anchor"), _f32(pos, "pos"), _f32(neg, "neg")
    B, D = anchor.shape
    rows = torch.empty((B,), device=anchor.device, dtype=torch.float32)
    check(lib().isx_triplet_loss_fwd(anchor.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, D, margin, 1 if normalized else 0,
                                     rows.data_ptr(), _stream()), "isx_triplet_loss_fwd")
    return rows


def triplet_loss_grads(anchor, pos, neg, loss_rows, scale, normalized=True, scale_dev=None):
    """scale_dev: optional 1-element float32 CUDA tensor (autograd's grad_output) multiplied in on the device -- no host synchronisation."""
    anchor, pos, neg = _f32(anchor, "anchor"), _f32(pos, "pos"), _f32(neg, "neg")
    B, D = anchor.shape
    ga, gp, gn = torch.empty_like(anchor), torch.empty_like(anchor), torch.empty_like(anchor)
    if scale_dev is not None:
        sd_ = _f32(scale_dev.reshape(-1), "scale_dev")
        check(lib().isx_triplet_loss_bwd_dev(anchor.data_ptr(), pos.data_ptr(), neg.data_ptr(), _f32(loss_rows, "loss_rows").data_ptr(), B, D,
                                             float(scale), sd_.data_ptr(), 1 if normalized else 0, ga.data_ptr(), gp.data_ptr(), gn.data_ptr(), _stream()),
              "isx_triplet_loss_bwd_dev")
        return ga, gp, gn
    check(lib().isx_triplet_loss_bwd(anchor.data_ptr(), pos.data_ptr(), neg.data_ptr(), _f32(loss_rows, "loss_rows").data_ptr(), B, D,
                                     float(scale), 1 if normalized else 0, ga.data_ptr(), gp.data_ptr(), gn.data_ptr(), _stream()),
          "isx_triplet_loss_bwd")
    return ga, gp, gn


def triplet_leaves(d_all, leaves, margin, normalized=True, scale_a=1.0, scale_b=1.0):
    """Triplet loss + gradient of every micro-batch of a step in one launch (isx_triplet_leaves).  d_all: (leaves * 3 k, D), per leaf the anchor,
    positive and negative rows.  Returns (per-leaf sum of the row losses (leaves,), gradient rows like d_all scaled by scale_a * scale_b)."""
    d_all = _f32(d_all, "d_all")
    R, D = d_all.shape
    if leaves <= 0 or R % (3 * leaves):
        raise _lib.IsxError("triplet_leaves: %d rows do not split into %d leaves of 3 k rows" % (R, leaves))
    k = R // (3 * leaves)
    loss = torch.empty((leaves,), device=d_all.device, dtype=torch.float32)
    dd = torch.empty_like(d_all)
    check(lib().isx_triplet_leaves(d_all.data_ptr(), leaves, k, D, float(margin), 1 if normalized else 0, float(scale_a), float(scale_b),
                                   loss.data_ptr(), dd.data_ptr(), _stream()), "isx_triplet_leaves")
    return loss, dd


# ---- half-precision filter path (csrc/fast.hip) ----------------------------------------------------
def rows_to_f16(x):
    """(h (B,D) float16, norm2 (B) upper bound of the squared row norm, amax (B) max |x|)."""
    x = _f32(x, "x")
    B, D = x.shape
    h = torch.empty((B, D), device=x.device, dtype=torch.float16)
    n2 = torch.empty((B,), device=x.device, dtype=torch.float32)
    am = torch.empty((B,), device=x.device, dtype=torch.float32)
    check(lib().isx_rows_to_f16(x.data_ptr(), B, D, h.data_ptr(), n2.data_ptr(), am.data_ptr(), _stream()), "isx_rows_to_f16")
    return h, n2, am


def cosine_sim_f16(Qh, Gh, out=None):
    assert Qh.dtype == torch.float16 and Gh.dtype == torch.float16 and Qh.is_cuda and Gh.is_cuda
    Qh, Gh = Qh.contiguous(), Gh.contiguous()
    M, D = Qh.shape
    N = Gh.shape[0]
    sim = torch.empty((M, N), device=Qh.device, dtype=torch.float32) if out is None else out
    check(lib().isx_cosine_sim_f16(Qh.data_ptr(), M, Gh.data_ptr(), N, D, sim.data_ptr(), _stream()), "isx_cosine_sim_f16")
    return sim
