"""The four network wrappers of the reference (model/siamese.py) with the same constructor
signatures, attributes and state-dict keys:

  TuneClassif          (:10-54)    classifier fine-tuned from a backbone
  TuneClassifSub       (:57-89)    same, fully convolutional: a class-score MAP per image
  DescriptorNet        (:92-130)   global siamese descriptor: features -> L2 -> Shift -> Linear -> L2
  RegionDescriptorNet  (:133-231)  descriptor summed over the k best-classified windows

Inference (no autograd graph, GPU): the BN-folded trunk and everything after the last conv run in the
hand-written HIP kernels of libisx.  Training on the GPU: the FROZEN PREFIX of the trunk (the reference trains
layer4 only, train/*_p.py:14-17) still runs as that folded HIP trunk, the trainable suffix and the head carry
the autograd graph (_SplitTrunk).  On the CPU the same arithmetic runs as plain torch modules.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .custom_modules import NormalizeL2, RowDeferredLinear, Shift
from .nn_utils import convolutionalize, extract_layers, get_feature_size, set_untrained_blocks


def _fast(x, trainable=None):
    """HIP path: GPU tensor and no autograd graph needed -- neither through x nor (while gradients are being recorded)
    into the parameters of `trainable`, the module about to consume x (its fast path reads them detached)."""
    if not x.is_cuda:
        return False
    if not torch.is_grad_enabled():
        return True
    return not x.requires_grad and not (trainable is not None and any(p.requires_grad for p in trainable.parameters()))


class BoxPool(nn.AvgPool2d):
    """nn.AvgPool2d(feature_size2d, stride=1) (reference model/siamese.py:67-71, 166-170) backed
    by libisx `isx_boxpool_s1` on the GPU."""

    def forward(self, x):
        kh, kw = self.kernel_size if isinstance(self.kernel_size, tuple) else (self.kernel_size,) * 2
        # stride 1, or a window that spans the whole map (one output: the stride is never used -- TuneClassif's AvgPool2d(7) on a 7 x 7 map)
        if _fast(x) and x.dim() == 4 and (self.stride in (1, (1, 1)) or tuple(x.shape[2:]) == (kh, kw)) and self.padding in (0, (0, 0)) and not self.ceil_mode:
            from isx import ops
            if ops.boxpool_s1_applicable_nhwc(x):
                return ops.boxpool_s1_nhwc(x, kh, kw)        # channels-last trunk output: pooled in place, stays channels-last
            return ops.boxpool_s1(x.float(), kh, kw)
        return super().forward(x)


class PointwiseConv(nn.Conv2d):
    """A convolutionalised Linear with a 1x1 window (reference model/nn_utils.py:26-39 via model/siamese.py:73-80): plain nn.Conv2d
    (same parameters and state-dict keys); on a channels-last GPU map without autograd it is ONE libisx GEMM over the locations
    (`isx_conv1x1_nhwc`, bias fused), so the class-score map is born channels-last for the region kernels behind it."""

    def forward(self, x):
        if (_fast(x, self) and x.dtype == torch.float32 and x.dim() == 4 and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last) and self.kernel_size == (1, 1) and self.stride == (1, 1)
                and self.padding == (0, 0) and self.dilation == (1, 1) and self.groups == 1 and self.in_channels % 4 == 0):
            from isx import ops
            bias = self.bias if self.bias is not None else x.new_zeros(self.out_channels)
            return ops.conv1x1_nhwc(x, self.weight.detach(), bias.detach(), None, relu=False)
        return super().forward(x)


def _pool_factor(reduc):
    f = 1
    for m in reduc:
        ks = m.kernel_size
        f *= ks[0] * ks[1] if isinstance(ks, tuple) else ks * ks
    return f


def _convolutionalize_classifier(classifier, feature_size2d, has_reduc):
    """Every Linear becomes a conv; only the first one, and only when no pooling precedes it
    (AlexNet-like), takes the full feature window -- the others are 1x1."""
    seen = 0
    for name, m in list(classifier._modules.items()):
        if isinstance(m, nn.Linear):
            size2d = (1, 1) if (has_reduc or seen > 0) else tuple(feature_size2d)
            conv = convolutionalize(m, size2d)
            if size2d == (1, 1):
                conv.__class__ = PointwiseConv              # same module, GEMM fast path on channels-last GPU maps
            classifier._modules[name] = conv
            seen += 1


class TuneClassif(nn.Module):
    def __init__(self, net, num_classes, untrained=-1, reduc=True):
        super().__init__()
        self.features, self.feature_reduc, self.classifier = extract_layers(net)
        set_untrained_blocks([self.features, self.classifier], untrained)
        # the last classifier module becomes Linear(in, num_classes) unless it already is one
        names = list(self.classifier._modules)
        last = self.classifier._modules[names[-1]]
        if not isinstance(last, nn.Linear) or last.out_features != num_classes:
            self.classifier._modules[names[-1]] = nn.Linear(last.in_features, num_classes)
        self.feature_size = num_classes
        if not reduc:
            # drop the spatial reduction: the first FC then sees every location
            factor = _pool_factor(self.feature_reduc)
            first = self.classifier._modules[names[0]]
            self.classifier._modules[names[0]] = nn.Linear(first.in_features * factor, first.out_features)
            self.feature_reduc = nn.Sequential()
        _rows_linears(self.classifier)
        _box_pools(self.feature_reduc)

    def forward(self, x):
        x = self.features(x)
        x = self.feature_reduc(x)
        return self.classifier(x.reshape(x.size(0), -1))     # logical (C,h,w) order whatever the memory format


class TuneClassifSub(TuneClassif):
    def __init__(self, net, num_classes, feature_size2d, untrained=-1):
        super().__init__(net, num_classes, untrained, reduc=True)
        has_reduc = len(self.feature_reduc) > 0
        if has_reduc:
            self.feature_reduc = nn.Sequential(BoxPool(tuple(feature_size2d), stride=1))
        _convolutionalize_classifier(self.classifier, feature_size2d, has_reduc)

    def forward_single(self, x):
        return self.classifier(self.feature_reduc(self.features(x)))

    def forward(self, *scales):
        return [self.forward_single(x) for x in scales]


# A/B switch: ISX_SPLIT_TRUNK=0 (or model.siamese.SPLIT_TRUNK = False) runs the training trunk as plain `features(x)` -- the
# "plain torch run" the split trunk is tested against.
SPLIT_TRUNK = os.environ.get("ISX_SPLIT_TRUNK", "1") != "0"
# A/B switch: ISX_SUFFIX_ENGINE=0 keeps the trainable suffix on the plain modules + torch autograd (MIOpen) behind the HIP prefix.
SUFFIX_ENGINE = os.environ.get("ISX_SUFFIX_ENGINE", "1") != "0"
# A/B switch: ISX_HEAD_ENGINE=0 keeps the descriptor head of a training step on torch autograd, micro-batch by micro-batch.
HEAD_ENGINE = os.environ.get("ISX_HEAD_ENGINE", "1") != "0"
PREFIX_CACHE_BUDGET = int(os.environ.get("ISX_PREFIX_CACHE_GB", "96")) << 30      # HBM the prefix-feature cache of ONE resident set may take


def first_trainable(features):
    """Index of the first module of `features` that holds a trainable parameter (len(features): the whole trunk is frozen).
    set_untrained_blocks freezes a PREFIX of the parameterised modules (reference model/nn_utils.py:5-23), so everything in front
    of this index needs no autograd graph."""
    for i, m in enumerate(features):
        if any(p.requires_grad for p in m.parameters()):
            return i
    return len(features)


class _SplitTrunk(object):
    """Siamese training on the GPU: the trunk is cut at the first trainable block (the reference's configurations train layer4 of a
    ResNet / conv5 of AlexNet: train/*_p.py:14-17,48; `untrained = -1` freezes everything).

      frozen prefix   -- needs no autograd graph and cannot change under the optimiser, so it runs as the BN-folded inference trunk with
                         the hand-written convolution kernels (model/nn_utils.fold_batch_norm) under no_grad, built once;
      trainable suffix -- the original modules with autograd on the prefix's output (channels-last).

    With BatchNorm learning (P.train_bn) or on the CPU the trunk is the plain `features(x)`."""

    def __init__(self):
        self.folded = None
        self.key = None
        self.split = 0
        self.cache = {}                 # id(resident set) -> [set, features (N, C, h, w) or None, have (N,) bool]: prefix features by image row
        self.cache_stats = {"rows_served": 0, "rows_computed": 0}

    @staticmethod
    def enabled(features):
        return SPLIT_TRUNK and not _bn_training(features) and first_trainable(features) > 0

    def usable(self, features, x):
        return x.is_cuda and x.dtype == torch.float32 and self.enabled(features)

    def prefix(self, features, x):
        """frozen prefix of `features` on x -> (feature tensor without graph, index the suffix starts at)"""
        split = self._refresh(features, x)
        with torch.no_grad():
            return self.folded(x.contiguous(memory_format=torch.channels_last)), split

    def _refresh(self, features, x):
        """(Re)build the folded prefix when its weights / buffers / split changed; returns the split index."""
        split = first_trainable(features)
        mods = list(features)[:split]
        # the folded copy is stale as soon as any weight or BN buffer of the prefix is written in place (load_state_dict) or replaced, or
        # the split moves: identity + version counter of every prefix tensor are the key (no state_dict() is built per call; writes
        # through `.data` bypass the counters -- set `self.folded = None` after such surgery)
        key = (id(features), str(x.device), split,
               tuple((id(t), t._version) for m in mods for t in m.parameters()), tuple((id(t), t._version) for m in mods for t in m.buffers()))
        if self.folded is None or self.key != key:
            from .nn_utils import fold_batch_norm
            self.folded = fold_batch_norm(nn.Sequential(*mods)).to(x.device).to(memory_format=torch.channels_last)
            self.key, self.split = key, split
            self.cache = {}             # features of another prefix
        return split

    def prefix_cached(self, features, xs):
        """Prefix features of image batches that are ROWS OF A RESIDENT SET (train/_common.ResidentImages.gather leaves the provenance on the
        tensor): looked up in an HBM table of the set's prefix features, the rows not seen yet computed first (launches of up to 512 images).
        A frozen prefix with frozen BatchNorm maps an image to the same bits whatever batch it rides in (the kernels are batch-invariant:
        tests/test_gpu_dropin.py), the preprocessed images do not change between steps, and an epoch of the reference's training touches every
        image a dozen times (36 steps x 192 images from a 512-image set; utils/train_general.py:51-74 recomputes the trunk per micro-batch): the
        table removes the repeats, bit for bit the same training.  288 GB of HBM hold the 0.8 MB per image of ~100 k images next to everything
        else (PREFIX_CACHE_BUDGET).  None: not applicable (images of no / several sets, table over budget)."""
        srcs = [getattr(x, "_isx_rows", None) for x in xs]
        if any(s is None for s in srcs) or len(set(id(s[0]) for s in srcs)) != 1:
            return None
        R = srcs[0][0]
        idx = torch.cat([s[1] for s in srcs])
        self._refresh(features, xs[0])                               # a prefix whose weights changed drops its table
        ent = self.cache.get(id(R))
        if ent is None:
            probe, _ = self.prefix(features, xs[0][:1])
            N = R.data.size(0)
            if N * probe[0].numel() * 4 > PREFIX_CACHE_BUDGET:
                return None
            feat = torch.empty((N,) + tuple(probe.shape[1:]), dtype=torch.float32, device=probe.device).contiguous(memory_format=torch.channels_last)
            ent = self.cache[id(R)] = [R, feat, torch.zeros(N, dtype=torch.bool, device=probe.device)]
        _, feat, have = ent
        missing = idx[~have[idx]]
        if missing.numel():
            rows = torch.unique(missing)
            for a in range(0, rows.numel(), 512):
                r = rows[a:a + 512]
                f, _ = self.prefix(features, R.normalised_rows(r))
                feat[r] = f
            have[rows] = True
            self.cache_stats["rows_computed"] += int(rows.numel())
        self.cache_stats["rows_served"] += int(idx.numel())
        f = feat.index_select(0, idx)
        return tuple(f.split([x.size(0) for x in xs], 0))


    @staticmethod
    def _engine_of(features, split, mods):
        """The SuffixEngine of `features[split:]`, cached ON the features module (it dies with the net; a class-level dict keyed by id() kept
        every net of a process alive -- round-4 ADVICE) and rebuilt when the block list changed.  False: the blocks are not the engine's."""
        from isx.suffix import SuffixEngine
        cache = features.__dict__.setdefault('_isx_suffix_engines', {})
        eng = cache.get(split)
        if eng is None or (eng is not False and eng.blocks != mods):
            eng = cache[split] = SuffixEngine(mods) if SuffixEngine.applicable(mods) else False
        return eng

    @classmethod
    def suffix(cls, features, f, split):
        """trainable suffix on the prefix output f (no graph).  Bottleneck blocks on the GPU run as ONE autograd node over libisx
        (isx/suffix.py: folded forward kernels, dgrad / wgrad GEMMs, gradients accumulated straight into .grad); anything else -- other
        block types, CPU, BatchNorm learning, SUFFIX_ENGINE off -- as the plain modules under torch autograd."""
        mods = list(features)[split:]
        if (SUFFIX_ENGINE and mods and f.is_cuda and f.dtype == torch.float32 and torch.is_grad_enabled() and not f.requires_grad
                and any(p.requires_grad for m in mods for p in m.parameters())):
            from isx.suffix import SuffixEngine
            eng = cls._engine_of(features, split, mods)
            if eng and SuffixEngine.applicable(mods):
                return eng(f)
        for m in mods:
            f = m(f)
        return f

    def __call__(self, features, x):
        if not self.usable(features, x):
            return features(x)
        f, split = self.prefix(features, x)
        return self.suffix(features, f, split)

    def inference(self, features, x):
        """Eval-mode forward WITHOUT a graph (the embedding pass of every training epoch, the evaluation between epochs): frozen prefix on its
        cached folded copy, the trainable suffix on a folded copy re-derived when its weights change (once per epoch) -- the whole trunk in
        libisx, as after train._common.prepare_for_inference, without MIOpen's per-shape kernel search on the first pass of a training run."""
        if (torch.is_grad_enabled() or not SPLIT_TRUNK or not (x.is_cuda and x.dtype == torch.float32) or _bn_training(features)
                or not any(isinstance(m, nn.BatchNorm2d) for m in features.modules())):
            return features(x)
        f, split = self.prefix(features, x) if first_trainable(features) > 0 else (x, 0)
        mods = list(features)[split:]
        if not mods:
            return f
        key = (id(features), str(x.device), split,
               tuple((id(t), t._version) for m in mods for t in m.parameters()), tuple((id(t), t._version) for m in mods for t in m.buffers()))
        if getattr(self, "_tail", None) is None or self._tail[0] != key:
            from .nn_utils import fold_batch_norm
            self._tail = (key, fold_batch_norm(nn.Sequential(*mods)).to(x.device).to(memory_format=torch.channels_last))
        return self._tail[1](f)


def _bn_training(features):
    return any(isinstance(m, nn.BatchNorm2d) and m.training for m in features.modules())


def _many(forward_single, xs):
    """forward_single over several same-shaped batches in ONE pass (eval-mode BN: samples are independent), split back."""
    if len(set(tuple(x.shape) for x in xs)) != 1:
        return tuple(forward_single(x) for x in xs)
    n = xs[0].size(0)
    out = forward_single(torch.cat(xs, 0))
    if isinstance(out, tuple):
        return tuple(tuple(o[i * n:(i + 1) * n] for o in out) for i in range(len(xs)))
    if out.requires_grad:
        return _SplitRows.apply(out, len(xs))
    return tuple(out[i * n:(i + 1) * n] for i in range(len(xs)))


class _SplitRows(torch.autograd.Function):
    """(k * n, D) -> k tensors of n consecutive rows; backward = ONE torch.cat of the k gradients (autograd's own slicing costs a zero-filled
    full-size tensor, a copy and an add per slice: 8 launches where the host-bound head of the training step can afford 1)."""

    @staticmethod
    def forward(ctx, x, k):
        n = x.size(0) // k
        ctx.shape = (n, tuple(x.shape[1:]))
        return tuple(x[i * n:(i + 1) * n] for i in range(k))

    @staticmethod
    def backward(ctx, *grads):
        n, rest = ctx.shape
        ref = next(g for g in grads if g is not None)
        return torch.cat([g if g is not None else ref.new_zeros((n,) + rest) for g in grads], 0), None


_SKIP_HEAD_INIT = [False]


class head_weights_follow(object):
    """`with head_weights_follow(bool(P.preload_net)):` around the construction of a siamese net whose state dict is loaded right after: the
    822 MB descriptor head is allocated without its random initialisation (0.6 s of `uniform_` on the host per evaluation run; the reference
    pays it too, model/siamese.py:110-114 + train/siamese_descriptor.py:158-160).  Without a file to load nothing changes."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _SKIP_HEAD_INIT[0]
        _SKIP_HEAD_INIT[0] = self.on
        return self

    def __exit__(self, *exc):
        _SKIP_HEAD_INIT[0] = self.prev


def _descriptor_head(in_features, out_features):
    if _SKIP_HEAD_INIT[0]:
        lin = torch.nn.utils.skip_init(RowDeferredLinear, in_features, out_features)
    else:
        lin = RowDeferredLinear(in_features, out_features)
    return nn.Sequential(NormalizeL2(), Shift(in_features), lin)


def _apply_head(head, rows):
    """feature_reduc1 on (R, F) rows: NormalizeL2 -> Shift -> Linear; fused prologue on the GPU."""
    if _fast(rows, head):
        from isx import ops
        lin = head[2]
        return _linear_rows(ops.l2norm_shift_rows(rows.float(), head[1].param.detach()), lin.weight, lin.bias)
    return head(rows)


def _linear_rows(rows, weight, bias, owner=None):
    """The descriptor head's Linear on GPU rows outside autograd: libisx's split-K GEMM (isx_head_linear_fwd_rows) -- the kernel the training step
    runs, so the per-epoch mining pass, the evaluation and the training forward of the same weights produce the same bits, and a descriptor does
    not depend on the batch it was computed in.  Shapes off the kernel's granules (N % 64, K % 32) run zero-padded on the SAME kernel
    (ops.head_linear_any); anything the kernel cannot take at all (not fp32, K beyond the index range) raises IsxError: a GPU tensor never
    falls to torch's GEMM here (no silent fallback)."""
    from isx import ops
    from isx._lib import IsxError
    if not rows.is_contiguous():
        rows = rows.contiguous()
    if not ops.head_linear_applicable(rows, weight, any_width=True):
        raise IsxError("Linear on GPU rows %s (%s) x weight %s (%s): outside isx_head_linear_fwd_rows (fp32, contiguous weight, K < 2**22, "
                       "rows < 2**24); run the module on the CPU (--device=-1) or cast to float32" %
                       (tuple(rows.shape), rows.dtype, tuple(weight.shape), weight.dtype))
    padded = None
    if (weight.size(0) % 64 or weight.size(1) % 32) and owner is not None:   # the padded copy lives with the module and follows its parameters
        from .nn_utils import _derived
        padded = _derived(owner, '_c_pad64', (weight,) + ((bias,) if bias is not None else ()), lambda: ops.pad_rows_to_64(weight, bias))
    return ops.head_linear_any(rows, weight.detach(), bias.detach() if bias is not None else None, padded)


class RowsLinear(nn.Linear):
    """nn.Linear (same parameters and state-dict keys) whose GPU inference runs on libisx's row-invariant GEMM (_linear_rows): the classifier
    layers of TuneClassif -- class scores used as descriptors (P.embeddings_classify, reference train/classif_finetune.py:87-90) and the
    classification test do not depend on the batch size, nor on how a data-parallel run splits the set."""

    def forward(self, x):
        if x.dim() == 2 and _fast(x, self):
            return _linear_rows(x, self.weight, self.bias, owner=self)
        return F.linear(x, self.weight, self.bias)


def _rows_linears(classifier):
    for m in classifier.modules():
        if type(m) is nn.Linear:
            m.__class__ = RowsLinear


def _box_pools(reduc):
    """TuneClassif's spatial reduction (ResNet: AvgPool2d(7)) on libisx's box-pooling kernel: with the classifier on the row-invariant GEMM the
    whole class-score path of an image is independent of its batch (MIOpen's pooling picks its kernel by batch size)."""
    for m in reduc.modules():
        if type(m) is nn.AvgPool2d:
            m.__class__ = BoxPool


class DescriptorNet(nn.Module):
    def __init__(self, net, feature_dim, feature_size2d, untrained=-1):
        super().__init__()
        self.features, _, classifier = extract_layers(net)
        set_untrained_blocks([self.features], untrained)
        in_features = get_feature_size(self.features, feature_size2d[0] * feature_size2d[1])
        self.feature_size = feature_dim if feature_dim > 0 else get_feature_size(classifier)
        self.feature_reduc1 = _descriptor_head(in_features, self.feature_size)
        self.feature_reduc2 = NormalizeL2()
        self._trunk = _SplitTrunk()

    def forward_single(self, x):
        x = self._trunk(self.features, x) if self.training else self._trunk.inference(self.features, x)
        x = x.reshape(x.size(0), -1)
        return self.feature_reduc2(_apply_head(self.feature_reduc1, x))

    def _tail_and_head(self, f):
        """trainable trunk suffix + descriptor head + final L2 on PREFIX features (B, C, h, w) that were computed elsewhere"""
        f = _SplitTrunk.suffix(self.features, f, self._trunk.split)
        return self.feature_reduc2(_apply_head(self.feature_reduc1, f.reshape(f.size(0), -1)))

    def trunk_precomputable(self):
        """True when precompute_trunk can serve training steps: training mode, GPU, a frozen trunk prefix, BatchNorm not learning."""
        p = next(self.features.parameters(), None)
        return bool(self.training and p is not None and p.is_cuda and _SplitTrunk.enabled(self.features))

    def precompute_trunk(self, *xs, cache=False):
        """Training with a frozen trunk PREFIX and frozen BatchNorm (the reference's configurations: stem + layers 1-3 frozen, layer4
        trained; or everything frozen with untrained = -1): the prefix output of an image does not depend on the batch it rides in and needs
        no autograd graph, so the prefix of a whole mini-batch runs as ONE launch of the folded inference trunk instead of once per
        micro-batch of 8 triplets (24 images: far too few to fill the chip).  Returns the prefix feature tensors of the given image batches
        (same split), or None when the trunk has to run inside the step (nothing frozen, BatchNorm learning, CPU tensors, eval mode)."""
        if not self.training or not xs or not all(x.is_cuda and x.dtype == torch.float32 for x in xs):
            return None
        if not self._trunk.usable(self.features, xs[0]) or len(set(tuple(x.shape[1:]) for x in xs)) != 1:
            return None
        if cache:                              # P.train_prefix_cache: rows of a resident set are looked up, not recomputed (_SplitTrunk.prefix_cached)
            got = self._trunk.prefix_cached(self.features, xs)
            if got is not None:
                return got
        sizes = [x.size(0) for x in xs]
        f, _ = self._trunk.prefix(self.features, torch.cat(xs, 0))
        return tuple(f.split(sizes, 0))

    def suffix_engine(self):
        """The libisx engine of the trainable trunk suffix when a training step may drive it directly (whole-slice forward / backward with
        per-micro-batch gradients, utils/train_general._Stepper), else None."""
        if not (self.trunk_precomputable() and SUFFIX_ENGINE and self._trunk.folded is not None):
            return None
        mods = list(self.features)[self._trunk.split:]
        if not mods or not any(p.requires_grad for m in mods for p in m.parameters()):
            return None
        from isx.suffix import SuffixEngine
        eng = _SplitTrunk._engine_of(self.features, self._trunk.split, mods)
        return eng if eng and SuffixEngine.applicable(mods) else None

    def head_engine(self):
        """The libisx engine of the descriptor head for all local micro-batches of a training step at once (isx/head.py), when the step may
        drive it by hand: GPU training with a precomputable trunk whose trainable part (if any) runs on the suffix engine."""
        if not (HEAD_ENGINE and self.trunk_precomputable() and self._trunk.folded is not None):
            return None
        mods = list(self.features)[self._trunk.split:]
        if any(p.requires_grad for m in mods for p in m.parameters()) and self.suffix_engine() is None:
            return None
        from isx.head import HeadEngine
        if not HeadEngine.applicable(self):
            return None
        eng = self.__dict__.get("_head_engine")
        if eng is None or eng.lin is not self.feature_reduc1[2] or eng.shift is not self.feature_reduc1[1]:
            eng = self.__dict__["_head_engine"] = HeadEngine(self)
        return eng

    def head_features(self, f1, f2=None, f3=None):
        """descriptor head + final L2 on the SUFFIX output of the branches (together, as forward_features sends them through suffix + head)"""
        return _many(self._head_only, [f for f in (f1, f2, f3) if f is not None])

    def _head_only(self, f):
        return self.feature_reduc2(_apply_head(self.feature_reduc1, f.reshape(f.size(0), -1)))

    def head_rows(self, f, n_branches):
        """head_features on the branches ALREADY concatenated along the batch dimension ((n_branches * n) rows): the descriptors, split per branch"""
        out = self._head_only(f)
        if out.requires_grad:
            return _SplitRows.apply(out, n_branches)          # one concatenation in the backward pass instead of three zero-fill + copy + add chains
        n = out.size(0) // n_branches
        return tuple(out[i * n:(i + 1) * n] for i in range(n_branches))

    def forward_features(self, f1, f2=None, f3=None):
        """forward() of training mode on precomputed prefix features: the branches go through suffix + head together, exactly as forward()
        sends them through the whole net together (same rows in the same batch: same bits)."""
        return _many(self._tail_and_head, [f for f in (f1, f2, f3) if f is not None])

    def forward(self, x1, x2=None, x3=None):
        # the reference runs one trunk pass per branch (model/siamese.py:124-130); the branches share the weights, so with
        # BN in eval mode (samples independent) they go through together here
        xs = [x for x in (x1, x2, x3) if x is not None]
        if not self.training:
            return self.forward_single(x1)
        if _bn_training(self.features):
            # BatchNorm is learning (P.train_bn): batch statistics and running-stat updates must be per branch,
            # exactly as the reference's one pass per branch
            return tuple(self.forward_single(x) for x in xs)
        return _many(self.forward_single, xs)


class RegionDescriptorNet(nn.Module):
    def __init__(self, net, k, feature_dim, feature_size2d, untrained=-1):
        super().__init__()
        self.k = k
        self.feature_size2d = tuple(feature_size2d)
        self.features, self.feature_reduc, self.classifier = extract_layers(net)
        in_features = get_feature_size(self.features, feature_size2d[0] * feature_size2d[1])
        # (the reference's feature_dim <= 0 branch names an undefined variable, model/siamese.py:158;
        #  the evident intent -- the classifier's width -- is implemented instead)
        self.feature_size = feature_dim if feature_dim > 0 else get_feature_size(self.classifier)
        has_reduc = len(self.feature_reduc) > 0
        if has_reduc:
            self.feature_reduc = nn.Sequential(BoxPool(self.feature_size2d, stride=1))
        _convolutionalize_classifier(self.classifier, self.feature_size2d, has_reduc)
        set_untrained_blocks([self.features, self.classifier], untrained)
        self.feature_reduc1 = _descriptor_head(in_features, self.feature_size)
        self.feature_reduc2 = NormalizeL2()

    def _single_image(self, x, c):
        """x: (1,C,Hf,Wf) feature map, c: (1,n_cls,H',W') class-score map of ONE image."""
        kh, kw = self.feature_size2d
        n_loc = c.size(2) * c.size(3)
        k = min(n_loc, self.k)
        if _fast(x, self.feature_reduc1):
            from isx import ops
            flat_idx, _ = ops.region_topk(c[0].float(), k)
            rows = ops.region_gather_l2(x[0].float(), kh, kw, flat_idx, c.size(3), self.feature_reduc1[1].param.detach())
            lin = self.feature_reduc1[2]
            acc = _linear_rows(rows, lin.weight, lin.bias).sum(0, keepdim=True)
        else:
            c_maxv = c.max(1)[0].view(-1)
            # canonical tie-break (score desc, index asc): a stable descending sort
            flat_idx = c_maxv.sort(descending=True, stable=True)[1][:k]
            acc = x.new_zeros(1, self.feature_size)
            for i in flat_idx.tolist():
                r, col = i // c.size(3), i % c.size(3)
                region = x[:, :, r:r + kh, col:col + kw].reshape(1, -1)
                acc = acc + self.feature_reduc1(region)
        cls_out = c.new_zeros(1, c.size(1), self.k)
        cls_out[0, :, :k] = c[0].reshape(c.size(1), -1)[:, flat_idx]
        return self.feature_reduc2(acc), cls_out

    def _batched_gpu(self, x, c):
        """Whole batch at once on the GPU: top-k windows of every image in one launch, all B*k window rows
        gathered + normalised + shifted in one launch, ONE Linear over the (B*k, F) rows (the F x D weight is
        streamed once per batch instead of once per window), per-image sum, L2."""
        from isx import ops
        B = x.size(0)
        kh, kw = self.feature_size2d
        k = min(c.size(2) * c.size(3), self.k)
        flat_idx, _ = ops.region_topk(c.float(), k)                                   # (B, k)
        lin = self.feature_reduc1[2]
        if x.dtype == torch.float32 and ops._is_nhwc(x) and x.size(1) % 4 == 0 and x.data_ptr() % 16 == 0:
            # channels-last trunk output: windows gathered as (h,w,C) runs (no transpose of the map), against the Shift vector and the
            # Linear weight with their columns permuted once to that order -- the same dot products, summed in another order
            shift_hwc, w_hwc = self._hwc_head(x.size(1), kh, kw)
            rows = ops.region_gather_l2_nhwc(x, kh, kw, flat_idx, c.size(3), shift_hwc)
            acc = _linear_rows(rows.view(B * k, -1), w_hwc, lin.bias).view(B, k, -1).sum(1)
        else:
            rows = ops.region_gather_l2(x.float(), kh, kw, flat_idx, c.size(3), self.feature_reduc1[1].param.detach())
            acc = _linear_rows(rows.view(B * k, -1), lin.weight, lin.bias).view(B, k, -1).sum(1)
        cls_out = c.new_zeros(B, c.size(1), self.k)
        cls_out[:, :, :k] = c.flatten(2).gather(2, flat_idx[:, None, :].expand(B, c.size(1), k))
        return self.feature_reduc2(acc), cls_out

    def _hwc_head(self, C, kh, kw):
        """(Shift parameter, Linear weight) with the C*kh*kw input features re-ordered (C,h,w) -> (h,w,C); cached, rebuilt when either
        tensor is written in place or replaced (version counters; `.data` surgery needs `self._hwc = None`)."""
        shift, lin = self.feature_reduc1[1].param, self.feature_reduc1[2]
        key = (shift.data_ptr(), shift._version, lin.weight.data_ptr(), lin.weight._version, str(lin.weight.device))
        if getattr(self, "_hwc", None) is None or self._hwc[0] != key:
            D = lin.weight.size(0)
            with torch.no_grad():
                s_hwc = shift.detach().view(C, kh, kw).permute(1, 2, 0).reshape(-1).contiguous()
                w_hwc = lin.weight.detach().view(D, C, kh, kw).permute(0, 2, 3, 1).reshape(D, -1).contiguous()
            self._hwc = (key, s_hwc, w_hwc)
        return self._hwc[1], self._hwc[2]

    def forward_single(self, x):
        # the reference handles one image per call (model/siamese.py:184); here a batch of same-sized
        # images goes through together on the GPU, and image by image on the autograd / CPU path
        x = self.features(x)
        c = self.classifier(self.feature_reduc(x))
        if _fast(x, self.feature_reduc1) and not c.requires_grad:
            return self._batched_gpu(x, c)
        outs = [self._single_image(x[b:b + 1], c[b:b + 1]) for b in range(x.size(0))]
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

    def forward(self, x1, x2=None, x3=None):
        if self.training and x3 is not None:
            return self.forward_single(x1), self.forward_single(x2), self.forward_single(x3)
        if self.training:
            return self.forward_single(x1), self.forward_single(x2)
        return self.forward_single(x1)[0]
