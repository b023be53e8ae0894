#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE'S OWN PYTHON in this container.

Runs only where /root/reference exists (never on the GPU box; nothing at test
time reads /root/reference -- the committed fixtures are plain data).

The reference targets Python 2 + torch 0.1.x; it is imported file-by-file, in
place and unmodified, with harness-side shims only:
  * KD: a torch.Tensor subclass whose dim-reductions keep the reduced dim
    (torch 0.1.x semantics the reference code relies on),
  * a stub `torchvision.models` exposing the three class names nn_utils touches,
  * NormalizeL2.forward / Shift.forward re-routed to the reference's own
    *Fun().forward (legacy autograd Function.__call__ no longer exists).
Closures that cannot be imported (train/*::get_embeddings -- module-level P)
are replayed line by line on KD tensors, citing the lines.

Usage:  python oracle/gen_golden.py      (writes tests/golden/; ISX_GOLDEN_OUT=<dir> writes elsewhere, e.g. to
        compare a regeneration with the committed fixtures: tools/check_golden_regen.py)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
R = "/root/reference"
OUT = os.environ.get("ISX_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class KD(torch.Tensor):
    """torch-0.1.x semantics: a reduction over a dim keeps that dim."""

    def sum(self, *a, **k):
        if a and "keepdim" not in k:
            k["keepdim"] = True
        return super().sum(*a, **k)

    def max(self, *a, **k):
        if a and isinstance(a[0], int) and "keepdim" not in k:
            k["keepdim"] = True
        return super().max(*a, **k)

    def kthvalue(self, kk, dim, **k):
        k.setdefault("keepdim", True)
        return super().kthvalue(kk, dim, **k)

    def sort(self, *a, **k):
        # torch's descending sort does not order ties by ascending index; the canonical
        # tie-break (score desc, index asc) is imposed on the reference's own loop here.
        k.setdefault("stable", True)
        return super().sort(*a, **k)


def install_shims():
    tv, tvm, tvr = (types.ModuleType(n) for n in ("torchvision", "torchvision.models", "torchvision.models.resnet"))

    class ResNet(nn.Module):
        pass

    class Bottleneck(nn.Module):
        pass

    class BasicBlock(nn.Module):
        pass

    tvm.ResNet, tvr.Bottleneck, tvr.BasicBlock, tvm.resnet, tv.models = ResNet, Bottleneck, BasicBlock, tvr, tvm
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    met = load("met", R + "/utils/metrics.py")
    nn_utils = load("nn_utils", R + "/model/nn_utils.py")
    cm = load("custom_modules", R + "/model/custom_modules.py")
    cm.NormalizeL2.forward = lambda self, x: cm.NormalizeL2Fun().forward(x.as_subclass(KD)).as_subclass(torch.Tensor)
    cm.Shift.forward = lambda self, x: cm.ShiftFun().forward(x, self.param)
    siam = load("siamese", R + "/model/siamese.py")
    md = load("ModelDefinition", R + "/model/ModelDefinition.py")
    general = load("general", R + "/utils/general.py")
    gp = load("global_p", R + "/train/global_p.py")
    return met, nn_utils, cm, siam, md, general, gp


class ToyResNetLike(nn.Module):
    """features / feature_reduc(AvgPool) / classifier(1 FC): first branch of extract_layers (nn_utils.py:57-58)."""

    def __init__(self, C, fs, ncls):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, C, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(C, C, 3, stride=2, padding=1), nn.ReLU())
        self.feature_reduc = nn.Sequential(nn.AvgPool2d(fs))
        self.classifier = nn.Sequential(nn.Linear(C, ncls))


class ToyAlexLike(nn.Module):
    """no reduc, two FCs (the first consumes C*fs*fs)."""

    def __init__(self, C, fs, hid, ncls):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, C, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(C, C, 3, stride=2, padding=1), nn.ReLU())
        self.feature_reduc = nn.Sequential()
        self.classifier = nn.Sequential(nn.Linear(C * fs * fs, hid), nn.ReLU(), nn.Linear(hid, ncls))


def npz(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in kw.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    met, nn_utils, cm, siam, md, general, gp = install_shims()
    torch.manual_seed(20260301)                       # toy-backbone weights come from the global RNG
    g = torch.Generator().manual_seed(20260301)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ---- NormalizeL2Fun.forward / ShiftFun.forward (model/custom_modules.py:52-57, 16-18)
    x = rn(7, 37)
    x[2] = 0.0                      # all-zero row -> 0 / sqrt(1e-10)
    x[3] *= 1e4
    x[4] *= 1e-6                    # below sqrt(eps): eps dominates
    y = cm.NormalizeL2Fun().forward(x.clone().as_subclass(KD)).as_subclass(torch.Tensor)
    p = rn(37)
    ys = cm.ShiftFun().forward(x, p)
    xw = rn(5, 2048)
    yw = cm.NormalizeL2Fun().forward(xw.clone().as_subclass(KD)).as_subclass(torch.Tensor)
    npz("l2norm_shift.npz", x=x, y=y, param=p, y_shift=ys, x_wide=xw, y_wide=yw)

    # ---- global pooling + L2: TuneClassif.forward with stripped classifier
    #      (model/siamese.py:49-54, train/classif_finetune.py:87-100)
    with torch.no_grad():
        bb = ToyResNetLike(16, 4, 11)
        net = siam.TuneClassif(bb, 11).eval()
        img = rn(3, 3, 16, 16)
        fmap = net.features(img)                       # (3,16,4,4)
        classifier = net.classifier
        net.classifier = nn.Sequential()
        out = net(img)
        out = cm.NormalizeL2Fun().forward(out.as_subclass(KD)).as_subclass(torch.Tensor)
        net.classifier = classifier
        logits = net(img)
        out_cls = cm.NormalizeL2Fun().forward(logits.as_subclass(KD)).as_subclass(torch.Tensor)
        fm7 = rn(2, 24, 7, 7)
        o7 = nn.AvgPool2d(7)(fm7).view(2, -1)
        o7 = cm.NormalizeL2Fun().forward(o7.as_subclass(KD)).as_subclass(torch.Tensor)
    npz("gap_l2.npz", fmap=fmap, desc=out, logits=logits, desc_classify=out_cls,
        fc_w=classifier[0].weight, fc_b=classifier[0].bias, fmap7=fm7, desc7=o7)

    # ---- DescriptorNet head (model/siamese.py:100-122)
    with torch.no_grad():
        bb = ToyAlexLike(8, 3, 12, 5)
        dn = siam.DescriptorNet(bb, 10, (3, 3)).eval()
        dn.feature_reduc1[1].param.data = rn(8 * 3 * 3) * 0.05
        img = rn(4, 3, 12, 12)
        fm = dn.features(img)                          # (4,8,3,3)
        d = dn(img)
    npz("descriptor_head.npz", fmap=fm, shift=dn.feature_reduc1[1].param, w=dn.feature_reduc1[2].weight,
        b=dn.feature_reduc1[2].bias, desc=d, feature_size=dn.feature_size)

    # ---- TuneClassifSub score maps (model/siamese.py:64-89), both backbone kinds
    with torch.no_grad():
        bb = ToyResNetLike(16, 3, 9)
        sub = siam.TuneClassifSub(bb, 9, (3, 3)).eval()
        img = rn(1, 3, 28, 20)
        fm_r = sub.features(img)                       # (1,16,7,5)
        pooled_r = sub.feature_reduc(fm_r)             # (1,16,5,3)
        map_r = sub(img)[0]                            # (1,9,5,3)
        conv_r = sub.classifier[0]
        bb2 = ToyAlexLike(8, 3, 12, 6)
        sub2 = siam.TuneClassifSub(bb2, 6, (3, 3)).eval()
        img2 = rn(1, 3, 24, 28)
        fm_a = sub2.features(img2)                     # (1,8,6,7)
        map_a = sub2(img2)[0]                          # (1,6,4,5)
        fc0 = bb2.classifier[0]
        conv0 = sub2.classifier[0]
    npz("classif_sub.npz", fmap_r=fm_r, pooled_r=pooled_r, map_r=map_r, w_r=conv_r.weight, b_r=conv_r.bias,
        fmap_a=fm_a, map_a=map_a, w0_a=conv0.weight, b0_a=conv0.bias, w1_a=sub2.classifier[2].weight,
        b1_a=sub2.classifier[2].bias)

    # ---- classif_regions get_embeddings (train/classif_regions.py:118-128), replayed on KD tensors
    def best_loc_ref(out):
        out = out.as_subclass(KD)
        max_pred, _ = out.max(1)
        max_pred1, max_i1 = max_pred.max(2)
        _, max_i2 = max_pred1.max(3)
        i2 = max_i2.view(-1)[0]
        i1 = max_i1.view(-1)[i2]
        o = out[:, :, i1, i2]
        o = cm.NormalizeL2Fun().forward(o.as_subclass(KD)).as_subclass(torch.Tensor)
        return o[0], int(i1), int(i2)
    maps, descs, locs = [], [], []
    for t in range(4):
        m = rn(1, 9, 5, 3) if t < 3 else rn(1, 9, 1, 1)
        if t == 2:                                     # exact tie of the class-max at two locations
            m[0, :, 3, 1] = m[0, :, 1, 1]
            m[0, 4, 1, 1] = m[0, 4, 3, 1] = 9.0
        d, i1, i2 = best_loc_ref(m)
        maps.append(m[0].numpy()); descs.append(d.numpy()); locs.append((i1, i2))
    d_r, i1_r, i2_r = best_loc_ref(map_r)
    np.savez_compressed(os.path.join(OUT, "best_location.npz"), map0=maps[0], map1=maps[1], map2=maps[2], map3=maps[3],
                        desc0=descs[0], desc1=descs[1], desc2=descs[2], desc3=descs[3], locs=np.array(locs),
                        map_r=map_r[0].numpy(), desc_r=d_r.numpy(), loc_r=np.array([i1_r, i2_r]))

    # ---- RegionDescriptorNet (model/siamese.py:148-223)
    reg = {}
    with torch.no_grad():
        for tag, k in (("k3", 3), ("k40", 40)):       # k < #locations and k > #locations
            bb = ToyResNetLike(16, 3, 9)
            rd = siam.RegionDescriptorNet(bb, k, 12, (3, 3)).eval()
            rd.feature_reduc1[1].param.data = rn(16 * 9) * 0.05
            img = rn(1, 3, 28, 20)
            fm = rd.features(img)
            c = rd.classifier(rd.feature_reduc(fm))
            c_maxv = c.as_subclass(KD).max(1)[0].view(-1)
            kk = min(c_maxv.size(0), k)
            _, flat_idx = c_maxv.topk(kk)
            _, stable = c_maxv.as_subclass(torch.Tensor).sort(descending=True, stable=True)
            assert torch.equal(flat_idx.as_subclass(torch.Tensor), stable[:kk]), "topk tie order differs from canonical"
            d = rd(img)
            reg.update({"fmap_" + tag: fm, "cls_" + tag: c, "idx_" + tag: flat_idx.as_subclass(torch.Tensor),
                        "shift_" + tag: rd.feature_reduc1[1].param, "w_" + tag: rd.feature_reduc1[2].weight,
                        "b_" + tag: rd.feature_reduc1[2].bias, "desc_" + tag: d,
                        "cw_" + tag: rd.classifier[0].weight, "cb_" + tag: rd.classifier[0].bias})
    npz("region_desc.npz", **reg)

    # ---- metrics (utils/metrics.py:8-55) -- the reference functions, unmodified
    def sets(qlab, glab):
        return [(None, int(l), None) for l in qlab], [(None, int(l), None) for l in glab]
    mg = torch.Generator().manual_seed(7)
    M, N, L = 12, 40, 6
    sim = torch.rand(M, N, generator=mg) * 2 - 1
    glab = torch.arange(N) % L
    qlab = torch.arange(M) % (L + 1)                   # label L never appears in the gallery -> skipped queries
    ts, rs = sets(qlab, glab)
    res = {"sim": sim, "qlab": qlab.int(), "glab": glab.int()}
    for kth in (1, 2, 3):
        p1 = met.precision1(sim.as_subclass(KD), ts, rs, kth)
        aps = [met.avg_precision(sim, i, ts, rs, kth) for i in range(M)]
        res["p1_kth%d" % kth] = np.array([p1[0], p1[1], p1[2]], dtype=np.float64)
        res["p1_maxsim_kth%d" % kth] = p1[3].as_subclass(torch.Tensor).reshape(-1)
        res["p1_maxlabel_kth%d" % kth] = np.array(p1[4])
        res["ap_kth%d" % kth] = np.array([np.nan if a is None else a for a in aps], dtype=np.float64)
        res["map_kth%d" % kth] = np.float64(met.mean_avg_precision(sim, ts, rs, kth))
    # duplicate scores: small N where torch's sort is checked to be index-ascending on ties
    simt = torch.round(torch.rand(5, 24, generator=mg) * 6) / 6
    glt = torch.arange(24) % 4
    qlt = torch.arange(5) % 4
    tst, rst = sets(qlt, glt)
    simt = simt.as_subclass(KD)                        # stable sort inside the reference's avg_precision
    assert any(torch.unique(simt[i]).numel() < 24 for i in range(5))
    res.update({"tie_sim": simt.as_subclass(torch.Tensor), "tie_qlab": qlt.int(), "tie_glab": glt.int(),
                "tie_ap": np.array([met.avg_precision(simt, i, tst, rst, 1) for i in range(5)], dtype=np.float64),
                "tie_map": np.float64(met.mean_avg_precision(simt, tst, rst, 1))})
    npz("metrics.npz", **res)

    # ---- synthetic retrieval set of SURVEY 8d (sigma = 4), reference mAP / P@1 on torch.mm scores
    syn = {}
    for N_, M_, D_ in ((100, 20, 32), (1000, 50, 32)):
        sg = torch.Generator().manual_seed(0)
        L_ = N_ // 10
        cent = torch.randn(L_, D_, generator=sg)
        gl = torch.arange(N_) % L_
        ql = torch.arange(M_) % L_
        G = cent[gl] + 4.0 * torch.randn(N_, D_, generator=sg)
        Q = cent[ql] + 4.0 * torch.randn(M_, D_, generator=sg)
        G = cm.NormalizeL2Fun().forward(G.as_subclass(KD)).as_subclass(torch.Tensor)
        Q = cm.NormalizeL2Fun().forward(Q.as_subclass(KD)).as_subclass(torch.Tensor)
        s = torch.mm(Q, G.t())
        # fp32 scores do collide at N = 1000: the KD subclass makes the reference's sort stable
        ts_, rs_ = sets(ql, gl)
        p1 = met.precision1(s.as_subclass(KD), ts_, rs_)
        t = "_n%d" % N_
        syn.update({"Q" + t: Q, "G" + t: G, "sim" + t: s, "qlab" + t: ql.int(), "glab" + t: gl.int(),
                    "map" + t: np.float64(met.mean_avg_precision(s.as_subclass(KD), ts_, rs_)),
                    "p1" + t: np.array([p1[0], p1[1], p1[2]], dtype=np.float64)})
    npz("synthetic_retrieval.npz", **syn)

    # ---- training step: TripletLossFun.forward (model/custom_modules.py:153-171, the reference's own code) and the
    #      negative mining of train/siamese_descriptor.py:94-128 replayed line by line
    tg_ = torch.Generator().manual_seed(11)
    nrm = lambda t: cm.NormalizeL2Fun().forward(t.as_subclass(KD)).as_subclass(torch.Tensor)
    A, Pp, Nn = nrm(torch.randn(9, 24, generator=tg_)), nrm(torch.randn(9, 24, generator=tg_)), nrm(torch.randn(9, 24, generator=tg_))
    Pp[:4] = nrm(A[:4] + 0.05 * torch.randn(4, 24, generator=tg_))            # easy positives -> some rows clamp to 0
    trip = {"a": A, "p": Pp, "n": Nn}
    for normalized in (True, False):
        for avg in (True, False):
            f = cm.TripletLossFun(0.1, avg, normalized)
            f.save_for_backward = lambda *a_: None
            loss = f.forward(A.clone().as_subclass(KD), Pp.clone().as_subclass(KD), Nn.clone().as_subclass(KD))
            trip["loss_n%d_a%d" % (normalized, avg)] = loss.as_subclass(torch.Tensor)
    mlf = cm.MetricLossFun(True)
    mlf.save_for_backward = lambda *a_: None
    yv = torch.tensor([1., -1., 1., -1., 1., -1., 1., -1., 1.])
    trip["metric_loss"] = mlf.forward(A.clone().as_subclass(KD), Pp.clone().as_subclass(KD), yv).as_subclass(torch.Tensor)
    trip["metric_y"] = yv
    Ns_ = 60
    E_ = nrm(torch.randn(Ns_, 12, generator=tg_))
    E_[7] = E_[3]                                                                  # duplicate item: tied similarities
    labs_ = torch.arange(Ns_) % 9
    S_ = torch.mm(E_, E_.t())
    couples_ = [(int(i), int(j)) for i in range(Ns_) for j in range(i, Ns_) if labs_[i] == labs_[j]][:80]
    mined = {}
    for semi in (1, 0):
        out = []
        for (c1, c2) in couples_:
            ind_exl = labs_ == labs_[c1]
            sim_pos = S_[c1, c2]
            if semi:
                ind_exl = ind_exl | S_[c1].ge(sim_pos)
            if int(ind_exl.sum()) >= S_.size(0):
                out.append(-1)
            else:
                sims = S_[c1].clone()
                sims[ind_exl] = -2
                _, kk = sims.max(0)
                out.append(int(kk))
        mined["neg_semi%d" % semi] = np.array(out, np.int64)
    trip.update({"mine_sim": S_, "mine_labels": labs_.int(), "mine_i1": np.array([c[0] for c in couples_], np.int64),
                 "mine_i2": np.array([c[1] for c in couples_], np.int64), **mined})
    npz("training.npz", **trip)

    # ---- host-side helpers: Maxnet structure, copyParameters, convolutionalize, parse/check, tables
    host = {}
    mx = md.Maxnet(17)
    host["maxnet_state"] = {k: list(v.shape) for k, v in mx.state_dict().items()}
    host["maxnet_modules"] = [type(m).__name__ for m in list(mx.features) + list(mx.classifier)]
    a, b = md.Maxnet(5), md.Maxnet(7)
    md.copyParameters(a, b)
    host["copy_same"] = [bool(torch.equal(a.features[i].weight, b.features[i].weight)) for i in (0, 3, 6, 8, 10)] + \
                        [bool(torch.equal(a.classifier[i].weight, b.classifier[i].weight)) for i in (1, 4, 6)]
    host["parse_dataset_id"] = {s: general.parse_dataset_id(s) for s in ("a/b/CLICIDE", "a/b/CLICIDE/", "oxford5k_video_384")}
    host["check_bool"] = {s: general.check_bool(s, "x", None) for s in ("true", "Yes", "y", "1", "0", "no", "False")}
    host["image_sizes"] = {k: list(v) for k, v in gp.image_sizes.items()}
    host["num_classes"] = gp.num_classes
    host["feature_sizes"] = [[list(k[:1]) + [list(k[1])], list(v)] for k, v in gp.feature_sizes.items()]
    host["flat_feature_sizes"] = [[list(k[:1]) + [list(k[1])], v] for k, v in gp.flat_feature_sizes.items()]
    host["mean_std_files"] = gp.mean_std_files
    host["match_label"] = {"fou": gp.match_label_fou_clean2("d/ab_cd_ef.jpg"), "video": gp.match_label_video("d/x12-3.jpg"),
                           "oxford": gp.match_label_oxford("d/all_souls_000013.jpg")}
    fc = nn.Linear(8 * 2 * 3, 4)
    cv = nn_utils.convolutionalize(fc, (2, 3))
    xin = rn(2, 8, 2, 3)
    with torch.no_grad():
        host["convolutionalize_equal"] = float((cv(xin).view(2, -1) - fc(xin.view(2, -1))).abs().max())
    host["get_feature_size"] = [nn_utils.get_feature_size(nn.Sequential(nn.Conv2d(3, 5, 1), nn.ReLU()), 4),
                                nn_utils.get_feature_size(nn.Sequential(nn.Linear(3, 6))), nn_utils.get_feature_size(nn.Sequential(), 1, -1)]
    # fold_batches (utils/train_general.py:27-38): load with `general` already importable
    sys.path.insert(0, R + "/utils")
    mod = types.ModuleType("model"); mod.nn_utils = nn_utils
    sys.modules["model"] = mod; sys.modules["model.nn_utils"] = nn_utils
    tg = load("train_general", R + "/utils/train_general.py")
    calls = {}
    for n, bs, cut in ((10, 3, False), (10, 3, True), (9, 3, False), (9, 3, True), (5, 0, False), (4, 8, False), (4, 8, True), (0, 2, False)):
        def f(last, idx, is_final, batch):
            return last + [[idx, bool(is_final), len(batch)]]
        calls["%d_%d_%d" % (n, bs, cut)] = tg.fold_batches(f, [], list(range(n)), bs, cut_end=cut)
    host["fold_batches"] = calls
    # utils/train_siamese.py (a14 / a15) is run by siamese_eval() below
    with open(os.path.join(OUT, "host_helpers.json"), "w") as fh:
        json.dump(host, fh, indent=1, sort_keys=True)
    print("golden fixtures written to", os.path.abspath(OUT))
    for f_ in sorted(os.listdir(OUT)):
        print("  %-28s %8d B" % (f_, os.path.getsize(os.path.join(OUT, f_))))


def trunk_block():
    """One ResNet bottleneck block (the unit of the `features` trunk the reference takes from torchvision,
    model/nn_utils.py:56-71) evaluated by torch itself in eval mode: conv1x1-BN-ReLU, conv3x3(stride 2)-BN-ReLU,
    conv1x1-BN, + conv1x1(stride 2)-BN shortcut, ReLU; and a stride-1 identity-shortcut block on its output.
    torch.nn.functional.conv2d / batch_norm are the third-party algorithms (PyTorch, this image: 2.10) the
    oracle's isxo_conv1x1_nhwc / isxo_conv3x3_nhwc + BN folding restate."""
    import torch.nn as nn
    import torch.nn.functional as F
    torch.manual_seed(20260302)

    def bn(c):
        m = nn.BatchNorm2d(c).eval()
        m.weight.data = torch.rand(c) + 0.5
        m.bias.data = torch.randn(c) * 0.1
        m.running_mean.data = torch.randn(c) * 0.1
        m.running_var.data = torch.rand(c) + 0.5
        return m

    out = {}
    x = torch.relu(torch.randn(2, 64, 9, 8))
    out["x"] = x
    y = x
    for blk, (cin, mid, cout, stride) in enumerate(((64, 32, 128, 2), (128, 32, 128, 1))):
        convs = [nn.Conv2d(cin, mid, 1, bias=False), nn.Conv2d(mid, mid, 3, stride, 1, bias=False), nn.Conv2d(mid, cout, 1, bias=False)]
        bns = [bn(mid), bn(mid), bn(cout)]
        with torch.no_grad():
            idt = y
            if stride != 1 or cin != cout:
                dconv, dbn = nn.Conv2d(cin, cout, 1, stride, bias=False), bn(cout)
                idt = dbn(dconv(y))
                out["b%d_dw" % blk] = dconv.weight
                for k_, v_ in (("g", dbn.weight), ("b", dbn.bias), ("m", dbn.running_mean), ("v", dbn.running_var)):
                    out["b%d_d%s" % (blk, k_)] = v_
            t = y
            for i, (c, b_) in enumerate(zip(convs, bns)):
                t = b_(c(t))
                if i < 2:
                    t = torch.relu(t)
                out["b%d_w%d" % (blk, i)] = c.weight
                for k_, v_ in (("g", b_.weight), ("b", b_.bias), ("m", b_.running_mean), ("v", b_.running_var)):
                    out["b%d_%s%d" % (blk, k_, i)] = v_
            y = torch.relu(t + idt)
        out["y%d" % blk] = y
    out["eps"] = np.float32(1e-5)
    npz("trunk_block.npz", **out)


class U8Mask(torch.Tensor):
    """A uint8 label indicator with the torch-0.1.x mask semantics test/instance_avg.py:26 relies on: `1 - mask` is the
    complement AND usable as an index mask (modern torch refuses uint8 masks)."""

    def __rsub__(self, other):
        assert other == 1
        return self.as_subclass(torch.Tensor) == 0


def dba():
    """Database-side feature augmentation: the reference's own test/instance_avg.py:7-33 loop, imported in place and run
    unmodified; harness-side shims only: a `utils` module exposing the reference's get_lab_indicators
    (utils/dataset.py:65-76, itself unmodified) with its ByteTensors wrapped in U8Mask.  Fixture: descriptors, labels
    (a singleton label, a pair, larger groups), outputs for k = -1, 0, 1, 2, 5."""
    if "general" not in sys.modules:
        load("general", R + "/utils/general.py")
    dataset = load("dataset", R + "/utils/dataset.py")
    u = types.ModuleType("utils")
    u.get_lab_indicators = lambda ds, device: {k: v.as_subclass(U8Mask) for k, v in dataset.get_lab_indicators(ds, device).items()}
    saved = sys.modules.get("utils")
    sys.modules["utils"] = u
    try:
        ia = load("instance_avg", R + "/test/instance_avg.py")
    finally:
        if saved is not None:
            sys.modules["utils"] = saved
        else:
            del sys.modules["utils"]
    g = torch.Generator().manual_seed(20260303)
    labs = [0, 1, 2, 0, 1, 0, 3, 1, 0, 2, 0, 1, 4, 4, 0, 5, 1, 2, 0, 4, 2, 1, 0, 6, 6, 6, 0, 2, 1, 0, 4, 2, 6, 0, 1, 6, 2, 0, 4, 1]
    cent = torch.randn(7, 48, generator=g)
    E = cent[torch.tensor(labs)] + 0.7 * torch.randn(len(labs), 48, generator=g)
    E = E / E.norm(dim=1, keepdim=True)                                   # label 3 and label 5 are singletons
    ds = [(None, "L%d" % l, None) for l in labs]
    out = {"emb": E, "labels": np.asarray(labs, np.int32)}
    for k in (-1, 0, 1, 2, 5):
        new, _ = ia.instance_avg(-1, E.clone(), ds, sorted(set("L%d" % l for l in labs)), k)
        out["k%s" % ("all" if k < 0 else k)] = new
    npz("dba.npz", **out)


def siamese_eval():
    """a14 / a15: the reference's own utils/train_siamese.py (embeddings_device_dim :30-43, get_similarities :48-55,
    test_descriptor_net :61-82), imported in place and run unmodified.  Its imports (`general`, `dataset`, `metrics`,
    `model.nn_utils`) are the reference's files loaded as top-level modules, as the file itself expects (it puts utils/
    on the path).  Harness-side shim only: descriptors handed over as KD tensors (torch-0.1.x keepdim reductions, stable
    descending sort -> canonical tie order).  get_embeddings is the identity on the descriptor stored in each dataset
    tuple, so the fixture pins everything AFTER the embedding pass: P@1 (kth 1 and 2), mAP, sum_pos, sum_neg, sum_max,
    lab_dict (including the reference's `setdefault(lab, get(lab, 0) + 1)` counting quirk)."""
    for name, path in (("general", "/utils/general.py"), ("dataset", "/utils/dataset.py"), ("metrics", "/utils/metrics.py")):
        if name not in sys.modules:
            load(name, R + path)
    if "torchvision" not in sys.modules:
        install_shims()
    nn_utils = sys.modules["nn_utils"]
    mod = types.ModuleType("model"); mod.nn_utils = nn_utils
    sys.modules["model"] = mod; sys.modules["model.nn_utils"] = nn_utils
    ts_mod = load("ref_train_siamese", R + "/utils/train_siamese.py")
    cm = sys.modules["custom_modules"]

    class P:
        cuda_device, feature_dim, embeddings_cuda_size, train_bn = 0, 0, 2 ** 30, False

    class Net(nn.Module):
        def __init__(self, fs):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
            self.feature_size = fs

    class Bare(object):
        pass

    # ---- embeddings_device_dim: every branch of :30-43
    dim_cases = []
    for cuda_device, feature_dim, fs, n, sim_matrix in (
            (0, 0, 32, 10, False), (0, 64, 32, 10, False), (3, -1, 2048, 1000, False), (-1, 32, 32, 10, False),
            (0, 0, 32, 2 ** 23, False), (0, 0, 32, 2 ** 23 + 1, False),            # slab exactly at / just over the budget
            (0, 0, 32, 16384, True), (0, 0, 32, 16385, True),                       # n x n exactly at / just over
            (0, 128, None, 10, False), (0, 0, None, 10, False)):                    # a net without feature_size
        P.cuda_device, P.feature_dim = cuda_device, feature_dim
        net = Net(fs) if fs is not None else Bare()
        dev, out = ts_mod.embeddings_device_dim(P, net, n, sim_matrix)
        dim_cases.append({"cuda_device": cuda_device, "feature_dim": feature_dim, "feature_size": fs, "n": n,
                          "sim_matrix": sim_matrix, "budget": P.embeddings_cuda_size, "device": dev, "out_size": out})

    # ---- test_descriptor_net / get_similarities on the SURVEY 8d synthetic set (sigma = 4), D = 32
    out = {}
    meta = {"embeddings_device_dim": dim_cases, "runs": {}}
    P.cuda_device, P.feature_dim = -1, 32
    for tag, (N_, M_) in (("n100", (100, 20)), ("n1000", (1000, 50))):
        sg = torch.Generator().manual_seed(0)
        L_, D_ = N_ // 10, 32
        cent = torch.randn(L_, D_, generator=sg)
        gl = torch.arange(N_) % L_
        ql = torch.arange(M_) % L_
        G = cent[gl] + 4.0 * torch.randn(N_, D_, generator=sg)
        Q = cent[ql] + 4.0 * torch.randn(M_, D_, generator=sg)
        G = cm.NormalizeL2Fun().forward(G.as_subclass(KD)).as_subclass(torch.Tensor)
        Q = cm.NormalizeL2Fun().forward(Q.as_subclass(KD)).as_subclass(torch.Tensor)
        if tag == "n100":
            ql = ql.clone(); ql[7] = 99                       # a query whose label is absent from the gallery (AP skipped, :31-32 of metrics)
            G[13] = G[3]                                      # duplicated gallery rows: tied scores in every row
        test_set = [(Q[i], "L%d" % int(l), None) for i, l in enumerate(ql)]
        ref_set = [(G[i], "L%d" % int(l), None) for i, l in enumerate(gl)]
        seen = []

        def get_embeddings(net, dataset, device, out_size):
            seen.append((len(dataset), device, out_size))
            return torch.stack([x for x, _, _ in dataset]).as_subclass(KD)

        net = Net(32)
        out.update({"Q_" + tag: Q, "G_" + tag: G, "qlab_" + tag: np.array([int(l) for l in ql], np.int32),
                    "glab_" + tag: gl.int()})
        for kth in (1, 2):
            # kth = 2 is the reference's train-against-train form (:113): queries drawn from the gallery itself
            if kth == 2:
                qs = [ref_set[i] for i in range(0, N_, 7)]
                out["self_idx_" + tag] = np.arange(0, N_, 7, dtype=np.int64)
            else:
                qs = test_set
            del seen[:]
            prec1, correct, total, sum_pos, sum_neg, sum_max, mAP, lab_dict = ts_mod.test_descriptor_net(P, get_embeddings, net, qs, ref_set, kth)
            key = "%s_kth%d" % (tag, kth)
            out["sums_" + key] = np.array([float(sum_pos), float(sum_neg), float(sum_max)], np.float64)
            out["sums_f32_" + key] = np.array([float(sum_pos), float(sum_neg), float(sum_max)], np.float32)
            out["p1_" + key] = np.array([prec1, correct, total], np.float64)
            out["map_" + key] = np.float64(mAP)
            meta["runs"][key] = {"lab_dict": {k: dict(v) for k, v in lab_dict.items()}, "get_embeddings_calls": list(map(list, seen))}
        # get_similarities (:48-55): eval mode for the pass, train mode with frozen BatchNorm afterwards
        net.train()
        sims, dev = ts_mod.get_similarities(P, get_embeddings, net, ref_set)
        sims = sims.as_subclass(torch.Tensor)
        out["selfsim_" + tag] = sims if N_ <= 100 else sims[::125]     # n1000: 8 of the 1000 rows keep the fixture small
        meta["runs"][tag + "_get_similarities"] = {"device": dev, "net_training": bool(net.training),
                                                   "bn_training": bool(net.features[1].training)}
    npz("siamese_eval.npz", **out)
    with open(os.path.join(OUT, "siamese_eval.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trunk":
        os.makedirs(OUT, exist_ok=True)
        trunk_block()
    elif len(sys.argv) > 1 and sys.argv[1] == "dba":
        os.makedirs(OUT, exist_ok=True)
        dba()
    elif len(sys.argv) > 1 and sys.argv[1] == "siamese_eval":
        os.makedirs(OUT, exist_ok=True)
        siamese_eval()
    else:
        main()
        trunk_block()
        dba()
        siamese_eval()
