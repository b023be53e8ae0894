"""The Python drop-in surface (model/, utils/, train/, test/ under instance-search_amd/) on
the CPU: same names and behaviour as the reference, checked against golden values produced by
the reference's own code (tests/golden/, oracle/gen_golden.py)."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HOST = json.load(open(os.path.join(GOLDEN, "host_helpers.json")))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ------------------------------------------------------------------ host helpers
def test_maxnet_structure_and_copy():
    from model.ModelDefinition import Maxnet, copyParameters, maxnet
    net = Maxnet(17)
    assert isinstance(net, maxnet)
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == HOST["maxnet_state"]
    assert [type(m).__name__ for m in list(net.features) + list(net.classifier)] == HOST["maxnet_modules"]
    a, b = Maxnet(5), Maxnet(7)
    copyParameters(a, b)
    same = [bool(torch.equal(a.features[i].weight, b.features[i].weight)) for i in (0, 3, 6, 8, 10)] + \
           [bool(torch.equal(a.classifier[i].weight, b.classifier[i].weight)) for i in (1, 4, 6)]
    assert same == HOST["copy_same"]
    x = torch.randn(2, 3, 224, 224)
    assert net.eval()(x).shape == (2, 17)


def test_general_and_tables():
    from utils import check_bool, parse_dataset_id, read_mean_std
    from train import global_p as gp
    for s, want in HOST["parse_dataset_id"].items():
        assert parse_dataset_id(s) == want
    for s, want in HOST["check_bool"].items():
        assert check_bool(s, "x", None) == want
    assert {k: list(v) for k, v in gp.image_sizes.items()} == HOST["image_sizes"]
    assert gp.num_classes == HOST["num_classes"]
    assert gp.mean_std_files == HOST["mean_std_files"]
    for (model, size), v in HOST["feature_sizes"]:
        assert list(gp.feature_sizes[(model[0], tuple(size))]) == v if isinstance(model, list) else True
    for key, v in HOST["feature_sizes"]:
        assert list(gp.feature_sizes[(key[0], tuple(key[1]))]) == v
    for key, v in HOST["flat_feature_sizes"]:
        assert gp.flat_feature_sizes[(key[0], tuple(key[1]))] == v
    assert gp.match_label_fou_clean2("d/ab_cd_ef.jpg") == HOST["match_label"]["fou"]
    assert gp.match_label_video("d/x12-3.jpg") == HOST["match_label"]["video"]
    assert gp.match_label_oxford("d/all_souls_000013.jpg") == HOST["match_label"]["oxford"]
    assert set(gp.match_label_functions) == set(HOST["image_sizes"])


def test_read_mean_std(tmp_path):
    from utils import read_mean_std
    p = tmp_path / "ms.txt"
    p.write_text("0.3643 0.3043 0.2774\n0.2122 0.2003 0.1976\n")
    m, s = read_mean_std(str(p))
    assert list(m) == [0.3643, 0.3043, 0.2774] and list(s) == [0.2122, 0.2003, 0.1976]


def test_fold_batches_call_sequence():
    from utils import fold_batches
    for key, want in HOST["fold_batches"].items():
        n, bs, cut = (int(v) for v in key.split("_"))
        got = fold_batches(lambda acc, i, fin, b: acc + [[i, bool(fin), len(b)]], [], list(range(n)), bs, cut_end=bool(cut))
        assert got == want, key


def test_nn_utils():
    from model.nn_utils import convolutionalize, extract_layers, get_feature_size, set_net_train, set_untrained_blocks
    from isx import backbones
    fc = nn.Linear(8 * 2 * 3, 4)
    cv = convolutionalize(fc, (2, 3))
    x = torch.randn(2, 8, 2, 3)
    with torch.no_grad():
        assert float((cv(x).view(2, -1) - fc(x.view(2, -1))).abs().max()) <= max(HOST["convolutionalize_equal"], 1e-6)
    with pytest.raises(ValueError):
        convolutionalize(fc, (5, 5))
    assert [get_feature_size(nn.Sequential(nn.Conv2d(3, 5, 1), nn.ReLU()), 4), get_feature_size(nn.Sequential(nn.Linear(3, 6))),
            get_feature_size(nn.Sequential(), 1, -1)] == HOST["get_feature_size"]
    r = backbones.resnet50()
    f, red, c = extract_layers(r)
    assert len(f) == 4 + 3 + 4 + 6 + 3 and isinstance(red[0], nn.AvgPool2d) and isinstance(c[0], nn.Linear)
    assert get_feature_size(f, 49) == 2048 * 49
    a = backbones.alexnet()
    f, red, c = extract_layers(a)
    assert len(f) == 13 and len(red) == 0 and len(c) == 7
    set_untrained_blocks([f, c], 2)
    flags = [p.requires_grad for m in f for p in m.parameters()]
    assert flags[:4] == [False] * 4 and all(flags[4:])
    set_untrained_blocks([f], -1)
    assert not any(p.requires_grad for p in f.parameters())

    class Wrap(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    w = Wrap()
    set_net_train(w, True)
    assert w.training and not w.features[1].training
    set_net_train(w, True, bn_train=True)
    assert w.features[1].training
    set_net_train(w, False)
    assert not w.training


def test_fold_batch_norm_is_the_same_function():
    from isx import backbones
    from model.nn_utils import extract_layers, fold_batch_norm
    torch.manual_seed(0)
    for ctor in (backbones.resnet18, backbones.resnet50):
        net = ctor(pretrained=True).eval()
        for m in net.modules():                      # non-trivial running statistics
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
        feats, _, _ = extract_layers(net)
        folded = fold_batch_norm(feats)
        assert not any(isinstance(m, nn.BatchNorm2d) for m in folded.modules())
        assert any(isinstance(m, nn.BatchNorm2d) for m in feats.modules())          # the original is untouched
        x = torch.randn(2, 3, 64, 64)
        with torch.no_grad():
            a, b = feats(x), folded(x)
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())


def test_backbone_state_dict_keys_are_torchvision_compatible():
    from isx import backbones
    keys = list(backbones.resnet50().state_dict())
    assert keys[0] == "conv1.weight" and "layer1.0.downsample.0.weight" in keys and "layer4.2.bn3.running_var" in keys
    assert keys[-2:] == ["fc.weight", "fc.bias"]
    assert len([k for k in backbones.resnet152().state_dict() if k.endswith("conv1.weight")]) == 1 + 3 + 8 + 36 + 3
    ak = list(backbones.alexnet().state_dict())
    assert ak == ["features.%d.%s" % (i, p) for i in (0, 3, 6, 8, 10) for p in ("weight", "bias")] + \
                 ["classifier.%d.%s" % (i, p) for i in (1, 4, 6) for p in ("weight", "bias")]
    a, b = backbones.resnet18(pretrained=True), backbones.resnet18(pretrained=True)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))    # seeded "pretrained"


# ------------------------------------------------------------------ modules vs golden
def _identity_features(C):
    conv = nn.Conv2d(C, C, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.eye(C).view(C, C, 1, 1))
        conv.bias.zero_()
    return nn.Sequential(conv)


class _Toy(nn.Module):
    def __init__(self, C, reduc, classifier):
        super().__init__()
        self.features = _identity_features(C)
        self.feature_reduc = reduc
        self.classifier = classifier


def test_custom_modules_golden(golden):
    from model.custom_modules import NormalizeL2, NormalizeL2Fun, Shift, ShiftFun
    g = golden("l2norm_shift.npz")
    x = t(g["x"])
    np.testing.assert_allclose(NormalizeL2()(x).numpy(), g["y"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(NormalizeL2Fun()(x).numpy(), g["y"], rtol=2e-6, atol=2e-7)     # legacy call form
    sh = Shift(37)
    assert float(sh.param.detach().abs().sum()) == 0.0
    sh.param.data = t(g["param"])
    np.testing.assert_array_equal(sh(x).detach().numpy(), g["y_shift"])
    np.testing.assert_array_equal(ShiftFun()(x, sh.param).detach().numpy(), g["y_shift"])
    # backward = analytic gradient of x / sqrt(|x|^2 + eps)
    xr = torch.randn(4, 9, dtype=torch.float64, requires_grad=True)
    torch.autograd.gradcheck(lambda v: NormalizeL2Fun.apply(v, 1e-10), (xr,))
    torch.autograd.gradcheck(lambda v, p: ShiftFun.apply(v, p), (xr, torch.randn(9, dtype=torch.float64, requires_grad=True)))


def test_tuneclassif_and_get_embeddings_golden(golden):
    from model.siamese import TuneClassif
    from train import classif_finetune as cf
    g = golden("gap_l2.npz")
    fmap = t(g["fmap"])                                            # (3,16,4,4)
    fc = nn.Linear(16, 11)
    with torch.no_grad():
        fc.weight.copy_(t(g["fc_w"])); fc.bias.copy_(t(g["fc_b"]))
    net = TuneClassif(_Toy(16, nn.Sequential(nn.AvgPool2d(4)), nn.Sequential(fc)), 11).eval()
    assert net.feature_size == 11 and net.classifier[0] is fc
    with torch.no_grad():
        np.testing.assert_allclose(net(fmap).numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    ds = [(fmap[i], "a", "p%d" % i) for i in range(3)]
    cf.P.test_pre_proc, cf.P.cuda_device, cf.P.test_batch_size = True, -1, 2
    cf.P.embeddings_classify = False
    emb = cf.get_embeddings(net, ds, -1, 16)
    np.testing.assert_allclose(emb.numpy(), g["desc"], rtol=2e-6, atol=2e-7)
    assert net.classifier[0] is fc                                # classifier restored after the pass
    cf.P.embeddings_classify = True
    emb = cf.get_embeddings(net, ds, -1, 11)
    np.testing.assert_allclose(emb.numpy(), g["desc_classify"], rtol=1e-5, atol=1e-6)
    # reduc=False: the first FC absorbs the pooling factor
    n2 = TuneClassif(_Toy(16, nn.Sequential(nn.AvgPool2d(4)), nn.Sequential(nn.Linear(16, 11))), 7, reduc=False)
    assert n2.classifier[0].in_features == 16 * 16 and n2.classifier[0].out_features == 7 and len(n2.feature_reduc) == 0


def test_descriptor_net_golden(golden):
    from model.siamese import DescriptorNet
    g = golden("descriptor_head.npz")
    net = DescriptorNet(_Toy(8, nn.Sequential(), nn.Sequential(nn.Linear(72, 12), nn.ReLU(), nn.Linear(12, 5))), 10, (3, 3)).eval()
    assert net.feature_size == int(g["feature_size"]) == 10
    assert sorted(k for k in net.state_dict() if k.startswith("feature_reduc1")) == \
        ["feature_reduc1.1.param", "feature_reduc1.2.bias", "feature_reduc1.2.weight"]
    with torch.no_grad():
        net.feature_reduc1[1].param.copy_(t(g["shift"]))
        net.feature_reduc1[2].weight.copy_(t(g["w"])); net.feature_reduc1[2].bias.copy_(t(g["b"]))
        np.testing.assert_allclose(net(t(g["fmap"])).numpy(), g["desc"], rtol=1e-5, atol=1e-6)
    d0 = DescriptorNet(_Toy(8, nn.Sequential(), nn.Sequential(nn.Linear(72, 12))), 0, (3, 3))
    assert d0.feature_size == 12                                   # feature_dim <= 0 -> classifier width


def test_classif_sub_golden(golden):
    from model.siamese import TuneClassifSub
    from train import classif_regions as cr
    g = golden("classif_sub.npz")
    fc = nn.Linear(16, 9)
    with torch.no_grad():
        fc.weight.copy_(t(g["w_r"]).view(9, 16)); fc.bias.copy_(t(g["b_r"]))
    sub = TuneClassifSub(_Toy(16, nn.Sequential(nn.AvgPool2d(3)), nn.Sequential(fc)), 9, (3, 3)).eval()
    assert isinstance(sub.classifier[0], nn.Conv2d) and sub.classifier[0].kernel_size == (1, 1)
    with torch.no_grad():
        out = sub(t(g["fmap_r"]))
    assert isinstance(out, list) and len(out) == 1
    np.testing.assert_allclose(out[0].numpy(), g["map_r"], rtol=1e-5, atol=1e-6)
    # AlexNet-like: first FC takes the (3,3) window, second becomes 1x1
    f0, f1 = nn.Linear(72, 12), nn.Linear(12, 6)
    with torch.no_grad():
        f0.weight.copy_(t(g["w0_a"]).reshape(12, 72)); f0.bias.copy_(t(g["b0_a"]))
        f1.weight.copy_(t(g["w1_a"]).reshape(6, 12)); f1.bias.copy_(t(g["b1_a"]))
    sub2 = TuneClassifSub(_Toy(8, nn.Sequential(), nn.Sequential(f0, nn.ReLU(), f1)), 6, (3, 3)).eval()
    assert sub2.classifier[0].kernel_size == (3, 3) and sub2.classifier[2].kernel_size == (1, 1)
    with torch.no_grad():
        np.testing.assert_allclose(sub2(t(g["fmap_a"]))[0].numpy(), g["map_a"], rtol=1e-5, atol=1e-6)
    # get_embeddings of the regions approach == best-location golden
    b = golden("best_location.npz")
    cr.P.test_pre_proc, cr.P.cuda_device, cr.P.test_batch_size = True, -1, 1
    ds = [(t(g["fmap_r"])[0], "a", "p")]
    emb = cr.get_embeddings(sub, ds, -1, 9)
    np.testing.assert_allclose(emb.numpy()[0], b["desc_r"], rtol=1e-5, atol=1e-6)
    for k in range(4):
        d = cr._best_location_descriptors(t(b["map%d" % k])[None])
        np.testing.assert_allclose(d.numpy()[0], b["desc%d" % k], rtol=2e-6, atol=2e-7)


def test_region_descriptor_net_golden(golden):
    from model.siamese import RegionDescriptorNet
    g = golden("region_desc.npz")
    for tag, k in (("k3", 3), ("k40", 40)):
        fc = nn.Linear(16, 9)
        with torch.no_grad():
            fc.weight.copy_(t(g["cw_" + tag]).view(9, 16)); fc.bias.copy_(t(g["cb_" + tag]))
        net = RegionDescriptorNet(_Toy(16, nn.Sequential(nn.AvgPool2d(3)), nn.Sequential(fc)), k, 12, (3, 3)).eval()
        with torch.no_grad():
            net.feature_reduc1[1].param.copy_(t(g["shift_" + tag]))
            net.feature_reduc1[2].weight.copy_(t(g["w_" + tag])); net.feature_reduc1[2].bias.copy_(t(g["b_" + tag]))
            d = net(t(g["fmap_" + tag]))
            np.testing.assert_allclose(d.numpy(), g["desc_" + tag], rtol=1e-5, atol=1e-6)
            desc, cls_out = net.forward_single(t(g["fmap_" + tag]))
        assert cls_out.shape == (1, 9, k)
        cls = g["cls_" + tag][0].reshape(9, -1)
        n = min(k, cls.shape[1])
        np.testing.assert_allclose(cls_out[0, :, :n].numpy(), cls[:, g["idx_" + tag]], rtol=1e-5, atol=1e-6)


def test_metrics_and_descriptor_eval_golden(golden):
    from utils import embeddings_device_dim, mean_avg_precision, precision1, test_descriptor_net
    g = golden("synthetic_retrieval.npz")
    for n in (100, 1000):
        tg = "_n%d" % n
        ts = [(None, int(l), None) for l in g["qlab" + tg]]
        rs = [(None, int(l), None) for l in g["glab" + tg]]
        sim = t(g["sim" + tg])
        assert mean_avg_precision(sim, ts, rs) == float(g["map" + tg])
        p = precision1(sim, ts, rs)
        assert p[:3] == tuple(g["p1" + tg]) and p[3].shape == (len(ts), 1)

    class P:
        cuda_device, feature_dim, embeddings_cuda_size, train_bn = -1, 32, 2 ** 30, False
    class Net:
        feature_size = 32
    assert embeddings_device_dim(P, Net, 10) == (-1, 32)
    P.cuda_device, P.feature_dim = 0, 0
    assert embeddings_device_dim(P, Net, 10) == (0, 32)
    assert embeddings_device_dim(P, Net, 2 ** 24) == (-1, 32)                  # slab over budget -> CPU
    # an n x n matrix over budget no longer sends anything to the CPU: its consumers work on row blocks
    assert embeddings_device_dim(P, Net, 20000, sim_matrix=True) == (0, 32)
    P.cuda_device, P.feature_dim = -1, 32
    Q, G = t(g["Q_n100"]), t(g["G_n100"])
    ts = [(Q[i], int(l), None) for i, l in enumerate(g["qlab_n100"])]
    rs = [(G[i], int(l), None) for i, l in enumerate(g["glab_n100"])]
    emb = lambda net, ds, d, o: torch.stack([x for x, _, _ in ds])
    res = test_descriptor_net(P, emb, Net, ts, rs)
    assert res[1:3] == (int(g["p1_n100"][1]), int(g["p1_n100"][2]))
    assert abs(res[6] - float(g["map_n100"])) <= 1e-4
    sp, sa = O.masked_sums((Q @ G.t()).numpy(), g["qlab_n100"], g["glab_n100"])
    assert abs(res[3] - sp) < 1e-4 and abs(res[3] + res[4] - sa) < 1e-4


def _siamese_eval_sets(g, tag, kth):
    Q, G = t(g["Q_" + tag]), t(g["G_" + tag])
    rs = [(G[i], "L%d" % int(l), None) for i, l in enumerate(g["glab_" + tag])]
    if kth == 2:                                    # the reference's train-against-train form: queries drawn from the gallery
        ts = [rs[int(i)] for i in g["self_idx_" + tag]]
    else:
        ts = [(Q[i], "L%d" % int(l), None) for i, l in enumerate(g["qlab_" + tag])]
    return ts, rs


def check_siamese_eval_against_reference(g, meta, device, tag, kth):
    """utils.train_siamese.test_descriptor_net against what the REFERENCE'S OWN utils/train_siamese.py:61-82 returned on the same
    descriptors (tests/golden/siamese_eval.*, oracle/gen_golden.py::siamese_eval).  Counts, labels and AP are exact; the three
    score sums are fp32 sums of up to 143 000 terms in the reference (a Python `sum` of tensor elements / Tensor.sum), float64 here:
    tolerance 1e-6 per unit of sum |term| <= n terms."""
    from utils import test_descriptor_net

    class P:
        cuda_device, feature_dim, embeddings_cuda_size, train_bn = device, 32, 2 ** 30, False

    class Net:
        feature_size = 32
    ts, rs = _siamese_eval_sets(g, tag, kth)
    calls = []

    def emb(net, ds, d, o):
        calls.append([len(ds), d, o])
        e = torch.stack([x for x, _, _ in ds])
        return e.cuda(d) if d >= 0 else e
    key = "%s_kth%d" % (tag, kth)
    prec1, correct, total, sum_pos, sum_neg, sum_max, mAP, lab_dict = test_descriptor_net(P, emb, Net, ts, rs, kth)
    want = meta["runs"][key]
    assert [c[0] for c in calls] == [c[0] for c in want["get_embeddings_calls"]] and all(c[2] == 32 for c in calls)
    assert (prec1, correct, total) == tuple(g["p1_" + key])
    if device < 0:
        assert mAP == float(g["map_" + key])        # float64, term by term the reference's loop on the same torch.mm scores
    else:                                           # isx_cosine_sim's fma chain vs torch.mm: a near-tie may swap (north star: 1e-4)
        assert abs(mAP - float(g["map_" + key])) <= 1e-6
    assert lab_dict == want["lab_dict"]             # incl. the setdefault(lab, get(lab, 0) + 1) quirk: every count is 1
    tol = 1e-6 * len(ts) * len(rs) ** 0.5
    got = np.array([sum_pos, sum_neg, sum_max])
    np.testing.assert_allclose(got, g["sums_" + key], rtol=0, atol=max(tol, 1e-4))


@pytest.mark.parametrize("tag,kth", [("n100", 1), ("n100", 2), ("n1000", 1), ("n1000", 2)])
def test_descriptor_net_evaluation_matches_the_reference_run(golden, tag, kth):
    g = golden("siamese_eval.npz")
    meta = json.load(open(os.path.join(GOLDEN, "siamese_eval.json")))
    check_siamese_eval_against_reference(g, meta, -1, tag, kth)
    # the oracle's masked sums (what the GPU kernel is tested against) agree with the reference's sums too
    ts, rs = _siamese_eval_sets(g, tag, kth)
    sim = torch.stack([x for x, _, _ in ts]) @ torch.stack([x for x, _, _ in rs]).t()
    ids = {}
    gl = np.array([ids.setdefault(l, len(ids)) for _, l, _ in rs], np.int32)
    ql = np.array([ids.setdefault(l, len(ids)) for _, l, _ in ts], np.int32)
    sp, sa = O.masked_sums(sim.numpy(), ql, gl)
    want = g["sums_%s_kth%d" % (tag, kth)]
    assert abs(sp - want[0]) < 1e-3 and abs(sa - sp - want[1]) < 1e-3


def test_embeddings_device_dim_and_get_similarities_match_the_reference_run(golden):
    """a14: every branch of the reference's embeddings_device_dim (utils/train_siamese.py:30-43) as the reference itself
    answered, with ONE deliberate deviation: an n x n score matrix beyond the budget no longer sends the slab to the CPU
    (its consumers work on query-row blocks, SimilarityRows) -- asserted as such, not hidden."""
    from utils import embeddings_device_dim, get_similarities
    g = golden("siamese_eval.npz")
    meta = json.load(open(os.path.join(GOLDEN, "siamese_eval.json")))
    deviations = 0
    for c in meta["embeddings_device_dim"]:
        class P:
            cuda_device, feature_dim, embeddings_cuda_size = c["cuda_device"], c["feature_dim"], c["budget"]
        net = type("Net", (), {"feature_size": c["feature_size"]} if c["feature_size"] is not None else {})()
        dev, out = embeddings_device_dim(P, net, c["n"], c["sim_matrix"])
        assert out == c["out_size"]
        if c["sim_matrix"] and c["n"] * c["n"] * 4 > c["budget"] and c["n"] * out * 4 <= c["budget"]:
            assert (dev, c["device"]) == (c["cuda_device"], -1)
            deviations += 1
        else:
            assert dev == c["device"]
    assert deviations == 1

    class P:
        cuda_device, feature_dim, embeddings_cuda_size, train_bn = -1, 32, 2 ** 30, False

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
            self.feature_size = 32
    for tag in ("n100", "n1000"):
        _, rs = _siamese_eval_sets(g, tag, 1)
        net = Net().train()
        sims, dev = get_similarities(P, lambda n_, ds, d, o: torch.stack([x for x, _, _ in ds]), net, rs)
        want = meta["runs"][tag + "_get_similarities"]
        assert dev == want["device"] and net.training == want["net_training"] and net.features[1].training == want["bn_training"]
        rows = sims if tag == "n100" else sims[::125]
        np.testing.assert_allclose(rows.numpy(), g["selfsim_" + tag], rtol=0, atol=2e-7)


# ------------------------------------------------------------------ entry points (BASELINE config 1: CPU plumbing)
def test_entry_points_cpu_plumbing(capsys):
    from test import classif_finetune_test, classif_regions_test, siamese_descriptor_test, siamese_regions_test
    from train import classif_finetune as cf
    torch.manual_seed(0)
    spec = "synthetic:CLICIDE_video_224sq:n=20:q=6:labels=4"
    p1, mAP = classif_finetune_test.main(spec, "alexnet", "", -1, False, 8, 2)
    out = capsys.readouterr().out
    assert "Classification (TEST): " in out and "Descriptor (TEST): " in out and "Descriptor (TEST DBA k=2): " in out
    line = [l for l in out.splitlines() if l.startswith("Descriptor (TEST): ")][0]
    assert line.endswith("acc: {0:.4f} - mAP:{1:.4f}".format(p1, mAP))
    assert cf.P.feature_dim == 9216 and cf.P.feature_size2d == (6, 6) and len(cf.labels) == 4
    # the printed metric equals the oracle's on the same embeddings
    net = cf.get_class_net().eval()
    from test._common import load_sets
    labs = []
    qs, rs = load_sets(spec, labs)
    cf.P.embeddings_classify = False
    E_q, E_r = cf.get_embeddings(net, qs, -1, 9216), cf.get_embeddings(net, rs, -1, 9216)
    ids = {l: i for i, l in enumerate(labs)}
    ql = np.array([ids[l] for _, l, _ in qs], np.int32); gl = np.array([ids[l] for _, l, _ in rs], np.int32)
    sim = (E_q @ E_r.t()).numpy()
    ap = O.average_precision(O.rank_full(sim), ql, gl)
    assert abs(O.mean_avg_precision(ap) - mAP) <= 1e-12
    assert classif_regions_test.main("synthetic:CLICIDE_video_224sq:n=8:q=3:labels=2:size=288", "alexnet", "", -1, 0) is not None
    assert siamese_descriptor_test.main("synthetic:CLICIDE_video_224sq:n=8:q=3:labels=2", "alexnet", "", -1, 32, 4, 0) is not None
    assert siamese_regions_test.main("synthetic:CLICIDE_video_224sq:n=6:q=3:labels=2:size=288", "alexnet", "", -1, 16, 3, 0) is not None


def test_fc7_tap_is_classifier_prefix_pinned_against_torch(capsys):
    """BASELINE configs[0] "AlexNet fc7" (extension, SURVEY 8 a17 note): descriptor = classifier[:6] of the reference's AlexNet topology
    (model/ModelDefinition.py:31-37) -- 4096-d, through the second ReLU -- L2-normalised.  Pinned against plain torch on the same net:
    features -> flatten -> Dropout(eval) -> Linear -> ReLU -> Dropout(eval) -> Linear -> ReLU -> x / sqrt(sum x^2 + 1e-10)."""
    from test import classif_finetune_test
    from test._common import load_sets
    from train import classif_finetune as cf
    torch.manual_seed(0)
    spec = "synthetic:CLICIDE_video_224sq:n=12:q=4:labels=3"
    p1, mAP = classif_finetune_test.main(spec, "alexnet", "", -1, False, 8, 0, fc7=True)
    assert cf.P.feature_dim == 4096 and cf.P.embeddings_fc7
    assert "Descriptor (TEST): " in capsys.readouterr().out
    net = cf.get_class_net().eval()
    labs = []
    qs, _ = load_sets(spec, labs)
    E = cf.get_embeddings(net, qs, -1, 4096)
    assert tuple(E.shape) == (len(qs), 4096) and len(net.classifier) == 7            # classifier restored after the pass
    x = torch.stack([im for im, _, _ in qs])
    with torch.no_grad():
        f = net.features(x).reshape(len(qs), -1)
        h = torch.relu(torch.nn.functional.linear(f, net.classifier[1].weight, net.classifier[1].bias))
        h = torch.relu(torch.nn.functional.linear(h, net.classifier[4].weight, net.classifier[4].bias))
        want = h / (h.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()
    assert torch.equal(E, want)
    # a ResNet head has no fc7: refused loudly, not silently pooled
    with pytest.raises(ValueError):
        classif_finetune_test.main("synthetic:CLICIDE_video_224sq:n=4:q=2:labels=2", "resnet50", "", -1, False, 4, 0, fc7=True)
    cf.P.embeddings_fc7 = False


def test_instance_avg_matches_reference_loop():
    """DBA restated batched == the reference's per-item loop (test/instance_avg.py:7-33) replayed here."""
    from test.instance_avg import instance_avg
    g = torch.Generator().manual_seed(3)
    E = torch.nn.functional.normalize(torch.randn(14, 10, generator=g), dim=1)
    labs = [i % 4 for i in range(14)] + []
    labs[13] = 99                                                  # singleton instance: kept as is
    ds = [(None, l, None) for l in labs]
    for k in (-1, 0, 1, 2):
        got, _ = instance_avg(-1, E, ds, sorted(set(labs)), k)
        sim = E @ E.t()
        want = E.clone()
        for i, l in enumerate(labs):
            same = torch.tensor([x == l for x in labs])
            nn_ = int(same.sum()) - 1
            if 0 <= k < nn_:
                nn_ = k
            if nn_ <= 0:
                continue
            row = sim[i].clone(); row[i] = -2; row[~same] = -2
            best = row.sort(descending=True, stable=True).indices
            agg = E[i].clone()
            for j in range(nn_):
                agg += E[best[j]] * ((nn_ - j) / float(nn_ + 1))
            want[i] = agg / (agg.norm() + 1e-10)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)


def test_instance_avg_blocks_match_oracle_without_nxn():
    """The CPU path of DBA walks the label blocks batched by block size (no N x N matrix): equal to the oracle's restatement of the
    reference loop on ragged instance sizes, k below / above the instance sizes, singletons."""
    from test.instance_avg import instance_avg
    rng = np.random.default_rng(5)
    E = rng.standard_normal((120, 24)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    labs = rng.integers(0, 17, 120).astype(np.int32)
    labs[3] = 99
    ds = [(None, int(l), None) for l in labs]
    for k in (-1, 1, 4, 50):
        got, _ = instance_avg(-1, torch.from_numpy(E), ds, None, k)
        np.testing.assert_allclose(got.numpy(), O.dba(E, labs, k), rtol=1e-5, atol=1e-6)


def test_instance_avg_matches_reference_fixture(golden):
    """DBA against the output of the reference's OWN test/instance_avg.py:7-33 (run unmodified by oracle/gen_golden.py with
    a harness-side uint8-mask shim): singleton labels kept, k = -1 / 0 / 1 / 2 / 5."""
    from test.instance_avg import instance_avg
    g = golden("dba.npz")
    E = torch.from_numpy(g["emb"])
    labs = ["L%d" % l for l in g["labels"]]
    ds = [(None, l, None) for l in labs]
    for key, k in (("kall", -1), ("k0", 0), ("k1", 1), ("k2", 2), ("k5", 5)):
        got, same_ds = instance_avg(-1, E, ds, sorted(set(labs)), k)
        assert same_ds is ds
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=1e-6, err_msg=key)
    np.testing.assert_array_equal(g["k0"], g["emb"])                     # k = 0: unchanged


def test_region_embeddings_bucket_ragged_images_by_shape():
    """Ragged region datasets (the reference walks them one image per step, train/classif_regions.py:107-132): images are
    bucketed by shape and batched per bucket; every row lands at its dataset index and equals the one-at-a-time result."""
    from isx import backbones
    from model.siamese import RegionDescriptorNet, TuneClassifSub
    from train import classif_regions as cr, siamese_regions as sr
    from train._common import fold_shape_buckets
    from utils.dataset import synthetic_image_set
    a = synthetic_image_set(5, 3, size=(3, 288, 288), seed=1)
    b = synthetic_image_set(4, 3, size=(3, 320, 288), seed=2)
    ds = [a[0], b[0], a[1], b[1], b[2], a[2], a[3], b[3], a[4]]                       # interleaved sizes
    seen = []
    fold_shape_buckets(lambda ii, items: seen.append((ii, [tuple(t.shape) for t, _, _ in items])), ds, 3)
    assert [ii for ii, _ in seen] == [[0, 2, 5], [6, 8], [1, 3, 4], [7]]
    assert all(len(set(shapes)) == 1 for _, shapes in seen)
    # stage = (trans, device): the batch arrives staged (here: CPU, a plain stack), same buckets, rows in dataset order; a launch the device cannot
    # hold is retried as two halves with batches staged on the spot
    staged, ooms = [], []

    def f(ii, items, x):
        if len(ii) == 3 and not ooms:
            ooms.append(ii)
            raise torch.cuda.OutOfMemoryError("simulated")
        assert x.shape[0] == len(ii) == len(items) and all(torch.equal(x[k], items[k][0]) for k in range(len(ii)))
        staged.append(ii)

    fold_shape_buckets(f, ds, 3, stage=(None, -1))
    assert ooms == [[0, 2, 5]] and staged == [[0], [2, 5], [6, 8], [1, 3, 4], [7]]
    # test_classif_net walks the same buckets (the reference: one image per pass) and counts every image once
    torch.manual_seed(0)
    cr.P.cuda_device, cr.P.test_pre_proc, cr.P.test_batch_size = -1, True, 4
    del cr.labels[:]
    cr.labels.extend(sorted(set(l for _, l, _ in ds)))
    sub0 = TuneClassifSub(backbones.alexnet(pretrained=True), 3, (6, 6)).eval()
    c4, t4 = cr.test_classif_net(sub0, ds)
    cr.P.test_batch_size = 1
    c1, t1 = cr.test_classif_net(sub0, ds)
    assert t4 == t1 == len(ds) and c4 == c1
    torch.manual_seed(0)
    sub = TuneClassifSub(backbones.alexnet(pretrained=True), 3, (6, 6)).eval()
    for P_ in (cr.P, sr.P):
        P_.cuda_device, P_.test_pre_proc = -1, True
    cr.P.test_batch_size = 4
    batched = cr.get_embeddings(sub, ds, -1, 3)
    cr.P.test_batch_size = 1
    single = cr.get_embeddings(sub, ds, -1, 3)
    np.testing.assert_allclose(batched.numpy(), single.numpy(), rtol=1e-5, atol=1e-6)
    rd = RegionDescriptorNet(backbones.alexnet(pretrained=True), 3, 8, (6, 6)).eval()
    sr.P.test_batch_size = 4
    batched = sr.get_embeddings(rd, ds, -1, 8)
    sr.P.test_batch_size = 1
    single = sr.get_embeddings(rd, ds, -1, 8)
    np.testing.assert_allclose(batched.numpy(), single.numpy(), rtol=1e-5, atol=1e-6)
    assert abs(float(batched.norm(dim=1).mean()) - 1.0) < 1e-5


def test_blocked_metrics_equal_the_one_matrix_evaluation(golden):
    """utils.metrics.retrieval_metrics in query-row blocks (any budget) == precision1 + mean_avg_precision on the whole
    matrix, bit for bit; SimilarityRows mining == whole-matrix mining."""
    from train.siamese_descriptor import mine_epoch_negatives
    from utils import mean_avg_precision, precision1
    from utils.metrics import retrieval_metrics, row_blocks
    from utils.train_siamese import SimilarityRows
    g = golden("synthetic_retrieval.npz")
    Q, G = t(g["Q_n1000"]), t(g["G_n1000"])
    Q, G = Q / Q.norm(dim=1, keepdim=True), G / G.norm(dim=1, keepdim=True)
    ts = [(None, int(l), None) for l in g["qlab_n1000"]]
    rs = [(None, int(l), None) for l in g["glab_n1000"]]
    sim = Q @ G.t()
    for kth in (1, 2):
        want_p = precision1(sim, ts, rs, kth)
        want_map = mean_avg_precision(sim, ts, rs, kth)
        for budget in (None, 4 * 1000 * 7, 4 * 1000):                       # one block, 7-row blocks, single rows
            m = retrieval_metrics(Q, G, ts, rs, kth, budget_bytes=budget, with_sums=True)
            assert (m["prec1"], m["correct"], m["total"]) == want_p[:3] and m["max_label"] == want_p[4]
            # CPU BLAS sums a row block in another order than the whole matrix (last-bit score differences); on the GPU
            # every score is its own k-ordered fma chain and the blocked evaluation is bit-identical (tests/test_gpu_parity.py)
            np.testing.assert_allclose(m["max_sim"].numpy(), want_p[3].numpy(), rtol=0, atol=2e-7)
            assert abs(m["mAP"] - want_map) <= 1e-12
            assert m["blocks"] == len(row_blocks(Q.size(0), G.size(0), budget))
        assert abs(m["sum_all"] - float(sim.double().sum())) < 1e-4            # same CPU-BLAS caveat
    assert row_blocks(10, 1000, 4 * 1000 * 3) == [(0, 3), (3, 6), (6, 9), (9, 10)] and row_blocks(0, 5) == []
    # mining on row blocks
    import utils.metrics as M
    ds = [(None, int(l), None) for l in g["glab_n100"]]
    E = t(g["G_n100"]); E = E / E.norm(dim=1, keepdim=True)
    full = E @ E.t()
    couples = [(ds[i][1], (i, j), (None, None)) for i in range(100) for j in range(i, 100) if ds[i][1] == ds[j][1]][:150]
    old = M.SIM_BUDGET_BYTES
    try:
        M.SIM_BUDGET_BYTES = 4 * 100 * 9
        for semi in (True, False):
            a = mine_epoch_negatives(full, ds, couples, semi)
            b = mine_epoch_negatives(SimilarityRows(E), ds, couples, semi)
            assert torch.equal(a, b)
    finally:
        M.SIM_BUDGET_BYTES = old


def test_folder_decode_on_threads_keeps_file_order(tmp_path, monkeypatch):
    """test/_common._decode_all: the folder images of a test run are decoded on a pool of threads; same tensors, same order as the
    sequential read (ISX_DECODE_THREADS=1)."""
    from PIL import Image
    from test import _common as C
    rng = np.random.default_rng(5)
    files = []
    for i in range(24):
        f = str(tmp_path / ("img_%02d.png" % i))
        Image.fromarray(rng.integers(0, 255, (9 + i % 3, 11, 3), dtype=np.uint8)).save(f)
        files.append(f)
    load = lambda f: C.to_raw_tensor(C.imread_rgb(f))
    monkeypatch.setenv("ISX_DECODE_THREADS", "1")
    seq = C._decode_all(load, files)
    monkeypatch.setenv("ISX_DECODE_THREADS", "4")
    par = C._decode_all(load, files)
    assert len(seq) == len(par) == 24
    for a, b in zip(seq, par):
        assert a.dtype == torch.uint8 and a.shape == b.shape and torch.equal(a, b)



def test_row_segments_choice_is_a_divisor_and_bounded():
    """ops._row_segments (few-query top-k by row segments): S divides N, every segment holds >= max(k, 1024) columns, the merge fits
    isx_topk_merge (S * k <= 4096), and large query blocks / short rows keep the one-launch path."""
    from isx import ops
    for M, N, k in ((1000, 100000, 100), (1000, 100000, 1), (100, 100000, 100), (37, 65536, 1), (5, 49152, 256), (1000, 20000, 10)):
        S = ops._row_segments(M, N, k)
        assert S >= 2 and N % S == 0 and N // S >= max(k, 1024) and S * k <= 4096 and S <= 64
    for M, N, k in ((5000, 100000, 100), (1000, 10000, 100), (3, 100003, 7), (10, 100000, 300)):
        assert ops._row_segments(M, N, k) == 0


def test_lazy_gallery_ingest_matches_the_decoded_list(tmp_path, monkeypatch):
    """train._common.LazyImage: the gallery of a folder dataset as (LazyImage, label, path) tuples -- labels / paths / len / slices are those of the
    decoded list, a staged batch has the same bytes, a finished batch drops its decoded images, a ragged folder is refused with a clear message
    (reference test/classif_finetune_test.py:62-73 decodes the whole folder before the first batch)."""
    import numpy as np
    from PIL import Image
    from test import _common as C
    from train import _common as TC
    root = tmp_path / "CLICIDE_video_224sq"
    (root / "test").mkdir(parents=True)
    (tmp_path / "data").mkdir()
    (tmp_path / "data" / "CLICIDE_224sq_train_ms.txt").write_text("0.485 0.456 0.406\n0.229 0.224 0.225\n")
    rng = np.random.default_rng(0)
    for lab in "ab":
        for i in range(5):
            Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(root / ("%s-%d.png" % (lab, i)))
        Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(root / "test" / ("%s-9.png" % lab))
    monkeypatch.chdir(tmp_path)
    l1, l2 = [], []
    q, ref = C.load_sets(str(root), l1, raw=True, lazy=True)
    q2, ref2 = C.load_sets(str(root), l2, raw=True, lazy=False)
    assert TC.is_lazy(ref) and not TC.is_lazy(q) and not TC.is_lazy(ref2) and l1 == l2
    assert [t[1:] for t in ref] == [t[1:] for t in ref2] and tuple(ref[0][0].shape) == (32, 32, 3) and ref[0][0].dtype == torch.uint8
    assert TC.make_resident(ref, 0) is None                                   # a lazy set is never copied wholesale
    for a in range(0, 10, 4):
        assert torch.equal(TC.stage_batch(ref[a:a + 4], None, -1), TC.stage_batch(ref2[a:a + 4], None, -1))
    assert all(t[0]._fut is None for t in ref)                                # decoded images are dropped once staged
    Image.fromarray(rng.integers(0, 256, (16, 32, 3), dtype=np.uint8)).save(root / "b-7.png")
    _, ragged = C.load_sets(str(root), [], raw=True, lazy=True)
    with pytest.raises(RuntimeError, match="same-sized"):
        TC.stage_batch(ragged, None, -1)


def test_gallery_slab_cli_round_trip_on_the_cpu(tmp_path, capsys):
    """--save-slab / --gallery-slab of the evaluation mains (extension; the reference re-extracts the gallery every run, test/classif_finetune_test.py:80-81):
    the second run ranks against the file and prints the same result line; a slab of another label set is refused."""
    from test import classif_finetune_test as T
    spec = "synthetic:CLICIDE_video_224sq:n=24:q=6:labels=4"
    f = str(tmp_path / "g.slab")
    torch.manual_seed(0)                                                      # the class-score layer is random-initialised per run
    r1 = T.main(spec, "alexnet", "", -1, True, 8, 0, save_slab=f)
    torch.manual_seed(0)
    r2 = T.main(spec, "alexnet", "", -1, True, 8, 0, gallery_slab=f)
    out = capsys.readouterr().out
    assert r1 == r2 and "descriptors written to" in out and "descriptors read from" in out
    with pytest.raises(ValueError, match="another label set"):
        T.main("synthetic:CLICIDE_video_224sq:n=24:q=6:labels=5", "alexnet", "", -1, True, 8, 0, gallery_slab=f)


def test_descriptor_head_skips_its_random_init_when_weights_follow(tmp_path):
    """get_siamese_net with P.preload_net: the head is allocated without the 0.6 s random initialisation and then filled from the file -- the
    loaded net equals the saved one; without a file the usual seeded init is untouched."""
    import copy
    from train import siamese_descriptor as sd
    saved = copy.copy(sd.P.__dict__)
    try:
        P = sd.P
        P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim, P.num_classes, P.classif_model, P.preload_net = -1, "alexnet", (6, 6), 16, 3, "", ""
        torch.manual_seed(0)
        a = sd.get_siamese_net()
        torch.manual_seed(0)
        b = sd.get_siamese_net()
        assert all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())            # seeded init unchanged
        f = str(tmp_path / "w.pth.tar")
        torch.save(a.state_dict(), f)
        P.preload_net = f
        calls = []
        real = torch.nn.Linear.reset_parameters
        torch.nn.Linear.reset_parameters = lambda self: calls.append((tuple(self.weight.shape), self.weight.device.type)) or real(self)
        try:
            c = sd.get_siamese_net()
        finally:
            torch.nn.Linear.reset_parameters = real
        assert ((16, 256 * 6 * 6), "meta") in calls and ((16, 256 * 6 * 6), "cpu") not in calls    # the head's Linear was not initialised (skip_init: on the meta device only) ...
        assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())            # ... and holds the file's weights
    finally:
        sd.P.__dict__.clear(); sd.P.__dict__.update(saved)


def test_thread_cap_follows_the_cpus_the_process_owns(monkeypatch):
    """utils.general.cap_torch_threads: torch's team is cut to affinity ∩ cgroup quota (boxes that show 256 CPUs and own 16) and never raised."""
    import builtins
    from utils import general as G
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == '/sys/fs/cgroup/cpu.max':
            import io
            return io.StringIO("200000 100000\n")          # a quota of two CPUs
        return real_open(path, *a, **k)

    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(builtins, "open", fake_open)
        assert G.usable_cpus() == min(2, len(os.sched_getaffinity(0)))
        torch.set_num_threads(max(before, 4))
        assert G.cap_torch_threads() == G.usable_cpus() == torch.get_num_threads()
        torch.set_num_threads(1)
        assert G.cap_torch_threads() == 1                    # a smaller explicit setting is left alone
    finally:
        monkeypatch.undo()
        torch.set_num_threads(before)


def test_training_entry_point_reads_a_folder_of_image_files(tmp_path, monkeypatch):
    """train.siamese_descriptor.run on a dataset FOLDER (reference train/siamese_descriptor.py:166-191): the images of the folder are the training
    set and the gallery, those of `test/` with a known label the queries; labels come from the file names, mean / std from the dataset's file."""
    import copy
    from PIL import Image
    from train import siamese_descriptor as sd
    root = tmp_path / "CLICIDE_video_224sq"
    (root / "test").mkdir(parents=True)
    (tmp_path / "data").mkdir()
    (tmp_path / "data" / "CLICIDE_224sq_train_ms.txt").write_text("0.4 0.5 0.6\n0.2 0.3 0.25\n")
    rng = np.random.default_rng(0)
    for lab in ("a", "b", "c"):
        for i in range(3):
            Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(root / ("%s-%d.png" % (lab, i)))
    for lab in ("a", "c", "zz"):                               # "zz" is not a gallery label: dropped
        Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(root / "test" / ("%s-9.png" % lab))
    monkeypatch.chdir(tmp_path)
    saved = copy.copy(sd.P.__dict__)
    seen = []
    monkeypatch.setattr(sd, "train_siam_triplets_pos_couples", lambda net, train_set, testset_tuple, *a, **k: seen.append((train_set, testset_tuple)) or 1)
    monkeypatch.setattr(sd, "test_print_descriptor", lambda *a, **k: 0)
    monkeypatch.setattr(sd, "get_siamese_net", lambda: torch.nn.Linear(2, 2))
    try:
        sd.P.cuda_device, sd.P.cnn_model = -1, "alexnet"
        sd.run(str(root))
        train_set, (test_set, test_train_set) = seen[0]
        assert len(train_set) == 9 and train_set is test_train_set and len(test_set) == 2 and sd.labels == ["a", "b", "c"]
        im, lab, path = train_set[0]
        assert im.shape == (3, 32, 32) and im.dtype == torch.float32 and path.endswith(".png")
        raw = torch.from_numpy(np.asarray(Image.open(path).convert("RGB")).copy()).permute(2, 0, 1).float() / 255.0
        want = (raw - torch.tensor([0.4, 0.5, 0.6]).view(3, 1, 1)) / torch.tensor([0.2, 0.3, 0.25]).view(3, 1, 1)
        assert torch.allclose(im, want, atol=1e-6)
    finally:
        sd.P.__dict__.clear(); sd.P.__dict__.update(saved)
