"""Pins the CPU oracle (oracle/isx_oracle.c) against golden vectors produced by the
reference's own code (oracle/gen_golden.py -> tests/golden/).  CPU only."""
import json
import os

import numpy as np

import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=2e-6, atol=2e-7)   # fp32: summation-order slack only


def test_l2norm_and_shift(golden):
    g = golden("l2norm_shift.npz")
    y = O.l2norm_rows(g["x"])
    np.testing.assert_allclose(y, g["y"], **TOL)
    assert np.all(y[2] == 0.0)                                   # zero row stays zero (eps inside the sqrt)
    np.testing.assert_allclose(O.l2norm_rows(g["x_wide"]), g["y_wide"], **TOL)
    np.testing.assert_array_equal(O.shift_rows(g["x"], g["param"]), g["y_shift"])


def test_gap_l2(golden):
    g = golden("gap_l2.npz")
    np.testing.assert_allclose(O.gap_l2(g["fmap"]), g["desc"], **TOL)
    np.testing.assert_allclose(O.gap_l2(g["fmap7"]), g["desc7"], **TOL)
    # embeddings_classify=True path: pooled -> FC -> L2 (train/classif_finetune.py:87-100)
    pooled = g["fmap"].mean(axis=(2, 3), dtype=np.float64).astype(np.float32)
    logits = pooled @ g["fc_w"].T + g["fc_b"]
    np.testing.assert_allclose(logits, g["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.l2norm_rows(g["logits"]), g["desc_classify"], **TOL)


def test_descriptor_head(golden):
    g = golden("descriptor_head.npz")
    fm = g["fmap"]
    x = O.l2norm_rows(fm.reshape(fm.shape[0], -1))
    x = O.shift_rows(x, g["shift"])
    x = x @ g["w"].T + g["b"]
    np.testing.assert_allclose(O.l2norm_rows(x), g["desc"], rtol=1e-5, atol=1e-6)


def _conv_valid(x, w, b):
    # x (C,H,W), w (O,C,kh,kw)
    O_, C, kh, kw = w.shape
    H, W = x.shape[1:]
    out = np.zeros((O_, H - kh + 1, W - kw + 1), np.float64)
    for i in range(out.shape[1]):
        for j in range(out.shape[2]):
            out[:, i, j] = np.tensordot(w.astype(np.float64), x[:, i:i + kh, j:j + kw].astype(np.float64), axes=3) + b
    return out.astype(np.float32)


def test_classif_sub_maps(golden):
    g = golden("classif_sub.npz")
    pooled = O.boxpool_s1(g["fmap_r"], 3, 3)
    np.testing.assert_allclose(pooled, g["pooled_r"], **TOL)
    m = _conv_valid(pooled[0], g["w_r"], g["b_r"])
    np.testing.assert_allclose(m, g["map_r"][0], rtol=1e-5, atol=1e-6)
    # AlexNet-like: first FC convolutionalised with kernel fs, second 1x1 (model/siamese.py:73-80)
    h = np.maximum(_conv_valid(g["fmap_a"][0], g["w0_a"], g["b0_a"]), 0)
    m2 = _conv_valid(h, g["w1_a"], g["b1_a"])
    np.testing.assert_allclose(m2, g["map_a"][0], rtol=1e-5, atol=1e-6)


def test_best_location(golden):
    g = golden("best_location.npz")
    for t in range(4):
        d, loc = O.best_location_desc(g["map%d" % t])
        assert tuple(loc) == tuple(g["locs"][t])
        np.testing.assert_allclose(d, g["desc%d" % t], **TOL)
    d, loc = O.best_location_desc(g["map_r"])
    assert tuple(loc) == tuple(g["loc_r"])
    np.testing.assert_allclose(d, g["desc_r"], **TOL)


def test_region_descriptor(golden):
    g = golden("region_desc.npz")
    for tag, k in (("k3", 3), ("k40", 40)):
        cls = g["cls_" + tag][0]
        idx, sc = O.region_topk(cls, k)
        np.testing.assert_array_equal(idx, g["idx_" + tag])
        assert len(idx) == min(k, cls.shape[1] * cls.shape[2])
        rows = O.region_gather_l2(g["fmap_" + tag][0], 3, 3, idx, cls.shape[2], shift=g["shift_" + tag])
        acc = (rows @ g["w_" + tag].T + g["b_" + tag]).sum(axis=0, keepdims=True)
        np.testing.assert_allclose(O.l2norm_rows(acc), g["desc_" + tag], rtol=1e-5, atol=1e-6)


def test_metrics(golden):
    g = golden("metrics.npz")
    sim, ql, gl = g["sim"], g["qlab"], g["glab"]
    ranked = O.rank_full(sim)
    ts, ti = O.topk_rows(sim, 3)
    np.testing.assert_array_equal(ti, ranked[:, :3])
    for kth in (1, 2, 3):
        ap = O.average_precision(ranked, ql, gl, kth)
        ref = g["ap_kth%d" % kth]
        np.testing.assert_array_equal(np.isnan(ap), np.isnan(ref))
        np.testing.assert_array_equal(ap[~np.isnan(ap)], ref[~np.isnan(ref)])      # float64, bit-exact
        assert O.mean_avg_precision(ap) == float(g["map_kth%d" % kth])
        p1, c, t, hit = O.precision1(ti, ql, gl, kth)
        assert (p1, c, t) == tuple(g["p1_kth%d" % kth])
        np.testing.assert_array_equal(hit, g["p1_maxlabel_kth%d" % kth])
        np.testing.assert_array_equal(ts[:, kth - 1], g["p1_maxsim_kth%d" % kth])
        # the literal per-query Python loop (timed by bench.py's cpu_baseline leg) against the reference's own values
        lit = [O.avg_precision_literal(sim[i], int(ql[i]), gl.tolist(), kth) for i in range(sim.shape[0])]
        assert [x is None for x in lit] == list(np.isnan(ref))
        assert [x for x in lit if x is not None] == list(ref[~np.isnan(ref)])
    rt = O.rank_full(g["tie_sim"])
    apt = O.average_precision(rt, g["tie_qlab"], g["tie_glab"], 1)
    np.testing.assert_array_equal(apt, g["tie_ap"])
    assert O.mean_avg_precision(apt) == float(g["tie_map"])


def test_synthetic_retrieval(golden):
    g = golden("synthetic_retrieval.npz")
    for n in (100, 1000):
        t = "_n%d" % n
        Q, G = g["Q" + t], g["G" + t]
        sim = O.cosine_sim(Q, G)
        assert np.abs(sim - g["sim" + t]).max() <= 1e-5           # vs torch.mm fp32 (north_star tolerance)
        # reference metrics on the reference's own scores: bit-exact
        ranked = O.rank_full(g["sim" + t])
        ap = O.average_precision(ranked, g["qlab" + t], g["glab" + t], 1)
        assert O.mean_avg_precision(ap) == float(g["map" + t])
        ts, ti = O.topk_rows(g["sim" + t], 1)
        p1 = O.precision1(ti, g["qlab" + t], g["glab" + t])
        assert p1[:3] == tuple(g["p1" + t])
        # metrics on the oracle's own fma-chain scores: within the north_star's 1e-4
        ap2 = O.average_precision(O.rank_full(sim), g["qlab" + t], g["glab" + t], 1)
        assert abs(O.mean_avg_precision(ap2) - float(g["map" + t])) <= 1e-4
        # fused top-k == full rank prefix
        ts2, ti2 = O.cosine_topk(Q, G, 10, idx_base=5)
        np.testing.assert_array_equal(ti2 - 5, O.rank_full(sim)[:, :10])


def test_topk_merge_is_shard_invariant():
    rng = np.random.default_rng(3)
    Q = rng.standard_normal((7, 16), dtype=np.float32)
    G = rng.standard_normal((50, 16), dtype=np.float32)
    G[10] = G[30]                                                  # duplicate rows -> tied scores across shards
    full_s, full_i = O.cosine_topk(Q, G, 8)
    for P in (2, 5):
        bounds = np.linspace(0, 50, P + 1).astype(int)
        parts = [O.cosine_topk(Q, G[a:b], 8, idx_base=a) for a, b in zip(bounds[:-1], bounds[1:])]
        ms, mi = O.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
        np.testing.assert_array_equal(mi, full_i)
        np.testing.assert_array_equal(ms, full_s)
    # k larger than a shard: padding entries (-inf, -1) never win
    parts = [O.cosine_topk(Q, G[a:b], 8, idx_base=a) for a, b in ((0, 3), (3, 50))]
    assert (parts[0][1][:, 3:] == -1).all()
    ms, mi = O.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    np.testing.assert_array_equal(mi, full_i)


def test_masked_sums():
    rng = np.random.default_rng(5)
    sim = rng.standard_normal((6, 9), dtype=np.float32)
    ql = np.arange(6) % 3
    gl = np.arange(9) % 3
    sp, sa = O.masked_sums(sim, ql, gl)
    mask = ql[:, None] == gl[None, :]
    assert abs(sp - sim[mask].astype(np.float64).sum()) < 1e-12
    assert abs(sa - sim.astype(np.float64).sum()) < 1e-12


def _fold(w, g, b, m, v, eps):
    """BatchNorm (eval) folded into the preceding bias-free convolution, fp32 like torch's fuse_conv_bn_eval."""
    s = (g / np.sqrt(v + eps)).astype(np.float32)
    return (w * s.reshape(-1, 1, 1, 1)).astype(np.float32), (b - m * s).astype(np.float32)


def test_trunk_block(golden):
    """isxo_conv1x1_nhwc / isxo_conv3x3_nhwc + BN folding == torch's conv2d / batch_norm on two bottleneck blocks
    (projection shortcut with stride 2, then identity shortcut)."""
    g = golden("trunk_block.npz")
    eps = float(g["eps"])
    y = np.ascontiguousarray(g["x"].transpose(0, 2, 3, 1))                   # NHWC
    for blk, stride in ((0, 2), (1, 1)):
        f = [_fold(g["b%d_w%d" % (blk, i)], g["b%d_g%d" % (blk, i)], g["b%d_b%d" % (blk, i)], g["b%d_m%d" % (blk, i)], g["b%d_v%d" % (blk, i)], eps)
             for i in range(3)]
        B, H, W, C = y.shape
        idt = y
        bias3 = f[2][1]
        if "b%d_dw" % blk in g:
            dw, db = _fold(g["b%d_dw" % blk], g["b%d_dg" % blk], g["b%d_db" % blk], g["b%d_dm" % blk], g["b%d_dv" % blk], eps)
            xs = np.ascontiguousarray(y[:, ::stride, ::stride])
            idt = O.conv1x1_nhwc(xs.reshape(-1, C), dw.reshape(dw.shape[0], -1), np.zeros(dw.shape[0], np.float32), None, False)
            idt = idt.reshape(xs.shape[0], xs.shape[1], xs.shape[2], -1)
            bias3 = bias3 + db                                                # shortcut bias merged into the last epilogue
        t = O.conv1x1_nhwc(y.reshape(-1, C), f[0][0].reshape(f[0][0].shape[0], -1), f[0][1], None, True).reshape(B, H, W, -1)
        t = O.conv3x3_nhwc(t, np.ascontiguousarray(f[1][0].transpose(0, 2, 3, 1)), f[1][1], stride, None, True)
        Bo, Ho, Wo, Cm = t.shape
        y_in = y
        y = O.conv1x1_nhwc(t.reshape(-1, Cm), f[2][0].reshape(f[2][0].shape[0], -1), bias3, idt.reshape(-1, idt.shape[-1]), True)
        y = y.reshape(Bo, Ho, Wo, -1)
        np.testing.assert_allclose(y.transpose(0, 3, 1, 2), g["y%d" % blk], rtol=2e-5, atol=2e-5)
        if "b%d_dw" % blk in g:
            # the fused form (last convolution and projection shortcut as one fma chain over [t ; x_strided])
            w_cat = np.concatenate([f[2][0].reshape(f[2][0].shape[0], -1), dw.reshape(dw.shape[0], -1)], axis=1)
            y2 = O.conv1x1_dual_nhwc(t, y_in, w_cat, bias3, stride, True)
            np.testing.assert_allclose(y2.transpose(0, 3, 1, 2), g["y%d" % blk], rtol=2e-5, atol=2e-5)


def test_images_u8_to_f32_matches_torch():
    """ToTensor + Normalize restatement == torch's own arithmetic (bit-exact)."""
    import torch
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (2, 9, 11, 3), dtype=np.uint8)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float().div(255.0)
    ref = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    np.testing.assert_array_equal(O.images_u8_to_f32(img, mean, std), ref.numpy())


def test_dba_oracle_matches_reference_fixture(golden):
    """The oracle's restatement of test/instance_avg.py:7-33 against the output of the reference's own function (dba.npz: singleton labels
    kept, k = -1 / 0 / 1 / 2 / 5)."""
    g = golden("dba.npz")
    for key, k in (("kall", -1), ("k0", 0), ("k1", 1), ("k2", 2), ("k5", 5)):
        np.testing.assert_allclose(O.dba(g["emb"], g["labels"], k), g[key], rtol=2e-6, atol=2e-7, err_msg=key)
    np.testing.assert_array_equal(O.dba(g["emb"], g["labels"], 0), g["emb"])
