"""ctypes/numpy front-end of the CPU oracle (oracle/isx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under instance-search_amd/ imports it.

Every function takes/returns numpy arrays (fp32 C-contiguous, int64 indices,
int32 labels) and forwards to the C restatement; see the C file for the
reference citations (file:line) of each function.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libisx_oracle.so")
_lib = None

F32P = C.POINTER(C.c_float)
I64P = C.POINTER(C.c_int64)
I32P = C.POINTER(C.c_int32)
F64P = C.POINTER(C.c_double)


def build(force=False):
    """(Re)build libisx_oracle.so with gcc if missing or stale."""
    src = os.path.join(_HERE, "isx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libisx_oracle.so"])
    return _SO


def lib():
    """ISX_ORACLE_LIB: another build of the same source (tools/sanitize_cpu.sh loads the ASan + UBSan one)."""
    global _lib
    if _lib is None:
        alt = os.environ.get("ISX_ORACLE_LIB")
        if alt:
            _lib = C.CDLL(alt)
        else:
            build()
            _lib = C.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


def l2norm_rows(x, eps=1e-10):
    x = _f32(x); B, D = x.shape; y = np.empty_like(x)
    lib().isxo_l2norm_rows(_p(x, F32P), C.c_int64(B), C.c_int64(D), C.c_float(eps), _p(y, F32P))
    return y


def shift_rows(x, param):
    x = _f32(x); param = _f32(param); y = np.empty_like(x)
    lib().isxo_shift_rows(_p(x, F32P), _p(param, F32P), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]), _p(y, F32P))
    return y


def gap_l2(fmap, eps=1e-10):
    fmap = _f32(fmap); B, Cc, H, W = fmap.shape; y = np.empty((B, Cc), np.float32)
    lib().isxo_gap_l2(_p(fmap, F32P), C.c_int64(B), Cc, H, W, C.c_float(eps), _p(y, F32P))
    return y


def boxpool_s1(fmap, kh, kw):
    fmap = _f32(fmap); B, Cc, H, W = fmap.shape
    out = np.empty((B, Cc, H - kh + 1, W - kw + 1), np.float32)
    lib().isxo_boxpool_s1(_p(fmap, F32P), C.c_int64(B), Cc, H, W, kh, kw, _p(out, F32P))
    return out


def best_location_desc(cls, eps=1e-10):
    cls = _f32(cls); K, Hp, Wp = cls.shape[-3:]
    desc = np.empty((K,), np.float32); loc = np.empty((2,), np.int64)
    lib().isxo_best_location_desc(_p(cls, F32P), K, Hp, Wp, C.c_float(eps), _p(desc, F32P), _p(loc, I64P))
    return desc, loc


def region_topk(cls, k):
    cls = _f32(cls); K, Hp, Wp = cls.shape[-3:]
    idx = np.empty((k,), np.int64); sc = np.empty((k,), np.float32)
    lib().isxo_region_topk.restype = C.c_int
    n = lib().isxo_region_topk(_p(cls, F32P), K, Hp, Wp, k, _p(idx, I64P), _p(sc, F32P))
    return idx[:n].copy(), sc[:n].copy()


def region_gather_l2(fmap, kh, kw, flat_idx, Wp, shift=None, eps=1e-10):
    fmap = _f32(fmap); Cc, Hf, Wf = fmap.shape[-3:]
    flat_idx = np.ascontiguousarray(flat_idx, np.int64); k = flat_idx.shape[0]
    rows = np.empty((k, Cc * kh * kw), np.float32)
    sp = None
    if shift is not None:
        shift = _f32(shift); sp = _p(shift, F32P)
    lib().isxo_region_gather_l2(_p(fmap, F32P), Cc, Hf, Wf, kh, kw, _p(flat_idx, I64P), k, Wp, sp,
                                C.c_float(eps), _p(rows, F32P))
    return rows


def cosine_sim(Q, G):
    Q = _f32(Q); G = _f32(G); M, D = Q.shape; N = G.shape[0]
    sim = np.empty((M, N), np.float32)
    lib().isxo_cosine_sim(_p(Q, F32P), C.c_int64(M), _p(G, F32P), C.c_int64(N), D, _p(sim, F32P))
    return sim


def images_u8_to_f32(img, mean, std):
    img = np.ascontiguousarray(img, np.uint8); B, H, W, _ = img.shape
    out = np.empty((B, 3, H, W), np.float32)
    m = np.asarray(mean, np.float32); s_ = np.asarray(std, np.float32)
    lib().isxo_images_u8_to_f32(img.ctypes.data_as(C.c_void_p), C.c_int64(B), H, W, _p(m, F32P), _p(s_, F32P), _p(out, F32P))
    return out


def bias_relu_maxpool_nhwc(y, bias):
    y = _f32(y); bias = _f32(bias); B, H, W, Cc = y.shape
    out = np.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), np.float32)
    lib().isxo_bias_relu_maxpool_nhwc(_p(y, F32P), _p(bias, F32P), C.c_int64(B), H, W, Cc, _p(out, F32P))
    return out


def stem7x7_pool_nhwc(x, w_ohwi, bias):
    """x: (B,H,W,3), w_ohwi: (64,7,7,3) -> relu(conv7x7/2 + bias) -> maxpool 3/2/1: (B,Hp,Wp,64)."""
    x = _f32(x); w = _f32(w_ohwi); bias = _f32(bias); B, H, W, _ = x.shape
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = np.empty((B, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, 64), np.float32)
    lib().isxo_stem7x7_pool_nhwc(_p(x, F32P), C.c_int64(B), H, W, _p(w, F32P), _p(bias, F32P), _p(out, F32P))
    return out


def conv1x1_nhwc(x, w, bias, res=None, relu=True):
    """x: (M, Cin) pixels, w: (Cout, Cin), res: (M, Cout) or None -> (M, Cout)."""
    x = _f32(x); w = _f32(w); bias = _f32(bias); M, Cin = x.shape; Cout = w.shape[0]
    y = np.empty((M, Cout), np.float32)
    r = _f32(res) if res is not None else None
    lib().isxo_conv1x1_nhwc(_p(x, F32P), C.c_int64(M), Cin, _p(w, F32P), Cout, _p(bias, F32P),
                            _p(r, F32P) if r is not None else None, 1 if relu else 0, _p(y, F32P))
    return y


def conv1x1_dual_nhwc(t, x, w_cat, bias, stride=1, relu=True):
    """t: (B,Ho,Wo,K1), x: (B,H,W,K2), w_cat: (Cout, K1+K2) -> (B,Ho,Wo,Cout)."""
    t = _f32(t); x = _f32(x); w = _f32(w_cat); bias = _f32(bias)
    B, H, W, K2 = x.shape; K1 = t.shape[3]; Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert t.shape == (B, Ho, Wo, K1) and w.shape == (Cout, K1 + K2)
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    lib().isxo_conv1x1_dual_nhwc(_p(t, F32P), K1, _p(x, F32P), C.c_int64(B), H, W, K2, stride, _p(w, F32P), Cout, _p(bias, F32P),
                                 1 if relu else 0, _p(y, F32P))
    return y


def conv3x3_nhwc(x, w_ohwi, bias, stride=1, res=None, relu=True):
    """x: (B,H,W,Cin), w_ohwi: (Cout,3,3,Cin), res: (B,Ho,Wo,Cout) or None -> (B,Ho,Wo,Cout); padding 1."""
    x = _f32(x); w = _f32(w_ohwi); bias = _f32(bias); B, H, W, Cin = x.shape; Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    r = _f32(res) if res is not None else None
    lib().isxo_conv3x3_nhwc(_p(x, F32P), C.c_int64(B), H, W, Cin, _p(w, F32P), Cout, stride, _p(bias, F32P),
                            _p(r, F32P) if r is not None else None, 1 if relu else 0, _p(y, F32P))
    return y


def cosine_topk(Q, G, k, idx_base=0):
    Q = _f32(Q); G = _f32(G); M, D = Q.shape; N = G.shape[0]
    ts = np.empty((M, k), np.float32); ti = np.empty((M, k), np.int64)
    lib().isxo_cosine_topk(_p(Q, F32P), C.c_int64(M), _p(G, F32P), C.c_int64(N), D, k, C.c_int64(idx_base),
                           _p(ts, F32P), _p(ti, I64P))
    return ts, ti


def rank_full(sim):
    sim = _f32(sim); M, N = sim.shape; r = np.empty((M, N), np.int64)
    lib().isxo_rank_full(_p(sim, F32P), C.c_int64(M), C.c_int64(N), _p(r, I64P))
    return r


def topk_rows(sim, k, idx_base=0):
    sim = _f32(sim); M, N = sim.shape
    ts = np.empty((M, k), np.float32); ti = np.empty((M, k), np.int64)
    lib().isxo_topk_rows(_p(sim, F32P), C.c_int64(M), C.c_int64(N), k, C.c_int64(idx_base), _p(ts, F32P), _p(ti, I64P))
    return ts, ti


def average_precision(ranked, qlab, glab, kth=1):
    ranked = np.ascontiguousarray(ranked, np.int64); M, N = ranked.shape
    qlab = np.ascontiguousarray(qlab, np.int32); glab = np.ascontiguousarray(glab, np.int32)
    ap = np.empty((M,), np.float64)
    lib().isxo_average_precision(_p(ranked, I64P), C.c_int64(M), C.c_int64(N), _p(qlab, I32P), _p(glab, I32P), kth,
                                 _p(ap, F64P))
    return ap


def avg_precision_literal(sim_row, query_label, gallery_labels, kth=1, tensor_iteration=False):
    """utils/metrics.py:25-45 as the reference runs it: ONE query, a descending sort of its score row and a pure-Python walk over
    every rank (old_recall / old_precision trapezoid).  `sim_row`: 1-d torch tensor or numpy array; labels: Python list.  Used by
    tests (against isxo_average_precision) and timed by bench.py's cpu_baseline leg as the reference's mAP cost per query."""
    row = np.asarray(sim_row, dtype=np.float32)
    n_pos = sum(1 for l in gallery_labels if l == query_label) - (kth - 1)
    if n_pos <= 0:
        return None
    ranked = np.argsort(-row, kind="stable")              # canonical tie-break (score desc, index asc); the reference's sort is unspecified on ties
    old_recall, old_precision, ap = 0.0, 1.0, 0.0
    inter, j = 0, 0
    if tensor_iteration:                                  # as the reference walks it: element by element of a torch LongTensor (0-d tensors as indices)
        import torch
        ranked_iter = torch.from_numpy(ranked)
    else:
        ranked_iter = ranked.tolist()
    for n, k in enumerate(ranked_iter):
        if n + 1 < kth:
            continue
        if gallery_labels[k] == query_label:
            inter += 1
        recall = inter / float(n_pos)
        precision = inter / (j + 1.0)
        ap += (recall - old_recall) * ((old_precision + precision) / 2.0)
        old_recall, old_precision = recall, precision
        j += 1
    return ap


def mean_avg_precision(ap):
    """utils/metrics.py:48-55: plain sequential Python sum over the non-skipped queries."""
    vals = [float(a) for a in ap if not np.isnan(a)]
    return sum(vals) / float(len(vals))


def topk_merge(scores, idx):
    scores = _f32(scores); idx = np.ascontiguousarray(idx, np.int64); P, M, k = scores.shape
    os_ = np.empty((M, k), np.float32); oi = np.empty((M, k), np.int64)
    lib().isxo_topk_merge(_p(scores, F32P), _p(idx, I64P), P, C.c_int64(M), k, _p(os_, F32P), _p(oi, I64P))
    return os_, oi


def masked_sums(sim, qlab, glab):
    sim = _f32(sim); M, N = sim.shape
    qlab = np.ascontiguousarray(qlab, np.int32); glab = np.ascontiguousarray(glab, np.int32)
    sp = C.c_double(); sa = C.c_double()
    lib().isxo_masked_sums(_p(sim, F32P), C.c_int64(M), C.c_int64(N), _p(qlab, I32P), _p(glab, I32P),
                           C.byref(sp), C.byref(sa))
    return sp.value, sa.value


def precision1(top_idx, qlab, glab, kth=1):
    """utils/metrics.py:8-19 on a canonical top-k list: the hit is the kth ranked item."""
    hit = np.asarray(glab)[np.asarray(top_idx)[:, max(kth, 1) - 1]]
    correct = int((hit == np.asarray(qlab)).sum())
    return correct / float(len(qlab)), correct, len(qlab), hit


def mine_negatives(sim, labels, i1, i2, semi_hard):
    sim = _f32(sim); N = sim.shape[0]
    labels = np.ascontiguousarray(labels, np.int32)
    i1 = np.ascontiguousarray(i1, np.int64); i2 = np.ascontiguousarray(i2, np.int64)
    neg = np.empty_like(i1)
    lib().isxo_mine_negatives(_p(sim, F32P), C.c_int64(N), _p(labels, I32P), _p(i1, I64P), _p(i2, I64P), C.c_int64(len(i1)),
                              1 if semi_hard else 0, _p(neg, I64P))
    return neg


def triplet_loss(a, p, n, margin, normalized=True, size_average=True):
    a, p, n = _f32(a), _f32(p), _f32(n); B, D = a.shape
    rows = np.empty((B,), np.float32); ga, gp, gn = np.empty_like(a), np.empty_like(a), np.empty_like(a)
    lib().isxo_triplet_loss.restype = C.c_float
    loss = lib().isxo_triplet_loss(_p(a, F32P), _p(p, F32P), _p(n, F32P), C.c_int64(B), D, C.c_float(margin),
                                   1 if normalized else 0, 1 if size_average else 0, _p(rows, F32P), _p(ga, F32P), _p(gp, F32P), _p(gn, F32P))
    return float(loss), rows, ga, gp, gn


def dba(emb, labels, k=-1):
    """test/instance_avg.py:7-33 restated: (N,D) descriptors, (N) int labels -> (N,D)."""
    emb = _f32(emb); N, D = emb.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    out = np.empty_like(emb)
    lib().isxo_dba(_p(emb, F32P), C.c_int64(N), C.c_int64(D), _p(lab, I32P), int(k), _p(out, F32P))
    return out
