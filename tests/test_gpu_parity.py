"""HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs, and vs the
golden vectors produced by the reference.  Integer / index / rank outputs: bit-exact.
Cosine scores: bit-exact vs the oracle's fma chain, <= 1e-5 vs torch.mm goldens.
Other fp32 outputs: rtol 2e-6 (summation-order slack only)."""
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


@pytest.fixture(scope="module")
def ops():
    from isx import ops as o
    return o


# ------------------------------------------------------------------ descriptor head
@pytest.mark.parametrize("B,D", [(1, 1), (7, 37), (5, 64), (33, 256), (9, 464), (64, 2048), (3, 9216), (2, 100352), (3, 100353)])
def test_l2norm_rows(ops, B, D):
    rng = np.random.default_rng(B * 1000 + D)
    x = rng.standard_normal((B, D), dtype=np.float32)
    x[0] = 0.0
    y = host(ops.l2norm_rows(dev(x)))
    np.testing.assert_allclose(y, O.l2norm_rows(x), **TOL)
    assert (y[0] == 0).all()
    sh = rng.standard_normal((D,), dtype=np.float32) * 0.1
    y2 = host(ops.l2norm_shift_rows(dev(x), dev(sh)))
    np.testing.assert_allclose(y2, O.shift_rows(O.l2norm_rows(x), sh), rtol=2e-6, atol=1e-7)


def test_l2norm_golden(ops, golden):
    g = golden("l2norm_shift.npz")
    np.testing.assert_allclose(host(ops.l2norm_rows(dev(g["x"]))), g["y"], **TOL)
    np.testing.assert_allclose(host(ops.l2norm_rows(dev(g["x_wide"]))), g["y_wide"], **TOL)


@pytest.mark.parametrize("B,C,H,W", [(1, 4, 1, 1), (3, 16, 4, 4), (2, 24, 7, 7), (5, 256, 6, 6), (4, 2048, 7, 7), (2, 2048, 14, 14),
                                     (3, 30, 5, 3), (2, 2048, 3, 5), (1, 8, 40, 40), (2, 5000, 2, 2)])
def test_gap_l2(ops, B, C, H, W):
    rng = np.random.default_rng(C + H)
    f = np.maximum(rng.standard_normal((B, C, H, W), dtype=np.float32), 0)    # post-ReLU like
    y = host(ops.gap_l2(dev(f)))
    np.testing.assert_allclose(y, O.gap_l2(f), **TOL)


@pytest.mark.parametrize("B,C,H,W", [(3, 16, 4, 4), (4, 2048, 7, 7), (2, 256, 6, 6), (2, 4096, 3, 3), (3, 30, 5, 3), (2, 8200, 2, 2), (1, 2048, 14, 14)])
def test_gap_l2_channels_last(ops, B, C, H, W):
    rng = np.random.default_rng(C * 7 + H)
    f = np.maximum(rng.standard_normal((B, C, H, W), dtype=np.float32), 0)
    x = dev(f).to(memory_format=torch.channels_last)
    assert not x.is_contiguous()
    y = host(ops.gap_l2(x))
    np.testing.assert_allclose(y, O.gap_l2(f), **TOL)
    np.testing.assert_allclose(y, host(ops.gap_l2(dev(f))), **TOL)      # NHWC and NCHW kernels agree (same pooled values)


def test_gap_l2_golden(ops, golden):
    g = golden("gap_l2.npz")
    np.testing.assert_allclose(host(ops.gap_l2(dev(g["fmap"]))), g["desc"], **TOL)
    np.testing.assert_allclose(host(ops.gap_l2(dev(g["fmap7"]))), g["desc7"], **TOL)


@pytest.mark.parametrize("B,C,H,W,kh,kw", [(1, 16, 7, 5, 3, 3), (2, 2048, 14, 14, 7, 7), (1, 256, 13, 13, 6, 6), (1, 3, 7, 7, 7, 7),
                                           (1, 2, 120, 130, 7, 7)])
def test_boxpool_s1(ops, B, C, H, W, kh, kw):
    rng = np.random.default_rng(H * W)
    f = rng.standard_normal((B, C, H, W), dtype=np.float32)
    np.testing.assert_array_equal(host(ops.boxpool_s1(dev(f), kh, kw)), O.boxpool_s1(f, kh, kw))   # same summation order


def test_boxpool_golden(ops, golden):
    g = golden("classif_sub.npz")
    np.testing.assert_allclose(host(ops.boxpool_s1(dev(g["fmap_r"]), 3, 3)), g["pooled_r"], **TOL)


# ------------------------------------------------------------------ region path
def test_best_location(ops, golden):
    g = golden("best_location.npz")
    for t in list(range(4)) + ["_r"]:
        m = g["map%s" % t] if t != "_r" else g["map_r"]
        d, loc = ops.best_location_desc(dev(m[None]))
        want_loc = g["locs"][t] if t != "_r" else g["loc_r"]
        assert tuple(host(loc)[0]) == tuple(want_loc)
        np.testing.assert_allclose(host(d)[0], g["desc%s" % t] if t != "_r" else g["desc_r"], **TOL)
    rng = np.random.default_rng(1)
    cls = rng.standard_normal((6, 464, 8, 8), dtype=np.float32)
    cls[2, :, 5, 2] = cls[2, :, 1, 6]            # tie between two locations: smallest column wins
    cls[2, 7, 5, 2] = cls[2, 7, 1, 6] = 50.0
    d, loc = ops.best_location_desc(dev(cls))
    for b in range(6):
        od, ol = O.best_location_desc(cls[b])
        assert tuple(host(loc)[b]) == tuple(ol)
        np.testing.assert_allclose(host(d)[b], od, **TOL)
    assert tuple(host(loc)[2]) == (5, 2)


@pytest.mark.parametrize("K,Hp,Wp,k", [(9, 5, 3, 3), (9, 5, 3, 40), (464, 8, 8, 6), (17, 1, 1, 6), (5, 60, 60, 10)])
def test_region_topk_and_gather(ops, K, Hp, Wp, k):
    rng = np.random.default_rng(K + Hp)
    cls = rng.standard_normal((K, Hp, Wp), dtype=np.float32)
    if Hp * Wp > 4:
        cls[:, 1, 1] = cls[:, 0, 0]              # tie
    idx, sc = ops.region_topk(dev(cls), k)
    oi, osc = O.region_topk(cls, k)
    n = len(oi)
    np.testing.assert_array_equal(host(idx)[:n], oi)
    np.testing.assert_array_equal(host(sc)[:n], osc)
    assert (host(idx)[n:] == -1).all()
    C, fs = 12, 3
    fmap = rng.standard_normal((C, Hp + fs - 1, Wp + fs - 1), dtype=np.float32)
    sh = rng.standard_normal((C * fs * fs,), dtype=np.float32) * 0.05
    rows = host(ops.region_gather_l2(dev(fmap), fs, fs, idx, Wp, dev(sh)))
    np.testing.assert_allclose(rows[:n], O.region_gather_l2(fmap, fs, fs, oi, Wp, sh), rtol=2e-6, atol=1e-7)
    assert (rows[n:] == 0).all()


def test_region_batched(ops):
    """Batched region_topk / region_gather_l2 == the per-image calls == the oracle."""
    rng = np.random.default_rng(9)
    B, K, Hp, Wp, C, fs, k = 5, 17, 8, 6, 10, 3, 7
    cls = rng.standard_normal((B, K, Hp, Wp), dtype=np.float32)
    fmap = rng.standard_normal((B, C, Hp + fs - 1, Wp + fs - 1), dtype=np.float32)
    sh = rng.standard_normal((C * fs * fs,), dtype=np.float32) * 0.05
    idx, sc = ops.region_topk(dev(cls), k)
    rows = ops.region_gather_l2(dev(fmap), fs, fs, idx, Wp, dev(sh))
    assert idx.shape == (B, k) and rows.shape == (B, k, C * fs * fs)
    for b in range(B):
        oi, osc = O.region_topk(cls[b], k)
        np.testing.assert_array_equal(host(idx)[b], oi)
        np.testing.assert_array_equal(host(sc)[b], osc)
        np.testing.assert_allclose(host(rows)[b], O.region_gather_l2(fmap[b], fs, fs, oi, Wp, sh), rtol=2e-6, atol=1e-7)


def test_region_descriptor_golden(ops, golden):
    g = golden("region_desc.npz")
    for tag, k in (("k3", 3), ("k40", 40)):
        cls = g["cls_" + tag][0]
        idx, _ = ops.region_topk(dev(cls), k)
        n = min(k, cls.shape[1] * cls.shape[2])
        np.testing.assert_array_equal(host(idx)[:n], g["idx_" + tag])
        rows = ops.region_gather_l2(dev(g["fmap_" + tag][0]), 3, 3, idx[:n], cls.shape[2], dev(g["shift_" + tag]))
        acc = (rows @ dev(g["w_" + tag]).t() + dev(g["b_" + tag])).sum(0, keepdim=True)
        np.testing.assert_allclose(host(ops.l2norm_rows(acc)), g["desc_" + tag], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ retrieval
def unit(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    return O.l2norm_rows(x)


@pytest.mark.parametrize("M,N,D", [(1, 1, 1), (3, 5, 2), (100, 100, 9216), (7, 300, 464), (50, 1000, 311), (33, 129, 17),
                                   (128, 128, 32), (130, 257, 2048), (256, 1000, 2048), (5, 2000, 36), (64, 64, 31)])
def test_cosine_sim_bitexact(ops, M, N, D):
    rng = np.random.default_rng(M * N + D)
    Q, G = unit(rng, M, D), unit(rng, N, D)
    sim = host(ops.cosine_sim(dev(Q), dev(G)))
    np.testing.assert_array_equal(sim, O.cosine_sim(Q, G))        # MFMA fp32 == k-ordered fma chain
    ref = (Q.astype(np.float64) @ G.astype(np.float64).T)
    assert np.abs(sim - ref).max() <= 1e-5


def test_cosine_sim_configs2_size_sampled_against_the_oracle(ops):
    """BASELINE configs[2]'s retrieval leg at ITS size -- 1 000 queries x 100 000 gallery rows of 464-d class-score descriptors (the clipped k-tail
    of csrc/gemm_tile.hpp load_tile: D a multiple of 4 but not of the k-tile; 128x128 tiles for the whole rounds + a second launch of 64x64 tiles
    for the remaining columns): 24 query rows x all columns and all rows x 64 columns spread over both launches, bit for bit against the oracle."""
    rng = np.random.default_rng(464)
    Q, G = unit(rng, 1000, 464), unit(rng, 100000, 464)
    sim = host(ops.cosine_sim(dev(Q), dev(G)))
    rows = np.r_[0:8, 500:508, 992:1000]
    np.testing.assert_array_equal(sim[rows], O.cosine_sim(Q[rows], G))
    cols = np.r_[0:16, 50000:50016, 98290:98306, 99984:100000]
    np.testing.assert_array_equal(sim[:, cols], O.cosine_sim(Q, G[cols]))


def test_cosine_sim_mfma_layout(ops):
    # A = I against an ASYMMETRIC B catches swapped row/col maps (guide section 3)
    n = 128
    Q = np.eye(n, dtype=np.float32)
    G = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) - 100.0
    np.testing.assert_array_equal(host(ops.cosine_sim(dev(Q), dev(G))), G.T)
    np.testing.assert_array_equal(host(ops.cosine_sim(dev(G), dev(Q))), G)


def test_cosine_golden(ops, golden):
    g = golden("synthetic_retrieval.npz")
    for n in (100, 1000):
        t = "_n%d" % n
        sim = ops.cosine_sim(dev(g["Q" + t]), dev(g["G" + t]))
        assert np.abs(host(sim) - g["sim" + t]).max() <= 1e-5      # vs torch.mm fp32 of the reference run
        ranked = ops.rank_full(sim)
        ap = host(ops.average_precision(ranked, dev(g["qlab" + t]), dev(g["glab" + t])))
        assert abs(O.mean_avg_precision(ap) - float(g["map" + t])) <= 1e-4
        # on the reference's own score matrix the whole metric chain is bit-exact
        ranked = ops.rank_full(dev(g["sim" + t]))
        ap = host(ops.average_precision(ranked, dev(g["qlab" + t]), dev(g["glab" + t])))
        assert O.mean_avg_precision(ap) == float(g["map" + t])
        ts, ti = ops.topk_rows(dev(g["sim" + t]), 1)
        assert O.precision1(host(ti), g["qlab" + t], g["glab" + t])[:3] == tuple(g["p1" + t])


@pytest.mark.parametrize("M,N,D,k", [(1, 1, 8, 1), (4, 50, 16, 100), (37, 1000, 64, 10), (100, 5000, 128, 100), (16, 20000, 32, 1024),
                                     (300, 3000, 2048, 100)])
def test_cosine_topk(ops, M, N, D, k):
    rng = np.random.default_rng(N + k)
    Q, G = unit(rng, M, D), unit(rng, N, D)
    if N > 40:
        G[N // 2] = G[3]                          # duplicated gallery rows -> exact score ties
        G[N - 1] = G[3]
    os_, oi = O.cosine_topk(Q, G, k, idx_base=11)
    ts, ti = ops.cosine_topk(dev(Q), dev(G), k, idx_base=11)
    np.testing.assert_array_equal(host(ti), oi)
    np.testing.assert_array_equal(host(ts), os_)
    # a tiny workspace forces many column chunks + carry merging: same answer
    a256 = lambda v: (v + 255) // 256 * 256
    small = torch.empty((a256(M * k * 8) + a256(M * 4) + a256(M * 8 * 4) + M * 256 * 4,), dtype=torch.uint8, device="cuda")
    ts2, ti2 = ops.cosine_topk(dev(Q), dev(G), k, idx_base=11, ws=small)
    np.testing.assert_array_equal(host(ti2), oi)
    np.testing.assert_array_equal(host(ts2), os_)


@pytest.mark.parametrize("k", [1, 100, 256, 300])
@pytest.mark.parametrize("order", ["random", "ascending", "descending"])
def test_cosine_topk_filtered_chunks(ops, k, order):
    """N > 8192 columns: bootstrap chunk + FILTERING GEMM chunks (k <= 256) or the materialised path
    (k = 300).  'ascending' is the adversarial order: every later column beats the running threshold
    of query 0, so every 32-column group qualifies and the filter degenerates to full materialisation."""
    rng = np.random.default_rng(k)
    M, N, D = 37, 40000, 48
    Q, G = unit(rng, M, D), unit(rng, N, D)
    if order != "random":
        s0 = G @ Q[0]
        G = G[np.argsort(s0 if order == "ascending" else -s0)]
    G[9000] = G[17]; G[39999] = G[17]; G[8191] = G[8192]          # ties across chunk boundaries
    want_s, want_i = O.cosine_topk(Q, G, k, idx_base=3)
    ts, ti = ops.cosine_topk(dev(Q), dev(G), k, idx_base=3)
    np.testing.assert_array_equal(host(ti), want_i)
    np.testing.assert_array_equal(host(ts), want_s)
    a256 = lambda v: (v + 255) // 256 * 256
    for nc in (4096, 1000):                                           # several filtered chunks / odd chunk width
        ws = torch.empty((a256(M * k * 8) + a256(M * 4) + a256(M * ((nc + 31) // 32)) + M * nc * 4,), dtype=torch.uint8, device="cuda")
        ts, ti = ops.cosine_topk(dev(Q), dev(G), k, idx_base=3, ws=ws)
        np.testing.assert_array_equal(host(ti), want_i)
        np.testing.assert_array_equal(host(ts), want_s)


@pytest.mark.parametrize("M,N,k", [(100, 100000, 100), (37, 65536, 1), (5, 16384 * 3, 256), (1000, 20000, 10), (3, 100003, 7)])
def test_topk_rows_few_rows_by_segments(ops, M, N, k):
    """Few query rows against a long gallery (configs[1] / [2] evaluation: 1 000 x 100 000): ops.topk_rows selects per row SEGMENT (the (M, N)
    matrix viewed as (M S, N / S)) and merges the segment lists with the canonical comparator -- the same lists, bit for bit, as the one-workgroup-
    per-row launch, ties included."""
    from isx._lib import check, lib
    g = torch.Generator(device="cuda").manual_seed(M + k)
    sim = torch.randn(M, N, device="cuda", generator=g)
    sim[:, N // 3] = sim[:, 5]                       # ties across segments: the smaller column index ranks first
    sim[0, :] = 0.25                                 # a whole row of equal scores
    s1, i1 = ops.topk_rows(sim, k, idx_base=17)
    s0 = torch.empty((M, k), device="cuda"); i0 = torch.empty((M, k), device="cuda", dtype=torch.int64)
    check(lib().isx_topk_rows(sim.data_ptr(), M, N, k, 17, s0.data_ptr(), i0.data_ptr(), torch.cuda.current_stream().cuda_stream), "isx_topk_rows")
    assert torch.equal(i1, i0) and torch.equal(s1.view(torch.int32), s0.view(torch.int32))
    assert torch.equal(i1[0], torch.arange(k, device="cuda") + 17)
    assert torch.equal(s1, torch.topk(sim, k, dim=1).values)
    assert (ops._row_segments(M, N, k) > 0) == (N % 2 == 0 or N % 3 == 0 or N % 5 == 0 or N % 7 == 0)


def test_topk_rows_adversarial(ops):
    # ascending scores: every element beats the running threshold (worst case for the filter)
    M, N, k = 3, 10000, 100
    sim = np.tile(np.linspace(-1, 1, N, dtype=np.float32), (M, 1))
    sim[1] = sim[1, ::-1]
    sim[2] = 0.25                                   # all tied: index order decides
    ts, ti = ops.topk_rows(dev(sim), k, idx_base=0)
    os_, oi = O.topk_rows(sim, k)
    np.testing.assert_array_equal(host(ti), oi)
    np.testing.assert_array_equal(host(ts), os_)
    assert (host(ti)[2] == np.arange(k)).all()


@pytest.mark.parametrize("N,k", [(300, 64), (20000, 64), (65536, 100), (100003, 7)])
def test_topk_rows_non_finite_and_signed_zero_scores(ops, N, k):
    """NaNs of both signs, infinities, signed zeros and denormals among the scores: the selection kernel's cheap reject (a float compare
    against the score of the threshold key) must hand every such value to the canonical key compare, whose order is the oracle's
    (+NaN above +inf, -NaN below -inf, -0 folded onto +0, index ascending among equals)."""
    rng = np.random.default_rng(N + k)
    M = 6
    sim = rng.standard_normal((M, N)).astype(np.float32)
    special = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32)
    special.view(np.uint32)[1] |= 0x80000000                                # make the second one a NEGATIVE NaN whatever numpy did with the sign
    for r in range(M):
        pos = rng.choice(N, size=min(N // 2, 40 * (r + 1)), replace=False)
        sim[r, pos] = special[rng.integers(0, len(special), pos.size)]
    sim[4] = np.where(rng.random(N) < 0.5, np.float32(0.0), np.float32(-0.0))      # a row of signed zeros only: index order decides
    sim[5, : N // 2] = np.nan                                                # more NaNs than k: the list is all NaN, lowest indices first
    ts, ti = ops.topk_rows(dev(sim), k, idx_base=3)
    os_, oi = O.topk_rows(sim, k, idx_base=3)
    np.testing.assert_array_equal(host(ti), oi)
    fold = lambda a: np.where(a == 0, np.float32(0.0), a).view(np.uint32)           # a score of -0 leaves the key domain as +0 (equal as numbers)
    np.testing.assert_array_equal(fold(host(ts)), fold(os_))                        # everything else bit for bit, NaN payloads included
    assert (host(ti)[4] == np.arange(k) + 3).all()
    assert np.isnan(host(ts)[5]).all() and (host(ti)[5] == np.arange(k) + 3).all()


@pytest.mark.parametrize("M,N", [(1, 1), (5, 24), (12, 40), (3, 4096), (4, 4097), (6, 10000), (2, 70000)])
def test_rank_full_and_ap(ops, M, N):
    rng = np.random.default_rng(N)
    sim = (np.round(rng.random((M, N)) * 50) / 50).astype(np.float32) if N < 100 else rng.standard_normal((M, N), dtype=np.float32)
    if N > 10:
        sim[:, 7] = sim[:, 2]
    ranked = ops.rank_full(dev(sim))
    np.testing.assert_array_equal(host(ranked), O.rank_full(sim))
    L = max(1, N // 10)
    gl = (np.arange(N) % L).astype(np.int32)
    ql = (np.arange(M) % (L + 1)).astype(np.int32)     # label L absent from the gallery -> NaN (skipped)
    for kth in (1, 2, 3):
        ap = host(ops.average_precision(ranked, dev(ql), dev(gl), kth))
        want = O.average_precision(O.rank_full(sim), ql, gl, kth)
        np.testing.assert_array_equal(np.isnan(ap), np.isnan(want))
        np.testing.assert_array_equal(ap[~np.isnan(ap)], want[~np.isnan(want)])     # float64, bit-exact


@pytest.mark.parametrize("M,N,L", [(12, 40, 6), (9, 1000, 100), (5, 5000, 50), (4, 3000, 3), (20, 10000, 1000), (7, 40001, 50), (12, 100000, 5000), (3, 16384, 600), (6, 32768, 40), (5, 65536, 8192), (4, 65537, 6000), (9, 10000, 2000), (5, 40000, 200), (3, 33001, 150), (4, 2048, 8), (3, 6000, 3), (4, 20480, 20), (2, 100000, 110)])
def test_average_precision_sim_equals_sorted_path(ops, M, N, L):
    """Sort-free AP == rank_full + average_precision == oracle, bit for bit (incl. kth > 1, skipped
    queries, tied scores, rows with up to 1024 positives counted in chunks of 32 -- 1000, 1024, 909 per query among the cases -- and rows with
    more (2000) that take the sorted fallback)."""
    rng = np.random.default_rng(N + L)
    sim = (np.round(rng.random((M, N)) * 200) / 200).astype(np.float32)        # many exact ties
    gl = (np.arange(N) % L).astype(np.int32)
    ql = (np.arange(M) % (L + 1)).astype(np.int32)
    ranked = O.rank_full(sim)
    for kth in (1, 2, 4):
        want = O.average_precision(ranked, ql, gl, kth)
        got = host(ops.average_precision_sim(dev(sim), dev(ql), dev(gl), kth))
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_array_equal(got[~np.isnan(got)], want[~np.isnan(want)])


def test_average_precision_sim_with_non_finite_scores(ops):
    """NaN / inf / signed-zero scores: the sort-free AP ranks by the canonical keys only, like the full sort."""
    rng = np.random.default_rng(99)
    M, N, L = 8, 30000, 60
    sim = rng.standard_normal((M, N)).astype(np.float32)
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45], np.float32)
    for r in range(M):
        pos = rng.choice(N, size=500 * (r + 1), replace=False)
        sim[r, pos] = special[rng.integers(0, len(special), pos.size)]
    gl = (np.arange(N) % L).astype(np.int32)
    ql = (np.arange(M) % L).astype(np.int32)
    want = O.average_precision(O.rank_full(sim), ql, gl, 1)
    got = host(ops.average_precision_sim(dev(sim), dev(ql), dev(gl), 1))
    np.testing.assert_array_equal(got, want)


def test_metrics_golden(ops, golden):
    g = golden("metrics.npz")
    sim, ql, gl = g["sim"], g["qlab"], g["glab"]
    ranked = ops.rank_full(dev(sim))
    ts, ti = ops.topk_rows(dev(sim), 3)
    for kth in (1, 2, 3):
        ap = host(ops.average_precision(ranked, dev(ql), dev(gl), kth))
        ref = g["ap_kth%d" % kth]
        np.testing.assert_array_equal(np.isnan(ap), np.isnan(ref))
        np.testing.assert_array_equal(ap[~np.isnan(ap)], ref[~np.isnan(ref)])
        assert O.mean_avg_precision(ap) == float(g["map_kth%d" % kth])
        p1, c, t, hit = O.precision1(host(ti), ql, gl, kth)
        assert (p1, c, t) == tuple(g["p1_kth%d" % kth])
        np.testing.assert_array_equal(host(ts)[:, kth - 1], g["p1_maxsim_kth%d" % kth])
    apt = host(ops.average_precision(ops.rank_full(dev(g["tie_sim"])), dev(g["tie_qlab"]), dev(g["tie_glab"])))
    np.testing.assert_array_equal(apt, g["tie_ap"])


def test_masked_sums(ops):
    rng = np.random.default_rng(5)
    sim = rng.standard_normal((20, 777), dtype=np.float32)
    ql = (np.arange(20) % 5).astype(np.int32)
    gl = (np.arange(777) % 5).astype(np.int32)
    out = host(ops.masked_sums(dev(sim), dev(ql), dev(gl)))
    sp, sa = O.masked_sums(sim, ql, gl)
    assert abs(out[:, 0].sum() - sp) < 1e-9 and abs(out[:, 1].sum() - sa) < 1e-9


@pytest.mark.parametrize("P,k", [(2, 10), (8, 100), (8, 512), (3, 1)])
def test_topk_merge(ops, P, k):
    rng = np.random.default_rng(P * k)
    M, N, D = 9, 4000, 32
    Q, G = unit(rng, M, D), unit(rng, N, D)
    G[100] = G[3900]                               # cross-shard tie
    bounds = np.linspace(0, N, P + 1).astype(int)
    parts = [O.cosine_topk(Q, G[a:b], k, idx_base=a) for a, b in zip(bounds[:-1], bounds[1:])]
    S, I = np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts])
    ms, mi = ops.topk_merge(dev(S), dev(I))
    fs, fi = O.cosine_topk(Q, G, k)
    np.testing.assert_array_equal(host(mi), fi)
    np.testing.assert_array_equal(host(ms), fs)


def test_native_rccl_allgather_single_rank(ops):
    """libisx's own RCCL communicator (dlopen'd librccl): with one rank the all-gather is the identity;
    exercises unique-id creation, ncclCommInitRank and the grouped ncclAllGather pair."""
    uid = ops.comm_unique_id()
    assert len(uid) == 128
    comm = ops.comm_init_rank(1, 0, uid)
    s = torch.randn(50, 10, device="cuda")
    i = torch.randint(0, 1000, (50, 10), device="cuda")
    all_s, all_i = ops.shard_topk_allgather(comm, 1, s, i)
    torch.cuda.synchronize()
    assert all_s.shape == (1, 50, 10) and torch.equal(all_s[0], s) and torch.equal(all_i[0], i)
    from isx.retrieval import NativeComm, ShardedGallery
    nc = NativeComm()
    g = ShardedGallery(ops.l2norm_rows(torch.randn(300, 64, device="cuda")), 0, native_comm=nc)
    q = ops.l2norm_rows(torch.randn(5, 64, device="cuda"))
    ts, ti = g.search(q, 7)
    want = O.cosine_topk(host(q), host(g.shard), 7)
    np.testing.assert_array_equal(host(ti), want[1])
    nc.close()
    ops.comm_destroy(comm)


# ------------------------------------------------------------------ full-size properties (BASELINE sizes)
def test_full_size_properties(ops):
    """1k x 100k x 2048 (config 3 scale) and a 10k-query slab: size-independent checks --
    sortedness, idempotence, shard-merge invariance, sampled rows against the oracle."""
    g = torch.Generator(device="cuda").manual_seed(0)
    M, N, D, k = 1000, 100000, 2048, 100
    G = torch.randn(N, D, device="cuda", generator=g)
    Q = torch.randn(M, D, device="cuda", generator=g)
    G, Q = ops.l2norm_rows(G), ops.l2norm_rows(Q)
    assert torch.allclose((G * G).sum(1), torch.ones(N, device="cuda"), atol=1e-5)
    ts, ti = ops.cosine_topk(Q, G, k)
    assert bool((ts[:, :-1] >= ts[:, 1:]).all())                                   # sorted
    tie = ts[:, :-1] == ts[:, 1:]
    assert bool((ti[:, :-1][tie] < ti[:, 1:][tie]).all())                          # ties by ascending index
    assert int(ti.min()) >= 0 and int(ti.max()) < N
    # idempotence: scores recomputed from the returned indices reproduce the list
    rows = [0, 1, 499, 999]
    sub = ops.cosine_sim(Q[rows], G)
    s2, i2 = ops.topk_rows(sub, k)
    assert torch.equal(i2, ti[rows]) and torch.equal(s2, ts[rows])
    # oracle on sampled rows (full 2048-d dot products on the CPU)
    os_, oi = O.cosine_topk(host(Q[rows]), host(G), k)
    np.testing.assert_array_equal(host(ti[rows]), oi)
    np.testing.assert_array_equal(host(ts[rows]), os_)
    # shard invariance: 8 row shards + merge == unsharded
    P = 8
    parts = [ops.cosine_topk(Q, G[p * N // P:(p + 1) * N // P], k, idx_base=p * N // P) for p in range(P)]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, ti) and torch.equal(ms, ts)
    # self-retrieval: each gallery row's best match is itself with score ~1
    s1, i1 = ops.cosine_topk(G[:2000], G, 1)
    assert torch.equal(i1[:, 0], torch.arange(2000, device="cuda"))
    assert bool(((s1[:, 0] - 1).abs() < 1e-5).all())


# ------------------------------------------------------------------ 1x1 convolution of the trunk (fused epilogue)
@pytest.mark.parametrize("M,Cin,Cout", [(1, 4, 4), (50, 64, 256), (300, 256, 64), (777, 512, 128), (130, 2048, 512), (64, 100, 36), (257, 24, 130)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (True, False)])
def test_conv1x1_nhwc(ops, M, Cin, Cout, res, relu):
    from isx._lib import lib
    rng = np.random.default_rng(M + Cin)
    x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, Cin), dtype=np.float32) * np.float32(Cin ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    r = rng.standard_normal((M, Cout), dtype=np.float32) if res else None
    want = O.conv1x1_nhwc(x, w, b, r, relu)
    # as a (1, Cin, M, 1) channels-last image
    xt = dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2)
    rt = dev(r).view(1, M, 1, Cout).permute(0, 3, 1, 2) if res else None
    y = ops.conv1x1_nhwc(xt, dev(w), dev(b), rt, relu)
    got = host(y.permute(0, 2, 3, 1).reshape(M, Cout))
    np.testing.assert_array_equal(got, want)                      # MFMA fp32 == fma chain: bit-exact
    # and against torch's convolution (different summation order): fp32 tolerance
    ref = torch.nn.functional.conv2d(xt, dev(w).view(Cout, Cin, 1, 1), dev(b))
    if res:
        ref = ref + rt
    if relu:
        ref = torch.relu(ref)
    np.testing.assert_allclose(got, host(ref.permute(0, 2, 3, 1).reshape(M, Cout)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,Cin,Cout", [(16384, 64, 256), (16384 + 77, 64, 256), (40000, 64, 64), (16384 + 5, 64, 64), (70001, 64, 256)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (True, False)])
def test_conv1x1_stream_equals_general(ops, M, Cin, Cout, res, relu):
    """The Cin = 64 layers (64 -> 64 / 256) with >= 16384 pixels take the streaming kernel (csrc/stream1x1.hip: persistent
    workgroups, weights in registers, LDS-DMA pixel ring, buffer-instruction epilogue): same bits as the tiled GEMM path (debug
    cfg 9) and as the oracle's fma chain, including ragged last tiles and more tiles than workgroups (70001 pixels = 1094 tiles of
    64 on 256 workgroups)."""
    from isx._lib import lib
    rng = np.random.default_rng(M + Cout + Cin)
    x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, Cin), dtype=np.float32) * np.float32(Cin ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    r = rng.standard_normal((M, Cout), dtype=np.float32) if res else None
    xt = dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2)
    rt = dev(r).view(1, M, 1, Cout).permute(0, 3, 1, 2) if res else None
    set_cfg = lib().isx_debug_set_conv_cfg
    try:
        set_cfg(9)
        general = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), rt, relu).permute(0, 2, 3, 1).reshape(M, Cout))
        set_cfg(-1)
        stream = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), rt, relu).permute(0, 2, 3, 1).reshape(M, Cout))
    finally:
        set_cfg(-1)
    np.testing.assert_array_equal(stream.view(np.int32), general.view(np.int32))
    rows = np.r_[0:200, M // 2:M // 2 + 200, M - 200:M]                   # oracle on three row windows (first, middle, ragged end)
    want = O.conv1x1_nhwc(x[rows], w, b, r[rows] if res else None, relu)
    np.testing.assert_array_equal(stream[rows], want)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(1, 4, 4, 32, 32, 1), (2, 7, 7, 64, 64, 1), (3, 8, 6, 64, 128, 2), (2, 14, 14, 128, 96, 1),
                                                   (1, 9, 11, 256, 130, 2), (5, 5, 5, 32, 64, 1), (2, 1, 1, 32, 32, 1), (1, 3, 2, 512, 512, 2)])
@pytest.mark.parametrize("res,relu", [(False, True), (True, True), (False, False)])
def test_conv3x3_nhwc(ops, B, H, W, Cin, Cout, stride, res, relu):
    from isx._lib import lib
    rng = np.random.default_rng(H * 100 + Cin + stride)
    x = np.maximum(rng.standard_normal((B, H, W, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * np.float32((9 * Cin) ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = rng.standard_normal((B, Ho, Wo, Cout), dtype=np.float32) if res else None
    want = O.conv3x3_nhwc(x, w, b, stride, r, relu)
    xt = dev(x).permute(0, 3, 1, 2)                        # channels-last view of (B,Cin,H,W)
    rt = dev(r).permute(0, 3, 1, 2) if res else None
    set_cfg = lib().isx_debug_set_conv_cfg
    try:
        for cfg in (-1, 0, 2, 3):                          # every instantiated tile shape: same bits
            set_cfg(cfg)
            y = ops.conv3x3_nhwc(xt, dev(w), dev(b), stride, rt, relu)
            got = host(y.permute(0, 2, 3, 1))
            np.testing.assert_array_equal(got, want)
    finally:
        set_cfg(-1)
    ref = torch.nn.functional.conv2d(xt, dev(w).permute(0, 3, 1, 2), dev(b), stride=stride, padding=1)
    if res:
        ref = ref + rt
    if relu:
        ref = torch.relu(ref)
    np.testing.assert_allclose(got, host(ref.permute(0, 2, 3, 1)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,W,Cin,stride,res,relu", [(2, 7, 7, 64, 1, True, True), (1, 9, 11, 32, 2, False, True), (3, 56, 56, 64, 1, True, True),
                                                       (1, 5, 5, 128, 1, True, False), (40, 14, 14, 64, 1, True, True), (1, 1, 1, 32, 1, False, True)])
def test_conv3x3_expand(ops, B, H, W, Cin, stride, res, relu):
    """conv2 (3x3 -> 64 channels, ReLU) + conv3 (1x1 -> 256, + residual, ReLU) of a Bottleneck as ONE kernel (conv3x3_expand_kernel: the
    mid activation goes registers -> LDS -> second MFMA loop): same bits as the two separate libisx kernels and as the oracle's
    conv3x3_nhwc followed by conv1x1_nhwc."""
    rng = np.random.default_rng(B * 100 + H + Cin)
    x = np.maximum(rng.standard_normal((B, H, W, Cin), dtype=np.float32), 0)
    w2 = rng.standard_normal((64, 3, 3, Cin), dtype=np.float32) * np.float32((9 * Cin) ** -0.5)
    b2 = rng.standard_normal(64, dtype=np.float32)
    w3 = rng.standard_normal((256, 64), dtype=np.float32) * np.float32(0.125)
    b3 = rng.standard_normal(256, dtype=np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = rng.standard_normal((B, Ho, Wo, 256), dtype=np.float32) if res else None
    xt = dev(x).permute(0, 3, 1, 2)
    rt = dev(r).permute(0, 3, 1, 2) if res else None
    got = host(ops.conv3x3_expand_nhwc(xt, dev(w2), dev(b2), stride, dev(np.ascontiguousarray(w3.T)), dev(b3), rt, relu).permute(0, 2, 3, 1))
    mid = ops.conv3x3_nhwc(xt, dev(w2), dev(b2), stride, None, True)
    two = host(ops.conv1x1_nhwc(mid, dev(w3), dev(b3), rt, relu).permute(0, 2, 3, 1))
    np.testing.assert_array_equal(got.view(np.int32), two.view(np.int32))
    want_mid = O.conv3x3_nhwc(x, w2, b2, stride, None, True)
    want = O.conv1x1_nhwc(want_mid.reshape(-1, 64), w3, b3, r.reshape(-1, 256) if res else None, relu).reshape(B, Ho, Wo, 256)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("B,H,W,Cin,relu", [(2, 7, 7, 64, True), (1, 9, 11, 32, True), (3, 56, 56, 64, True), (1, 5, 5, 128, False), (1, 1, 1, 32, True)])
def test_conv3x3_expand_dual(ops, B, H, W, Cin, relu):
    """First block of the 64-channel stage as ONE kernel (conv3x3_expand_kernel<.., DUAL>): conv2 + ReLU, then [W3 | Wd] . [mid ; x2] + bias:
    same bits as isx_conv3x3_nhwc followed by isx_conv1x1_dual_nhwc and as the oracle's composition."""
    rng = np.random.default_rng(B * 10 + H + Cin)
    t = np.maximum(rng.standard_normal((B, H, W, Cin), dtype=np.float32), 0)
    x2 = np.maximum(rng.standard_normal((B, H, W, 64), dtype=np.float32), 0)
    w2 = rng.standard_normal((64, 3, 3, Cin), dtype=np.float32) * np.float32((9 * Cin) ** -0.5)
    b2 = rng.standard_normal(64, dtype=np.float32)
    wc = rng.standard_normal((256, 128), dtype=np.float32) * np.float32(128 ** -0.5)
    b = rng.standard_normal(256, dtype=np.float32)
    tt, xt = dev(t).permute(0, 3, 1, 2), dev(x2).permute(0, 3, 1, 2)
    got = host(ops.conv3x3_expand_dual_nhwc(tt, dev(w2), dev(b2), xt, dev(np.ascontiguousarray(wc.T)), dev(b), relu).permute(0, 2, 3, 1))
    mid = ops.conv3x3_nhwc(tt, dev(w2), dev(b2), 1, None, True)
    two = host(ops.conv1x1_dual_nhwc(mid, xt, dev(wc), dev(b), 1, relu).permute(0, 2, 3, 1))
    np.testing.assert_array_equal(got.view(np.int32), two.view(np.int32))
    want = O.conv1x1_dual_nhwc(O.conv3x3_nhwc(t, w2, b2, 1, None, True), x2, wc, b, 1, relu)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("Cout,stride,res", [(128, 1, False), (256, 1, True), (128, 2, True)])
def test_conv3x3_tail_split(ops, Cout, stride, res):
    """A launch a little above a whole number of rounds of 128x128 tiles (here 1047 row tiles: one round of 1024 + 23) runs its last rows as
    64x64 tiles in the same grid (conv3x3_tail_kernel).  Same bits as the plain 128x128 launch (debug cfg 7), as 64x64 tiles everywhere
    (cfg 3), and as the oracle on the first and the last image (the last one lies in the 64x64 region)."""
    from isx._lib import lib
    B, Ho, Cin = 9, 122, 32
    H = Ho * stride
    rng = np.random.default_rng(Cout + stride)
    x = np.maximum(rng.standard_normal((B, H, H, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * np.float32((9 * Cin) ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    r = rng.standard_normal((B, Ho, Ho, Cout), dtype=np.float32) if res else None
    xt, rt = dev(x).permute(0, 3, 1, 2), (dev(r).permute(0, 3, 1, 2) if res else None)
    set_cfg = lib().isx_debug_set_conv_cfg
    out = {}
    try:
        for cfg in (0, 7, 3):
            set_cfg(cfg)
            out[cfg] = host(ops.conv3x3_nhwc(xt, dev(w), dev(b), stride, rt, True).permute(0, 2, 3, 1))
    finally:
        set_cfg(-1)
    np.testing.assert_array_equal(out[0].view(np.int32), out[7].view(np.int32))
    np.testing.assert_array_equal(out[0].view(np.int32), out[3].view(np.int32))
    for i in (0, B - 1):
        want = O.conv3x3_nhwc(x[i:i + 1], w, b, stride, r[i:i + 1] if res else None, True)
        np.testing.assert_array_equal(out[0][i:i + 1], want)


@pytest.mark.parametrize("B,H,W", [(1, 224, 224), (3, 224, 224), (2, 50, 36), (1, 8, 4), (5, 100, 224), (2, 230, 200), (1, 1, 4), (300, 32, 32),
                                   (1, 17, 12), (1, 223, 224), (2, 300, 4), (257, 16, 8), (4, 129, 220), (100, 64, 64), (37, 100, 60)])
def test_stem7x7_pool(ops, B, H, W):
    """The fused stem (csrc/stem.hip: conv 7x7/2 + bias + ReLU + maxpool 3/2/1, one kernel, convolution output never stored) against
    the oracle's fma chain bit for bit: full 224 x 224 images (one workgroup per image and, for few images, bands of rows with a
    recomputed carry row), ragged heights (last step partly outside), narrow images (masked columns / partial DMA rows), more images
    than CUs; and within 1e-4 of torch's conv2d + relu + max_pool2d on the same device."""
    rng = np.random.default_rng(H * 1000 + W + B)
    x = rng.standard_normal((B, H, W, 3), dtype=np.float32)
    w = rng.standard_normal((64, 7, 7, 3), dtype=np.float32) * np.float32(147 ** -0.5)
    b = rng.standard_normal(64, dtype=np.float32)
    xt = dev(x).permute(0, 3, 1, 2)
    got = host(ops.stem7x7_pool(xt, dev(w), dev(b)).permute(0, 2, 3, 1))
    want = O.stem7x7_pool_nhwc(x, w, b)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    ref = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(xt, dev(w).permute(0, 3, 1, 2), dev(b), stride=2, padding=3)), 3, 2, 1)
    np.testing.assert_allclose(got, host(ref.permute(0, 2, 3, 1)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,W,C", [(1, 1, 1, 4), (2, 7, 9, 8), (3, 112, 112, 64), (2, 12, 13, 36), (1, 2, 2, 4)])
def test_bias_relu_maxpool(ops, B, H, W, C):
    rng = np.random.default_rng(H + C)
    y = rng.standard_normal((B, H, W, C), dtype=np.float32)
    b = rng.standard_normal(C, dtype=np.float32)
    got = host(ops.bias_relu_maxpool(dev(y).permute(0, 3, 1, 2), dev(b)).permute(0, 2, 3, 1))
    np.testing.assert_array_equal(got, O.bias_relu_maxpool_nhwc(y, b))
    ref = torch.nn.functional.max_pool2d(torch.relu(dev(y).permute(0, 3, 1, 2) + dev(b).view(1, -1, 1, 1)), 3, 2, 1)
    np.testing.assert_array_equal(got, host(ref.permute(0, 2, 3, 1)))


@pytest.mark.parametrize("B,H,W", [(1, 1, 1), (2, 5, 7), (3, 224, 224), (2, 13, 3)])
@pytest.mark.parametrize("cl", [True, False])
def test_images_u8_to_f32(ops, B, H, W, cl):
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    img[0, 0, 0] = (0, 255, 128)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    got = host(ops.images_u8_to_f32(dev(img), mean, std, channels_last=cl))
    np.testing.assert_array_equal(got, O.images_u8_to_f32(img, mean, std))
    # torch's ToTensor + Normalize arithmetic
    x = torch.from_numpy(img).permute(0, 3, 1, 2).float().div(255.0)
    ref = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    np.testing.assert_array_equal(got, ref.numpy())


def test_stage_batch_raw_ingest(ops):
    """uint8 datasets: stage_batch normalises on the GPU exactly like the CPU ToTensor + Normalize path."""
    from train import _common as TC
    rng = np.random.default_rng(3)
    raw = [(torch.from_numpy(rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)), i % 3, "im%d" % i) for i in range(5)]
    TC.RAW_INGEST["mean"], TC.RAW_INGEST["std"] = [0.4, 0.5, 0.6], [0.2, 0.25, 0.3]
    try:
        g = TC.stage_batch(raw, None, 0)
        c = TC.stage_batch(raw, None, -1)
    finally:
        TC.RAW_INGEST["mean"] = TC.RAW_INGEST["std"] = None
    assert g.is_cuda and g.shape == (5, 3, 32, 48) and g.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_array_equal(host(g), c.numpy())


@pytest.mark.parametrize("B,H,W,K1,K2,Cout,stride", [(1, 2, 2, 32, 32, 32, 1), (2, 8, 6, 64, 64, 256, 1), (2, 9, 7, 128, 256, 512, 2),
                                                     (3, 5, 5, 32, 96, 130, 2), (1, 14, 14, 256, 512, 1024, 2), (2, 1, 1, 64, 32, 64, 1)])
@pytest.mark.parametrize("relu", [True, False])
def test_conv1x1_dual_nhwc(ops, B, H, W, K1, K2, Cout, stride, relu):
    from isx._lib import lib
    rng = np.random.default_rng(H * 10 + K1 + stride)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    t = np.maximum(rng.standard_normal((B, Ho, Wo, K1), dtype=np.float32), 0)
    x = np.maximum(rng.standard_normal((B, H, W, K2), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, K1 + K2), dtype=np.float32) * np.float32((K1 + K2) ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    want = O.conv1x1_dual_nhwc(t, x, w, b, stride, relu)
    tt, xt = dev(t).permute(0, 3, 1, 2), dev(x).permute(0, 3, 1, 2)
    set_cfg = lib().isx_debug_set_conv_cfg
    try:
        for cfg in (-1, 0, 2, 3):
            set_cfg(cfg)
            got = host(ops.conv1x1_dual_nhwc(tt, xt, dev(w), dev(b), stride, relu).permute(0, 2, 3, 1))
            np.testing.assert_array_equal(got, want)
    finally:
        set_cfg(-1)
    F = torch.nn.functional
    ref = F.conv2d(tt, dev(w[:, :K1]).reshape(Cout, K1, 1, 1)) + F.conv2d(xt, dev(w[:, K1:]).reshape(Cout, K2, 1, 1), stride=stride) + dev(b).view(1, -1, 1, 1)
    if relu:
        ref = torch.relu(ref)
    np.testing.assert_allclose(got, host(ref.permute(0, 2, 3, 1)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Cin,Cout,res", [(32, 128, False), (64, 200, True), (40, 256, True)])
def test_conv1x1_tail_split(ops, Cin, Cout, res):
    """1x1 convolutions whose grid of 128x128 tiles ends in a partial round (133 956 pixels = 1047 row tiles) run the rows past the last
    whole round as 64x64 tiles in the same launch (conv1x1_tail_kernel): same bits as the plain launch (debug cfg 7) and as the oracle on
    row windows at the start, across the split at row 131 072 and at the ragged end; aligned and unaligned K, ragged Cout."""
    from isx._lib import lib
    M = 133956
    rng = np.random.default_rng(Cin * 7 + Cout)
    x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, Cin), dtype=np.float32) * np.float32(Cin ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    r = rng.standard_normal((M, Cout), dtype=np.float32) if res else None
    xt = dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2)
    rt = dev(r).view(1, M, 1, Cout).permute(0, 3, 1, 2) if res else None
    set_cfg = lib().isx_debug_set_conv_cfg
    try:
        set_cfg(7)
        plain = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), rt, True).permute(0, 2, 3, 1).reshape(M, Cout))
        set_cfg(-1)
        tail = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), rt, True).permute(0, 2, 3, 1).reshape(M, Cout))
    finally:
        set_cfg(-1)
    np.testing.assert_array_equal(tail.view(np.int32), plain.view(np.int32))
    rows = np.r_[0:200, 131072 - 150:131072 + 150, M - 200:M]
    np.testing.assert_array_equal(tail[rows], O.conv1x1_nhwc(x[rows], w, b, r[rows] if res else None, True))


@pytest.mark.parametrize("tiles_m", [1024 + 819, 1024 + 820, 2048 + 1, 1024])
def test_conv1x1_tail_split_boundaries(ops, tiles_m):
    """Around the decision of gemm_tail_split_rows (one whole round + a remainder of at most 80 % of a round is split; a fuller last round,
    an exact number of rounds are not): whichever way the launch goes, the result equals the plain launch bit for bit and the oracle on
    windows at the start, at the 1024-tile boundary and at the ragged end."""
    from isx._lib import lib
    Cin, Cout = 32, 128
    M = tiles_m * 128 - (37 if tiles_m % 2 else 0)
    rng = np.random.default_rng(tiles_m)
    x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, Cin), dtype=np.float32) * np.float32(Cin ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    xt = dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2)
    set_cfg = lib().isx_debug_set_conv_cfg
    try:
        set_cfg(7)
        plain = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), None, True).permute(0, 2, 3, 1).reshape(M, Cout))
        set_cfg(-1)
        auto = host(ops.conv1x1_nhwc(xt, dev(w), dev(b), None, True).permute(0, 2, 3, 1).reshape(M, Cout))
    finally:
        set_cfg(-1)
    np.testing.assert_array_equal(auto.view(np.int32), plain.view(np.int32))
    rows = np.r_[0:100, 131072 - 100:min(M, 131072 + 100), M - 100:M]
    np.testing.assert_array_equal(auto[rows], O.conv1x1_nhwc(x[rows], w, b, None, True))


@pytest.mark.parametrize("stride,Cout", [(1, 128), (2, 256)])
def test_conv1x1_dual_tail_split(ops, stride, Cout):
    """The fused projection GEMM with the 64x64 tail (conv1x1_dual_tail_kernel): same bits as the plain 128x128 launch and as the oracle on
    the first and the last image."""
    from isx._lib import lib
    B, Ho, K1, K2 = 9, 122, 32, 64
    H = Ho * stride
    rng = np.random.default_rng(stride + Cout)
    t = np.maximum(rng.standard_normal((B, Ho, Ho, K1), dtype=np.float32), 0)
    x = np.maximum(rng.standard_normal((B, H, H, K2), dtype=np.float32), 0)
    w = rng.standard_normal((Cout, K1 + K2), dtype=np.float32) * np.float32((K1 + K2) ** -0.5)
    b = rng.standard_normal(Cout, dtype=np.float32)
    tt, xt = dev(t).permute(0, 3, 1, 2), dev(x).permute(0, 3, 1, 2)
    set_cfg = lib().isx_debug_set_conv_cfg
    out = {}
    try:
        for cfg in (0, 7):
            set_cfg(cfg)
            out[cfg] = host(ops.conv1x1_dual_nhwc(tt, xt, dev(w), dev(b), stride, True).permute(0, 2, 3, 1))
    finally:
        set_cfg(-1)
    np.testing.assert_array_equal(out[0].view(np.int32), out[7].view(np.int32))
    for i in (0, B - 1):
        np.testing.assert_array_equal(out[0][i:i + 1], O.conv1x1_dual_nhwc(t[i:i + 1], x[i:i + 1], w, b, stride, True))


@pytest.mark.gpu
def test_blocked_metrics_bit_identical_and_1m_gallery(ops):
    """Exact P@1 / mAP without the whole score matrix (utils.metrics.retrieval_metrics, query-row blocks): bit-identical to
    the one-matrix evaluation at 1 k x 100 k for several block sizes; then a 1 k x 1 M run (4 GB of scores, 1 GB resident
    at a time) whose AP is checked on sampled queries against the oracle's rank_full + average_precision of the same rows."""
    from utils import mean_avg_precision, precision1
    from utils.metrics import retrieval_metrics
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, D = 1000, 100000, 256
    L = N // 10
    cent = torch.randn(L, D, device="cuda", generator=g)
    glab = torch.arange(N, device="cuda") % L
    qlab = torch.arange(M, device="cuda") % L
    G = ops.l2norm_rows(cent[glab] + 4.0 * torch.randn(N, D, device="cuda", generator=g))
    Q = ops.l2norm_rows(cent[qlab] + 4.0 * torch.randn(M, D, device="cuda", generator=g))
    ts = [(None, int(l), None) for l in qlab.tolist()]
    rs = [(None, int(l), None) for l in glab.tolist()]
    sim = ops.cosine_sim(Q, G)
    want_p, want_map = precision1(sim, ts, rs), mean_avg_precision(sim, ts, rs)
    for budget in (None, 4 * N * 128, 4 * N * 37):
        m = retrieval_metrics(Q, G, ts, rs, budget_bytes=budget)
        assert (m["prec1"], m["correct"], m["total"]) == want_p[:3] and m["max_label"] == want_p[4]
        assert torch.equal(m["max_sim"], want_p[3]) and m["mAP"] == want_map
    assert m["blocks"] == (M + 36) // 37 and 0.0 < want_map < 1.0
    # 1 M-row gallery, 1 k queries, 1 GB of scores at a time
    N2 = 1000000
    L2 = N2 // 10
    cent2 = torch.randn(L2, D, device="cuda", generator=g)
    glab2 = torch.arange(N2, device="cuda") % L2
    G2 = torch.empty(N2, D, device="cuda")
    for lo in range(0, N2, 250000):
        G2[lo:lo + 250000] = ops.l2norm_rows(cent2[glab2[lo:lo + 250000]] + 4.0 * torch.randn(250000, D, device="cuda", generator=g))
    Q2 = ops.l2norm_rows(cent2[qlab] + 4.0 * torch.randn(M, D, device="cuda", generator=g))
    rs2 = [(None, int(l), None) for l in glab2.tolist()]
    m2 = retrieval_metrics(Q2, G2, ts, rs2, budget_bytes=1 << 30)
    assert m2["blocks"] == 4 and 0.0 < m2["mAP"] < 1.0
    rows = [0, 500, 999]
    sim_rows = O.cosine_sim(host(Q2[rows]), host(G2))
    ap = O.average_precision(O.rank_full(sim_rows), host(qlab[rows]).astype(np.int32), host(glab2).astype(np.int32))
    from utils.metrics import _average_precisions
    got = _average_precisions(ops.cosine_sim(Q2[rows].contiguous(), G2), qlab[rows].int().cpu(), glab2.int().cpu(), 1)
    np.testing.assert_array_equal(got.numpy(), ap)
