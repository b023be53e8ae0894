"""isx_cosine_topk_fast (fp16-MFMA filter + exact fp32 re-scoring, csrc/fast.hip) must return the SAME
bits as the CPU oracle / isx_cosine_topk for every input: random unit vectors, post-ReLU descriptors,
dense clusters that overflow the candidate window (exact-fallback rows), exact ties, adversarial
column orders, un-normalised magnitudes, values far below the fp16 normal range, NaN-free extremes.
Also: the fp16 building blocks against numpy's IEEE half arithmetic (incl. subnormals)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def unit(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.fixture(scope="module")
def ops():
    from isx import ops as o
    return o


def fallback_rows(ws, M, N, D, k, cached):
    from isx._lib import lib
    f = lib().isx_debug_fast_fallback_rows
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return f(ws.data_ptr(), M, N, D, k, 1 if cached else 0)


def run_fast(ops, Q, G, k, idx_base=0, cached=True):
    Qd, Gd = dev(Q), dev(G)
    M, D = Q.shape
    N = G.shape[0]
    gh = ops.gallery_to_f16(Gd) if cached else None
    ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, cached),), dtype=torch.uint8, device="cuda")
    ts, ti = ops.cosine_topk_fast(Qd, Gd, k, idx_base=idx_base, gallery_f16=gh, ws=ws)
    return host(ts), host(ti), fallback_rows(ws, M, N, D, k, cached)


def check_vs_oracle(ops, Q, G, k, idx_base=0, cached=True):
    want_s, want_i = O.cosine_topk(Q, G, k, idx_base=idx_base)
    ts, ti, fb = run_fast(ops, Q, G, k, idx_base, cached)
    np.testing.assert_array_equal(ti, want_i)
    np.testing.assert_array_equal(ts.view(np.int32), want_s.view(np.int32))
    return fb


@pytest.mark.parametrize("M,N,D,k", [(37, 3000, 64, 10), (130, 9000, 256, 100), (5, 40000, 48, 1), (64, 12000, 128, 128),
                                     (257, 20000, 512, 33), (3, 70000, 8, 5)])
@pytest.mark.parametrize("cached", [True, False])
def test_random_unit_vectors(ops, M, N, D, k, cached):
    rng = np.random.default_rng(N + k)
    Q, G = unit(rng, M, D), unit(rng, N, D)
    G[N // 2] = G[3]
    G[N - 1] = G[3]                               # duplicated rows: exact ties, index order decides
    fb = check_vs_oracle(ops, Q, G, k, idx_base=11, cached=cached)
    assert fb >= 0                                # the filter path ran (not the whole-call fp32 delegate)


def test_post_relu_descriptors(ops):
    rng = np.random.default_rng(5)
    Q = np.maximum(rng.standard_normal((200, 512), dtype=np.float32), 0)
    G = np.maximum(rng.standard_normal((20000, 512), dtype=np.float32), 0)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    check_vs_oracle(ops, Q, G, 50)


def test_dense_clusters_take_the_exact_fallback(ops):
    """Every gallery row is a 1e-4 perturbation of one of 8 centres: thousands of scores fall inside the
    error window of the k-th best, the KL candidates cannot cover it, the rows must be recomputed exactly."""
    rng = np.random.default_rng(7)
    c = unit(rng, 8, 256)
    G = c[rng.integers(0, 8, 20000)] + 1e-4 * rng.standard_normal((20000, 256), dtype=np.float32)
    G = (G / np.linalg.norm(G, axis=1, keepdims=True)).astype(np.float32)
    Q = unit(rng, 150, 256)
    fb = check_vs_oracle(ops, Q, G, 20)
    assert fb == 150


def test_mixed_rows_some_fallback(ops):
    rng = np.random.default_rng(8)
    c = unit(rng, 1, 256)
    G = np.concatenate([unit(rng, 10000, 256), c + 1e-5 * rng.standard_normal((2000, 256), dtype=np.float32)]).astype(np.float32)
    Q = np.concatenate([unit(rng, 100, 256), c + 1e-3 * rng.standard_normal((100, 256), dtype=np.float32)]).astype(np.float32)
    fb = check_vs_oracle(ops, Q, G, 30, idx_base=5)
    assert 0 < fb < 200                           # near-centre queries fall back, the others do not


@pytest.mark.parametrize("order", ["ascending", "descending"])
def test_adversarial_column_order(ops, order):
    rng = np.random.default_rng(9)
    M, N, D, k = 21, 40000, 64, 100
    Q, G = unit(rng, M, D), unit(rng, N, D)
    s0 = G @ Q[0]
    G = G[np.argsort(s0 if order == "ascending" else -s0)]
    G[9000] = G[17]; G[39999] = G[17]; G[8191] = G[8192]
    check_vs_oracle(ops, Q, G, k)


@pytest.mark.parametrize("qs,gs", [(300.0, 1000.0), (1e-4, 1e-3), (1.0, 1e-6), (3e4, 3e4)])
def test_unnormalised_magnitudes(ops, qs, gs):
    rng = np.random.default_rng(13)
    Q, G = unit(rng, 80, 128) * np.float32(qs), unit(rng, 8000, 128) * np.float32(gs)
    fb = check_vs_oracle(ops, Q, G, 10)
    assert fb >= 0


def test_out_of_fp16_range_runs_exact(ops):
    rng = np.random.default_rng(14)
    Q, G = unit(rng, 50, 128) * np.float32(1e6), unit(rng, 8000, 128)
    fb = check_vs_oracle(ops, Q, G, 10)
    assert fb == 50                               # max|Q| > 2^15: every row takes the exact path


def test_wide_dynamic_range_inside_rows(ops):
    """Half of every row is 1e-7 times smaller than the rest (below the fp16 normal range even after scaling)."""
    rng = np.random.default_rng(15)
    Q = np.concatenate([unit(rng, 90, 128), 1e-7 * unit(rng, 90, 128)], axis=1).astype(np.float32)
    G = np.concatenate([unit(rng, 9000, 128), 1e-7 * unit(rng, 9000, 128)], axis=1).astype(np.float32)
    check_vs_oracle(ops, Q, G, 10)


@pytest.mark.parametrize("M,N,D,k", [(10, 200, 64, 10), (10, 5000, 60, 10), (10, 5000, 64, 200)])
def test_shapes_outside_the_filter_delegate_to_fp32(ops, M, N, D, k):
    rng = np.random.default_rng(16)
    Q, G = unit(rng, M, D), unit(rng, N, D)
    fb = check_vs_oracle(ops, Q, G, k, cached=False)
    assert fb == -1


def test_big_tiles_and_many_chunks(ops):
    """Large enough for the 256x256 fp16 tile and several filtered chunks; compared with isx_cosine_topk."""
    g = torch.Generator(device="cuda").manual_seed(3)
    Q = torch.randn(3000, 256, device="cuda", generator=g)
    G = torch.randn(60000, 256, device="cuda", generator=g)
    Q, G = ops.l2norm_rows(Q), ops.l2norm_rows(G)
    ref = ops.cosine_topk(Q, G, 100, idx_base=7)
    got = ops.cosine_topk_fast(Q, G, 100, idx_base=7, gallery_f16=ops.gallery_to_f16(G))
    assert torch.equal(ref[1], got[1])
    assert torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32))
    # small workspace: narrow chunks, same answer
    need = ops.cosine_topk_fast_workspace(3000, 60000, 256, 100, True)
    ws = torch.empty((need // 3,), dtype=torch.uint8, device="cuda")
    got = ops.cosine_topk_fast(Q, G, 100, idx_base=7, gallery_f16=ops.gallery_to_f16(G), ws=ws)
    assert torch.equal(ref[1], got[1]) and torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32))


# ------------------------------------------------------------------ building blocks
def test_rows_to_f16_matches_ieee_half(ops):
    rng = np.random.default_rng(20)
    x = rng.standard_normal((67, 200), dtype=np.float32)
    x[0, :50] = rng.standard_normal(50).astype(np.float32) * 1e-6       # fp16 subnormal range
    x[1, :8] = [65504.0, -65504.0, 6.1e-5, 5.96e-8, 2.9e-8, 0.0, -0.0, 1.0009765625]
    h, n2, am = ops.rows_to_f16(dev(x))
    np.testing.assert_array_equal(host(h).view(np.uint16), x.astype(np.float16).view(np.uint16))     # RNE, gradual underflow
    assert (host(n2) >= (x.astype(np.float64) ** 2).sum(1) * (1 - 1e-6)).all()
    np.testing.assert_array_equal(host(am), np.abs(x).max(1))


@pytest.mark.parametrize("M,N,D", [(130, 257, 64), (300, 1000, 200), (257, 600, 2048)])
def test_f16_gemm_is_exact_on_half_inputs(ops, M, N, D):
    """fp16 products are exact in fp32; only the fp32 accumulation rounds: <= D * 2^-23 relative to sum |q g|.
    Rows include fp16 SUBNORMALS: the matrix cores must not flush them (the error bound of the fast path
    assumes flushing at worst, this documents the actual behaviour)."""
    rng = np.random.default_rng(D)
    q = rng.standard_normal((M, D)).astype(np.float16)
    g = rng.standard_normal((N, D)).astype(np.float16)
    q[0] = (rng.standard_normal(D) * 3e-6).astype(np.float16)            # subnormal halves
    g[0] = (rng.standard_normal(D) * 3e-6).astype(np.float16)
    sim = host(ops.cosine_sim_f16(dev(q), dev(g)))
    want = q.astype(np.float64) @ g.astype(np.float64).T
    bound = (np.abs(q.astype(np.float64)) @ np.abs(g.astype(np.float64)).T) * D * 2.0 ** -23 + 1e-30
    assert (np.abs(sim - want) <= bound).all()
    assert np.abs(sim[0, 1:]).max() > 0                                  # subnormal operands contributed


@pytest.fixture
def force_f16_tile():
    """isx_debug_set_f16_tile: 0 = 128x128, 1 = 256x256 (ping-pong LDS-DMA kernel when D % 64 == 0), 2 = 256x256 register-staged."""
    from isx._lib import lib
    f = lib().isx_debug_set_f16_tile
    f.restype, f.argtypes = None, [ctypes.c_int]
    yield f
    f(-1)


@pytest.mark.parametrize("M,N,D", [(130, 257, 64), (300, 1000, 128), (513, 700, 192), (257, 600, 2048), (256, 512, 256), (1, 1, 64)])
def test_f16_pingpong_kernel_edge_shapes(ops, force_f16_tile, M, N, D):
    """The ping-pong 256x256 kernel (LDS-DMA half-tiles, 16x16x32 MFMA; csrc/fast.hip 2c) on ragged M / N, one, two and three
    k-tiles: same bits as the 128x128 and the register-staged 256x256 kernels, and inside the fp32-accumulation bound."""
    rng = np.random.default_rng(M + N + D)
    q = rng.standard_normal((M, D)).astype(np.float16)
    g = rng.standard_normal((N, D)).astype(np.float16)
    out = {}
    for tile in (0, 1, 2):
        force_f16_tile(tile)
        out[tile] = host(ops.cosine_sim_f16(dev(q), dev(g)))
    np.testing.assert_array_equal(out[1].view(np.int32), out[0].view(np.int32))
    np.testing.assert_array_equal(out[1].view(np.int32), out[2].view(np.int32))
    want = q.astype(np.float64) @ g.astype(np.float64).T
    bound = (np.abs(q.astype(np.float64)) @ np.abs(g.astype(np.float64)).T) * D * 2.0 ** -23 + 1e-30
    assert (np.abs(out[1] - want) <= bound).all()


@pytest.mark.parametrize("M,N,D,k", [(37, 9000, 64, 10), (300, 20000, 128, 100), (513, 30000, 192, 33)])
def test_fast_search_through_the_pingpong_filter_on_small_shapes(ops, force_f16_tile, M, N, D, k):
    """Forces the 256x256 ping-pong filter kernel (normally reserved for launches of >= 512 tiles) on shapes with ragged edges
    in both the plain (bootstrap chunk) and the filter epilogue: the search result must still equal the oracle bit for bit."""
    rng = np.random.default_rng(k)
    Q, G = unit(rng, M, D), unit(rng, N, D)
    force_f16_tile(1)
    check_vs_oracle(ops, Q, G, k, idx_base=3)
    # narrow chunks: several filtered chunks with a ragged last one
    Qd, Gd = dev(Q), dev(G)
    need = ops.cosine_topk_fast_workspace(M, N, D, k, True)
    ws = torch.empty((max(need // 3, 1 << 20),), dtype=torch.uint8, device="cuda")
    want_s, want_i = O.cosine_topk(Q, G, k, idx_base=3)
    ts, ti = ops.cosine_topk_fast(Qd, Gd, k, idx_base=3, gallery_f16=ops.gallery_to_f16(Gd), ws=ws)
    np.testing.assert_array_equal(host(ti), want_i)
    np.testing.assert_array_equal(host(ts).view(np.int32), want_s.view(np.int32))


def test_gallery_to_f16_scaling(ops):
    rng = np.random.default_rng(21)
    G = unit(rng, 500, 64) * np.float32(0.3)
    amax = np.abs(G).max()
    scale = 2.0 ** (13 - np.floor(np.log2(amax)))
    G[3, :5] = np.float32(1e-10) * np.arange(1, 6, dtype=np.float32)            # below the fp16 normal range after scaling: stored as zero
    gh, gstats = ops.gallery_to_f16(dev(G))
    want = (G * np.float32(scale)).astype(np.float16)
    want[np.abs(want.astype(np.float32)) < 2.0 ** -14] = 0
    assert (want[3, :5] == 0).all()
    np.testing.assert_array_equal(host(gh).view(np.uint16), want.view(np.uint16))
    st = host(gstats)
    assert st.shape == (4,) and st[1] == amax and st[0] >= (G.astype(np.float64) ** 2).sum(1).max() * (1 - 1e-6)
    # st[2]: the largest squared norm of what a row lost in the conversion -- an upper bound, and a tight one
    lost = ((G.astype(np.float64) - want.astype(np.float64) / scale) ** 2).sum(1).max()
    assert lost <= st[2] <= lost * 1.001
    # the loss is what the error bound of the search now rests on: well below the format's worst case 2^-11 |g|
    assert st[2] < (2.0 ** -11) ** 2 * st[0] * 0.5


def _f16_scale(amax):
    return 2.0 ** (13 - np.floor(np.log2(amax)))


@pytest.mark.parametrize("case", ["random", "coherent", "tiny_tail", "alternating", "one_hot_plus_dust"])
def test_fast_error_bound_covers_the_approximate_scores(ops, case):
    """The window of the exact-fast search rests on |S' - S| <= eps_i with eps_i built from what the fp16 conversion LOST (fast.hip header):
    eps_i = 1.01 (rq_i (|g|max + rg) + |q_i| rg + 3 D 2^-24 (|q_i| + rq_i)(|g|max + rg)).  Checked pair by pair against float64 scores on
    rows built to make the bound work hard: `coherent` = every element of every row the same value a hair below an fp16 rounding midpoint
    (all losses have one sign and Cauchy-Schwarz is an equality: the bound must be nearly TIGHT there, which is what gives this test teeth)."""
    rng = np.random.default_rng(5)
    M, N, D = 96, 160, 2048
    if case == "random":
        Q, G = unit(rng, M, D), unit(rng, N, D)
    elif case == "coherent":
        c = np.float32(1.0 + 2.0 ** -11 * 0.998)                       # between the halves 1 and 1 + 2^-10: rounds down, loses ~2^-11
        Q = np.full((M, D), c, np.float32)                             # |q| = c sqrt(D): the scaling is a power of two, c stays a hair below the midpoint
        G = np.full((N, D), c, np.float32)
        Q[1::2] *= np.float32(-1)
    elif case == "tiny_tail":
        Q, G = unit(rng, M, D), unit(rng, N, D)
        Q[:, D // 2:] *= np.float32(1e-7); G[:, D // 3:] *= np.float32(3e-8)     # flushed to zero by the conversion: the loss is the element itself
    elif case == "alternating":
        c = np.float32(1.0 + 2.0 ** -11 * 0.998)
        sgn = np.where(np.arange(D) % 2 == 0, 1, -1).astype(np.float32)
        Q = np.full((M, D), c, np.float32) * sgn
        G = np.full((N, D), c, np.float32) * sgn
    else:
        Q = np.zeros((M, D), np.float32); G = np.zeros((N, D), np.float32)
        Q[np.arange(M), rng.integers(0, D, M)] = 1; G[np.arange(N), rng.integers(0, D, N)] = 1
        Q += rng.standard_normal((M, D)).astype(np.float32) * np.float32(2e-5); G += rng.standard_normal((N, D)).astype(np.float32) * np.float32(2e-5)
    qh, qst = ops.gallery_to_f16(dev(Q))
    gh, gst = ops.gallery_to_f16(dev(G))
    qst, gst = host(qst), host(gst)
    sq, sg = _f16_scale(qst[1]), _f16_scale(gst[1])
    approx = host(ops.cosine_sim_f16(qh, gh)).astype(np.float64) / (sq * sg)
    exact = Q.astype(np.float64) @ G.astype(np.float64).T
    qt, gt = host(qh).astype(np.float64) / sq, host(gh).astype(np.float64) / sg
    rq = np.sqrt(((Q - qt) ** 2).sum(1)); rg_rows = np.sqrt(((G - gt) ** 2).sum(1))
    qn = np.sqrt((Q.astype(np.float64) ** 2).sum(1))
    gn, rg = np.sqrt(float(gst[0])), np.sqrt(float(gst[2]))
    assert rg >= rg_rows.max() and rg <= rg_rows.max() * 1.001 + 1e-30           # the statistic the library keeps IS the largest loss
    eps = 1.01 * (rq * (gn + rg) + qn * rg + 3 * D * 2.0 ** -24 * (qn + rq) * (gn + rg))
    err = np.abs(approx - exact)
    assert (err <= eps[:, None]).all(), (case, float((err / eps[:, None]).max()))
    if case == "coherent":
        assert (err / eps[:, None]).max() > 0.6                                   # the bound is within 1.7x of an error that really occurs (rounding term: equality)
    if case == "random":
        assert eps.max() < 1e-3                                                    # ~7.6e-4 at D = 2048, against 1.34e-3 for the format's worst case


def test_rows_on_both_sides_of_a_rounding_midpoint(ops):
    """Constant rows whose value straddles an fp16 rounding midpoint: every loss has one sign (the error bound's Cauchy-Schwarz step is an
    equality), rows below the midpoint collapse onto ONE approximate score while their exact scores all differ, rows above it jump by 2^-10.
    The windows of the queries that rank into the collapsed block cannot be covered by the candidate list -- those rows must take the exact
    fallback -- and the lists must still be the oracle's, bit for bit, in canonical order."""
    rng = np.random.default_rng(31)
    D, N, M, k = 256, 6000, 40, 50
    mid = 1.0 + 2.0 ** -11
    c = (mid + (rng.random(N) - 0.99) * 2.0 ** -11 * 0.9).astype(np.float32)           # ~1 % of the rows above the midpoint, the rest below
    G = np.repeat(c[:, None], D, 1)
    G[:, 0] += (rng.random(N).astype(np.float32) - 0.5) * np.float32(1e-3)             # a little individuality outside the collapsed pattern
    Q = np.abs(unit(rng, M, D)) + np.float32(0.05)
    fb = check_vs_oracle(ops, Q, G.astype(np.float32), k, idx_base=2)
    assert fb > 0


def test_sharded_gallery_fast_equals_fp32(ops):
    from isx.retrieval import ShardedGallery
    g = torch.Generator(device="cuda").manual_seed(4)
    Q = ops.l2norm_rows(torch.randn(500, 128, device="cuda", generator=g))
    G = ops.l2norm_rows(torch.randn(30000, 128, device="cuda", generator=g))
    a = ShardedGallery(G, 1000, fast=False).search(Q, 20)
    b = ShardedGallery(G, 1000, fast=True).search(Q, 20)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32))


def test_randomised_shapes_fast_equals_fp32(ops):
    """40 random (M, N, D, k, workspace, clustering) configurations: the fast path and the all-fp32 path return the same bits."""
    rng = np.random.default_rng(2026)
    g = torch.Generator(device="cuda").manual_seed(77)
    for trial in range(40):
        M = int(rng.integers(1, 700))
        N = int(rng.integers(300, 60000))
        D = int(rng.choice([8, 24, 64, 128, 200, 512, 1000]))
        k = int(rng.integers(1, 129))
        Q = torch.randn(M, D, device="cuda", generator=g)
        G = torch.randn(N, D, device="cuda", generator=g)
        mode = trial % 4
        if mode == 1:                                   # near-duplicate gallery rows: dense score clusters
            G = G[torch.randint(0, max(2, N // 50), (N,), device="cuda", generator=g)] + 1e-4 * torch.randn(N, D, device="cuda", generator=g)
        elif mode == 2:                                 # non-negative descriptors
            Q, G = Q.relu(), G.relu() + 1e-3
        elif mode == 3:                                 # exact duplicates
            G[N // 2:] = G[:N - N // 2].clone()
        Q, G = ops.l2norm_rows(Q), ops.l2norm_rows(G)
        idx_base = int(rng.integers(0, 1000))
        ref = ops.cosine_topk(Q, G, k, idx_base=idx_base)
        need = ops.cosine_topk_fast_workspace(M, N, D, k, trial % 2 == 0)
        ws = torch.empty((need if trial % 3 else max(need // 2, 1 << 20),), dtype=torch.uint8, device="cuda")
        gh = ops.gallery_to_f16(G) if trial % 2 == 0 else None
        try:
            got = ops.cosine_topk_fast(Q, G, k, idx_base=idx_base, gallery_f16=gh, ws=ws)
        except Exception as e:                          # a halved workspace may be below the documented minimum: that must be an error, not a wrong answer
            assert "workspace" in str(e) and trial % 3 == 0
            continue
        assert torch.equal(ref[1], got[1]), (trial, M, N, D, k, mode)
        assert torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32)), (trial, M, N, D, k, mode)


def test_non_finite_inputs_take_the_exact_path(ops):
    """NaN / Inf anywhere in Q or G makes the error bound meaningless: every row must run the exact fp32 search, so the
    two entry points still agree bit for bit (whatever order the canonical key gives non-finite scores)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    Q = ops.l2norm_rows(torch.randn(64, 128, device="cuda", generator=g))
    G = ops.l2norm_rows(torch.randn(6000, 128, device="cuda", generator=g))
    for where in ("q_nan", "g_inf"):
        Q2, G2 = Q.clone(), G.clone()
        if where == "q_nan":
            Q2[3, 7] = float("nan")
        else:
            G2[100, 5] = float("inf")
        ref = ops.cosine_topk(Q2, G2, 10)
        need = ops.cosine_topk_fast_workspace(64, 6000, 128, 10, False)
        ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
        got = ops.cosine_topk_fast(Q2, G2, 10, ws=ws)
        assert fallback_rows(ws, 64, 6000, 128, 10, False) == 64
        assert torch.equal(ref[1], got[1])
        assert torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32))


def test_full_size_shard_invariance(ops):
    """Config-3/5 scale (2 k queries x 400 k rows x 2048, top-100): the fast search on the whole gallery, on 8 row shards +
    merge, and the all-fp32 search give the same bits; sampled rows against the CPU oracle."""
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, D, k, P = 2000, 400000, 2048, 100, 8
    Q = ops.l2norm_rows(torch.randn(M, D, device="cuda", generator=g))
    G = torch.empty(N, D, device="cuda")
    for i in range(0, N, 100000):
        G[i:i + 100000] = ops.l2norm_rows(torch.randn(100000, D, device="cuda", generator=g))
    ts, ti = ops.cosine_topk_fast(Q, G, k, gallery_f16=ops.gallery_to_f16(G))
    assert bool((ts[:, :-1] >= ts[:, 1:]).all()) and int(ti.min()) >= 0 and int(ti.max()) < N
    parts = [ops.cosine_topk_fast(Q, G[p * N // P:(p + 1) * N // P], k, idx_base=p * N // P) for p in range(P)]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, ti) and torch.equal(ms.view(torch.int32), ts.view(torch.int32))
    ref = ops.cosine_topk(Q, G, k)
    assert torch.equal(ref[1], ti) and torch.equal(ref[0].view(torch.int32), ts.view(torch.int32))
    rows = [0, 777, 1999]
    os_, oi = O.cosine_topk(host(Q[rows]), host(G), k)
    np.testing.assert_array_equal(host(ti[rows]), oi)
    np.testing.assert_array_equal(host(ts[rows]), os_)


def test_replicated_gallery_single_process(ops):
    from isx.retrieval import ReplicatedGallery, ShardedGallery
    g = torch.Generator(device="cuda").manual_seed(6)
    Q = ops.l2norm_rows(torch.randn(300, 64, device="cuda", generator=g))
    G = ops.l2norm_rows(torch.randn(20000, 64, device="cuda", generator=g))
    a = ReplicatedGallery(G).search(Q, 10)
    b = ShardedGallery(G, 0, fast=False).search(Q, 10)
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])


def test_sharded_gallery_leaves_the_fast_path_when_the_data_defeats_it(ops):
    """Every row falls back on a gallery of tight clusters: after the first search the gallery switches itself to the fp32
    search; results are the same before and after.  A well-separated gallery stays on the fast path."""
    from isx.retrieval import ShardedGallery
    g = torch.Generator(device="cuda").manual_seed(8)
    Q = ops.l2norm_rows(torch.randn(200, 128, device="cuda", generator=g))
    c = ops.l2norm_rows(torch.randn(8, 128, device="cuda", generator=g))
    G = ops.l2norm_rows(c[torch.randint(0, 8, (20000,), device="cuda", generator=g)] + 1e-4 * torch.randn(20000, 128, device="cuda", generator=g))
    gal = ShardedGallery(G, 0)
    a = gal.search(Q, 20)
    torch.cuda.synchronize()
    b = gal.search(Q, 20)                          # reads the counter of the first search
    assert gal.fast is False
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    ok = ShardedGallery(ops.l2norm_rows(torch.randn(20000, 128, device="cuda", generator=g)), 0)
    ok.search(Q, 20); torch.cuda.synchronize(); ok.search(Q, 20)
    assert ok.fast is True
    counter = ops.cosine_topk_fast_fallback_counter(ok._ws, 200, 20000, 128, 20, True)
    torch.cuda.synchronize()
    assert int(counter.item()) == 0


def _oracle_topk_chunked(Qrows, G, k, chunk=125000):
    """O.cosine_topk of a few query rows against a gallery that stays on the GPU: 125k-row pieces are copied to the host one at a
    time (1 GB each), searched by the scalar oracle with their global idx_base, and merged with the canonical comparator."""
    q = host(Qrows)
    cs, ci = [], []
    for lo in range(0, G.size(0), chunk):
        s, i = O.cosine_topk(q, host(G[lo:lo + chunk]), k, idx_base=lo)
        cs.append(s); ci.append(i)
    s, i = np.concatenate(cs, 1), np.concatenate(ci, 1)
    out_s, out_i = np.empty((q.shape[0], k), np.float32), np.empty((q.shape[0], k), np.int64)
    for r in range(q.shape[0]):
        order = np.lexsort((i[r], -s[r].astype(np.float64)))[:k]                 # score desc, index asc
        out_s[r], out_i[r] = s[r][order], i[r][order]
    return out_s, out_i


def test_config5_full_size_on_one_gpu(ops):
    """BASELINE configs[4] at its stated size: 10 000 queries x 1 000 000 gallery rows x 2048, top-100, on ONE MI355X (8.2 GB
    gallery + 4.1 GB fp16 image).  (a) the unsharded fast search, (b) the 8 x 125 k shards a node would hold, each searched with
    its idx_base, merged with isx_topk_merge: identical bits; (c) the first shard's own 10 k x 125 k lists and (d) the merged
    lists against the CPU oracle on sampled rows; (e) the all-fp32 search on sampled query rows."""
    g = torch.Generator(device="cuda").manual_seed(0)
    M, N, D, k, P = 10000, 1000000, 2048, 100, 8
    Q = ops.l2norm_rows(torch.randn(M, D, device="cuda", generator=g))
    G = torch.empty(N, D, device="cuda")
    for i in range(0, N, 125000):
        G[i:i + 125000] = ops.l2norm_rows(torch.randn(125000, D, device="cuda", generator=g))
    ts, ti = ops.cosine_topk_fast(Q, G, k, gallery_f16=ops.gallery_to_f16(G))
    assert bool((ts[:, :-1] >= ts[:, 1:]).all()) and int(ti.min()) >= 0 and int(ti.max()) < N
    parts = [ops.cosine_topk_fast(Q, G[p * N // P:(p + 1) * N // P], k, idx_base=p * N // P) for p in range(P)]
    ms, mi = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(mi, ti) and torch.equal(ms.view(torch.int32), ts.view(torch.int32))
    rows = [0, 4242, 9999]
    os_, oi = O.cosine_topk(host(Q[rows]), host(G[:N // P]), k)                               # (c) the per-GPU shard of config 5
    np.testing.assert_array_equal(host(parts[0][1][rows]), oi)
    np.testing.assert_array_equal(host(parts[0][0][rows]), os_)
    ws_, wi_ = _oracle_topk_chunked(Q[rows], G, k)                                            # (d) whole gallery
    np.testing.assert_array_equal(host(ti[rows]), wi_)
    np.testing.assert_array_equal(host(ts[rows]), ws_)
    fs, fi = ops.cosine_topk(Q[:64].contiguous(), G, k)                                       # (e) every score on the fp32 matrix cores
    assert torch.equal(fi, ti[:64]) and torch.equal(fs.view(torch.int32), ts[:64].view(torch.int32))
