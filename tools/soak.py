"""Randomised soak test (not part of the suite): many random shapes through the kernels with bit-exact comparisons.
    python tools/soak.py [seconds]"""
import sys, time, numpy as np, torch
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "instance-search_amd")); sys.path.insert(0, os.path.join(_ROOT, "oracle"))
from isx import ops
import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
t0 = time.time(); n = {"fast": 0, "conv1big": 0, "conv1": 0, "conv3": 0, "dual": 0, "topk": 0, "sel": 0, "stem": 0, "pool": 0, "region": 0, "dba": 0, "gemm": 0, "ap": 0}
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
last = t0
while time.time() - t0 < budget:
    if time.time() - last > 60:                      # a line a minute: gpurun takes seven silent minutes for a hang
        last = time.time(); print("... %.0f s" % (last - t0), n, flush=True)
    kind = rng.integers(0, 13)
    if len(sys.argv) > 2: kind = int(rng.choice([int(v) for v in sys.argv[2].split(',')]))      # restrict to some kinds: soak.py 300 0,11
    if kind == 11:     # sort-free AP vs full sort + AP (float64, bit for bit), any number of positives per query, tied scores
        M = int(rng.integers(1, 60)); N = int(rng.integers(10, 70000)); L = int(rng.integers(1, max(2, N // int(rng.integers(1, 40))) + 1)); kth = int(rng.integers(1, 4))
        sim = torch.round(torch.rand(M, N, device="cuda", generator=g) * float(rng.choice([50, 1000, 1e6]))) / 1000
        if rng.integers(3) == 0: sim = sim.sort(dim=1, descending=bool(rng.integers(2))).values
        gl = torch.randint(0, L, (N,), device="cuda", generator=g).int(); ql = torch.randint(0, L + 1, (M,), device="cuda", generator=g).int()
        a = ops.average_precision_sim(sim, ql, gl, kth); b = ops.average_precision(ops.rank_full(sim), ql, gl, kth)
        assert torch.equal(a.isnan(), b.isnan()) and torch.equal(a[~a.isnan()], b[~b.isnan()]), ("ap", M, N, L, kth)
        n["ap"] += 1
    elif kind == 0:      # fast vs fp32 search
        M = int(rng.integers(1, 3000)); N = int(rng.integers(300, 200000)); D = int(rng.choice([8, 16, 64, 96, 256, 512, 1024, 2048])); k = int(rng.integers(1, 129))
        if M * N * D > 4e11: continue
        Q = torch.randn(M, D, device="cuda", generator=g); G = torch.randn(N, D, device="cuda", generator=g)
        mode = int(rng.integers(0, 4))
        if mode == 1: G = G[torch.randint(0, max(2, N // int(rng.integers(2, 200))), (N,), device="cuda", generator=g)] + float(10 ** rng.uniform(-6, -2)) * torch.randn(N, D, device="cuda", generator=g)
        if mode == 2: Q, G = Q.relu() + 1e-4, G.relu() + 1e-4
        if mode == 3: G[N // 2:] = G[:N - N // 2].clone()
        Q, G = ops.l2norm_rows(Q) * float(10 ** rng.uniform(-3, 3)), ops.l2norm_rows(G) * float(10 ** rng.uniform(-3, 3))
        ib = int(rng.integers(0, 10 ** 6))
        a = ops.cosine_topk(Q, G, k, idx_base=ib); b = ops.cosine_topk_fast(Q, G, k, idx_base=ib, gallery_f16=ops.gallery_to_f16(G) if rng.integers(2) else None)
        assert torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), ("fast", M, N, D, k, mode)
        n["fast"] += 1
    elif kind == 12:   # conv1x1 on FORCED 128x128 tiles (the whole-tile residual fetch of round 6, ragged last tile rows / columns) vs oracle
        from isx._lib import lib
        M = int(rng.integers(1, 6000)); Cin = int(rng.choice([32, 64, 128])); Cout = int(rng.choice([36, 64, 128, 130, 256]))
        x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0); w = rng.standard_normal((Cout, Cin), dtype=np.float32) * 0.1
        b = rng.standard_normal(Cout, dtype=np.float32); res = rng.integers(2); relu = bool(rng.integers(2))
        r = rng.standard_normal((M, Cout), dtype=np.float32) if res else None
        lib().isx_debug_set_gemm_cfg(0)
        try:
            y = ops.conv1x1_nhwc(dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2), dev(w), dev(b), dev(r).view(1, M, 1, Cout).permute(0, 3, 1, 2) if res else None, relu)
        finally:
            lib().isx_debug_set_gemm_cfg(-1)
        assert np.array_equal(y.permute(0, 2, 3, 1).reshape(M, Cout).cpu().numpy(), O.conv1x1_nhwc(x, w, b, r, relu)), ("conv1big", M, Cin, Cout, res)
        n["conv1big"] += 1
    elif kind == 1:    # conv1x1 vs oracle
        M = int(rng.integers(1, 3000)); Cin = int(rng.choice([4, 32, 64, 100, 256, 512])); Cout = int(rng.choice([4, 36, 64, 128, 130, 512]))
        x = np.maximum(rng.standard_normal((M, Cin), dtype=np.float32), 0); w = rng.standard_normal((Cout, Cin), dtype=np.float32) * 0.1
        b = rng.standard_normal(Cout, dtype=np.float32); res = rng.integers(2); relu = bool(rng.integers(2))
        r = rng.standard_normal((M, Cout), dtype=np.float32) if res else None
        y = ops.conv1x1_nhwc(dev(x).view(1, M, 1, Cin).permute(0, 3, 1, 2), dev(w), dev(b), dev(r).view(1, M, 1, Cout).permute(0, 3, 1, 2) if res else None, relu)
        assert np.array_equal(y.permute(0, 2, 3, 1).reshape(M, Cout).cpu().numpy(), O.conv1x1_nhwc(x, w, b, r, relu)), ("conv1", M, Cin, Cout)
        n["conv1"] += 1
    elif kind == 2:    # conv3x3 vs oracle
        B = int(rng.integers(1, 4)); H = int(rng.integers(1, 20)); W = int(rng.integers(1, 20)); Cin = int(rng.choice([32, 64, 96, 128])); Cout = int(rng.choice([32, 64, 100, 256])); s = int(rng.integers(1, 3))
        x = np.maximum(rng.standard_normal((B, H, W, Cin), dtype=np.float32), 0); w = rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.05; b = rng.standard_normal(Cout, dtype=np.float32)
        y = ops.conv3x3_nhwc(dev(x).permute(0, 3, 1, 2), dev(w), dev(b), s, None, True)
        assert np.array_equal(y.permute(0, 2, 3, 1).cpu().numpy(), O.conv3x3_nhwc(x, w, b, s, None, True)), ("conv3", B, H, W, Cin, Cout, s)
        n["conv3"] += 1
    elif kind == 3:    # dual
        B = int(rng.integers(1, 4)); H = int(rng.integers(1, 16)); W = int(rng.integers(1, 16)); K1 = int(rng.choice([32, 64, 128])); K2 = int(rng.choice([32, 64, 256])); Cout = int(rng.choice([32, 100, 256])); s = int(rng.integers(1, 3))
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        t = np.maximum(rng.standard_normal((B, Ho, Wo, K1), dtype=np.float32), 0); x = np.maximum(rng.standard_normal((B, H, W, K2), dtype=np.float32), 0)
        w = rng.standard_normal((Cout, K1 + K2), dtype=np.float32) * 0.1; b = rng.standard_normal(Cout, dtype=np.float32)
        y = ops.conv1x1_dual_nhwc(dev(t).permute(0, 3, 1, 2), dev(x).permute(0, 3, 1, 2), dev(w), dev(b), s, True)
        assert np.array_equal(y.permute(0, 2, 3, 1).cpu().numpy(), O.conv1x1_dual_nhwc(t, x, w, b, s, True)), ("dual", B, H, W, K1, K2, Cout, s)
        n["dual"] += 1
    elif kind == 4:    # fp32 fused topk vs materialised topk_rows
        M = int(rng.integers(1, 500)); N = int(rng.integers(1, 120000)); D = int(rng.choice([8, 20, 64, 256])); k = int(rng.integers(1, 400))
        Q = torch.randn(M, D, device="cuda", generator=g); G = torch.randn(N, D, device="cuda", generator=g)
        a = ops.cosine_topk(Q, G, k); sim = ops.cosine_sim(Q, G); b = ops.topk_rows(sim, k)
        assert torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), ("topk", M, N, D, k)
        n["topk"] += 1
    elif kind == 5:    # topk_rows vs torch (values; ties only by equal scores)
        M = int(rng.integers(1, 300)); N = int(rng.integers(1, 50000)); k = int(rng.integers(1, min(N, 256) + 1))
        sim = torch.randn(M, N, device="cuda", generator=g)
        s_, i_ = ops.topk_rows(sim, k); ts = torch.topk(sim, k, dim=1).values
        assert torch.equal(s_, ts) and torch.equal(sim.gather(1, i_), s_), ("sel", M, N, k)
        n["sel"] += 1
    elif kind == 6:    # fused stem incl. column bands (W up to 896) vs oracle
        B = int(rng.integers(1, 4)); H = int(rng.integers(1, 120)); W = 4 * int(rng.integers(1, 225))
        x = rng.standard_normal((B, H, W, 3), dtype=np.float32); w = rng.standard_normal((64, 7, 7, 3), dtype=np.float32) * np.float32(147 ** -0.5); b = rng.standard_normal(64, dtype=np.float32)
        y = ops.stem7x7_pool(dev(x).permute(0, 3, 1, 2), dev(w), dev(b))
        assert np.array_equal(y.permute(0, 2, 3, 1).cpu().numpy(), O.stem7x7_pool_nhwc(x, w, b)), ("stem", B, H, W)
        n["stem"] += 1
    elif kind == 7:    # channels-last box pooling vs oracle
        B = int(rng.integers(1, 4)); C = 4 * int(rng.integers(1, 80)); kh = int(rng.integers(1, 8)); kw = int(rng.integers(1, 8)); H = kh + int(rng.integers(0, 12)); W = kw + int(rng.integers(0, 12))
        f = rng.standard_normal((B, C, H, W), dtype=np.float32)
        ft = dev(f).contiguous(memory_format=torch.channels_last)
        if not ops.boxpool_s1_applicable_nhwc(ft): continue
        assert np.array_equal(ops.boxpool_s1_nhwc(ft, kh, kw).cpu().numpy(), O.boxpool_s1(f, kh, kw)), ("pool", B, C, H, W, kh, kw)
        n["pool"] += 1
    elif kind == 8:    # channels-last best location / top-k / gather vs the NCHW kernels (themselves pinned against the oracle)
        B = int(rng.integers(1, 5)); K = int(rng.integers(1, 500)); Hp = int(rng.integers(1, 12)); Wp = int(rng.integers(1, 12)); k = int(rng.integers(1, 9)); C = 4 * int(rng.integers(1, 20)); fs = int(rng.integers(1, 5))
        cls = rng.standard_normal((B, K, Hp, Wp), dtype=np.float32)
        if Hp * Wp > 3: cls[:, :, Hp - 1, Wp - 1] = cls[:, :, 0, 0]
        ct = dev(cls).contiguous(memory_format=torch.channels_last)
        if not ops._is_nhwc(ct): continue
        d1, l1 = ops.best_location_desc(ct); d0, l0 = ops.best_location_desc(dev(cls))
        assert torch.equal(l1, l0) and torch.equal(d1, d0), ("bestloc", B, K, Hp, Wp)
        i1, s1 = ops.region_topk(ct, k); i0, s0 = ops.region_topk(dev(cls), k)
        assert torch.equal(i1, i0) and torch.equal(s1, s0), ("region_topk", B, K, Hp, Wp, k)
        fm = rng.standard_normal((B, C, Hp + fs - 1, Wp + fs - 1), dtype=np.float32); sh = rng.standard_normal(C * fs * fs).astype(np.float32) * 0.05
        fmt = dev(fm).contiguous(memory_format=torch.channels_last)
        if ops._is_nhwc(fmt):
            r1 = ops.region_gather_l2_nhwc(fmt, fs, fs, i1, Wp, dev(np.ascontiguousarray(sh.reshape(C, fs, fs).transpose(1, 2, 0)).reshape(-1)))
            r0 = ops.region_gather_l2(dev(fm), fs, fs, i0, Wp, dev(sh)).view(B, k, C, fs, fs).permute(0, 1, 3, 4, 2).reshape(B, k, -1)
            assert torch.allclose(r1, r0, rtol=2e-6, atol=1e-7), ("gather", B, C, Hp, Wp, fs, k)
        n["region"] += 1
    elif kind == 9:    # DBA groups kernel vs oracle
        N = int(rng.integers(1, 1500)); D = int(rng.choice([3, 8, 30, 64, 256])); L = int(rng.integers(1, 60)); k = int(rng.integers(-1, 8))
        E = rng.standard_normal((N, D)).astype(np.float32); E /= np.linalg.norm(E, axis=1, keepdims=True); labs = rng.integers(0, L, N).astype(np.int32)
        sys.path.insert(0, "/root/repo/instance-search_amd")
        from test.instance_avg import instance_avg
        got, _ = instance_avg(0, dev(E), [(None, int(l), None) for l in labs], None, k)
        assert np.allclose(got.cpu().numpy(), O.dba(E, labs, k), rtol=2e-6, atol=2e-7), ("dba", N, D, L, k)
        n["dba"] += 1
    else:              # score GEMM incl. the few-row split along the gallery and D % 32 != 0, vs the oracle on sampled rows
        M = int(rng.integers(1, 1200)); N = int(rng.integers(1000, 150000)); D = 4 * int(rng.integers(1, 130))
        if M * N * D > 3e11: continue
        Q = torch.randn(M, D, device="cuda", generator=g); G = torch.randn(N, D, device="cuda", generator=g)
        sim = ops.cosine_sim(Q, G)
        rows = rng.integers(0, M, 2); cols = np.r_[0:50, N - 200:N]
        want = O.cosine_sim(Q[rows].cpu().numpy(), G[cols].cpu().numpy()) if hasattr(O, "cosine_sim") else None
        if want is not None:
            assert np.array_equal(sim[rows][:, cols].cpu().numpy(), want), ("gemm", M, N, D)
        n["gemm"] += 1
torch.cuda.synchronize()
print("soak OK", n, "in %.0f s" % (time.time() - t0))
