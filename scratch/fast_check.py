"""GPU check of the exact-fast top-k: bit-identical to cosine_topk on random / clustered / adversarial data, timing."""
import sys, time, torch
sys.path.insert(0, "instance-search_amd")
from isx import ops
torch.manual_seed(0)
dev = "cuda"

def unit(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)

def check(name, Q, G, k, cached=True):
    ref = ops.cosine_topk(Q, G, k)
    gh = ops.gallery_to_f16(G) if cached else None
    got = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh)
    torch.cuda.synchronize()
    same_i = torch.equal(ref[1], got[1])
    same_s = torch.equal(ref[0].view(torch.int32), got[0].view(torch.int32))
    print(f"{name}: M={Q.shape[0]} N={G.shape[0]} D={Q.shape[1]} k={k} idx_equal={same_i} score_bits_equal={same_s}", flush=True)
    if not (same_i and same_s):
        bad = (ref[1] != got[1]).any(1).nonzero().flatten()
        print("   bad rows:", bad[:10].tolist(), "of", bad.numel())
        r = bad[0].item()
        print("   ref", ref[1][r][:12].tolist(), ref[0][r][:6].tolist())
        print("   got", got[1][r][:12].tolist(), got[0][r][:6].tolist())
    return same_i and same_s

ok = True
ok &= check("random", unit(300, 256, 1), unit(5000, 256, 2), 10)
ok &= check("random-own-gallery", unit(300, 256, 1), unit(5000, 256, 2), 10, cached=False)
ok &= check("random-k100", unit(1000, 2048, 3), unit(40000, 2048, 4), 100)
ok &= check("random-k1", unit(257, 128, 5), unit(33000, 128, 6), 1)
# relu-like non-negative descriptors
Q = torch.relu(torch.randn(500, 512, device=dev)); Q = Q / Q.norm(dim=1, keepdim=True)
G = torch.relu(torch.randn(20000, 512, device=dev)); G = G / G.norm(dim=1, keepdim=True)
ok &= check("relu", Q, G, 50)
# clustered: every gallery row a tiny perturbation of one of 8 centres -> windows overflow -> fallback
c = unit(8, 256, 7)
G = c[torch.randint(0, 8, (20000,), device=dev)] + 1e-4 * torch.randn(20000, 256, device=dev)
G = G / G.norm(dim=1, keepdim=True)
ok &= check("clustered(fallback)", unit(200, 256, 8), G, 20)
# duplicates: exact ties resolved by index
G = unit(3000, 128, 9).repeat(4, 1)
ok &= check("duplicates", unit(150, 128, 10), G, 16)
# mixed: half the queries near a dense cluster (fallback), half not
G = torch.cat([unit(10000, 256, 11), c[:1] + 1e-5 * torch.randn(2000, 256, device=dev)])
Q = torch.cat([unit(100, 256, 12), c[:1] + 1e-3 * torch.randn(100, 256, device=dev)])
ok &= check("mixed", Q, G, 30)
# un-normalised, large / small magnitudes
ok &= check("scaled-up", unit(100, 256, 13) * 300.0, unit(8000, 256, 14) * 1000.0, 10)
ok &= check("scaled-down", unit(100, 256, 13) * 1e-4, unit(8000, 256, 14) * 1e-3, 10)
ok &= check("huge(exact path)", unit(100, 256, 13) * 1e6, unit(8000, 256, 14), 10)
ok &= check("tiny values mixed", torch.cat([unit(100, 256, 15)[:, :128], 1e-7 * unit(100, 256, 16)[:, :128]], 1),
            torch.cat([unit(9000, 256, 17)[:, :128], 1e-7 * unit(9000, 256, 18)[:, :128]], 1), 10)
print("ALL OK" if ok else "FAILURES", flush=True)

# timing at the config-5 shard shape
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device=dev, dtype=torch.uint8)
ws0 = torch.empty((ops.cosine_topk_workspace(M, N, D, k),), device=dev, dtype=torch.uint8)
for name, fn in (("fp32 exact", lambda: ops.cosine_topk(Q, G, k, ws=ws0)), ("fast exact", lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"{name}: {dt*1e3:.2f} ms  {M*N/dt/1e9:.1f} G dist/s", flush=True)
a = ops.cosine_topk(Q, G, k, ws=ws0); b = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
print("big equal:", torch.equal(a[1], b[1]), torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)))
