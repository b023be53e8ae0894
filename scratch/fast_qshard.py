import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
N, D, k = 1000000, 2048, 100
G = torch.empty(N, D, device=dev)
for i in range(0, N, 125000):
    G[i:i + 125000] = ops.l2norm_rows(torch.randn(125000, D, device=dev, generator=g))
gh = ops.gallery_to_f16(G)
for M in (1250, 2500, 5000, 10000):
    Q = ops.l2norm_rows(torch.randn(M, D, device=dev, generator=g))
    ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device=dev, dtype=torch.uint8)
    fn = lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"M={M}: {dt*1e3:.2f} ms  {M*N/dt/1e9:.1f} G dist/s", flush=True)
