import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
def unit(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device=dev, dtype=torch.uint8)
fn = lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
fn(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): fn()
torch.cuda.synchronize()
print(f"fast search {(time.perf_counter() - t) / 10 * 1e3:.3f} ms")
