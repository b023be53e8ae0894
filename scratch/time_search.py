"""Wall clock of the shard search (10 000 x 125 000 x 2048, k = 100): exact-fast and exact fp32, plus isx_topk_rows at two shapes."""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
def unit(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)
def t(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device=dev, dtype=torch.uint8)
print("fast  10k x 125k x 2048 k=100: %.3f ms" % t(lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)), flush=True)
print("exact 10k x 125k x 2048 k=100: %.3f ms" % t(lambda: ops.cosine_topk(Q, G, k), n=3, w=1), flush=True)
a = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws); b = ops.cosine_topk(Q, G, k)
print("fast == exact:", all(torch.equal(x, y) for x, y in zip(a, b)))
sim = torch.randn(1000, 100000, device=dev)
print("topk_rows 1000 x 100000 k=100: %.3f ms" % t(lambda: ops.topk_rows(sim, 100)))
sim = torch.randn(10000, 32768, device=dev)
print("topk_rows 10000 x 32768 k=100: %.3f ms" % t(lambda: ops.topk_rows(sim, 100)))
