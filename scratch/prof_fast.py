"""Workload for rocprofv3 --kernel-trace: exact-fast searches of the config-5 shard (10 000 x 125 000 x 2048, k = 100)."""
import sys, torch
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
for _ in range(4):
    out = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
torch.cuda.synchronize()
