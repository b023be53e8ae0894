import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
dev = "cuda"
def unit(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device=dev, dtype=torch.uint8)
for _ in range(3):
    ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
torch.cuda.synchronize()
print("fallback rows:", lib().isx_debug_fast_fallback_rows(ws.data_ptr(), M, N, D, k, 1))
