"""A/B: ping-pong f16 filter GEMM, persistent (default) vs one tile per workgroup (debug tile cfg 3): whole exact-fast search + plain f16 GEMM."""
import sys, ctypes, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
force = lib().isx_debug_set_f16_tile
force.argtypes = [ctypes.c_int]
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
res = {}
for rep in range(3):
    for cfg in (-1, 3):
        force(cfg)
        ms = t(lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws))
        res[cfg] = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
        print("cfg %2d: fast search %.3f ms" % (cfg, ms), flush=True)
force(-1)
print("identical:", all(torch.equal(a, b) for a, b in zip(res[-1], res[3])))
