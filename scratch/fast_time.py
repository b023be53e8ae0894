import sys, time, torch
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
ws0 = torch.empty((ops.cosine_topk_workspace(M, N, D, k),), device=dev, dtype=torch.uint8)
ref = ops.cosine_topk(Q, G, k, ws=ws0)
for tile in (2, 1, 2, 1):
    lib().isx_debug_set_f16_tile(tile)
    fn = lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
    out = fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"tile={tile}: {dt*1e3:.2f} ms  {M*N/dt/1e9:.1f} G dist/s  equal={torch.equal(ref[1], out[1]) and torch.equal(ref[0], out[0])}", flush=True)
print("fallback rows:", lib().isx_debug_fast_fallback_rows(ws.data_ptr(), M, N, D, k, 1))
