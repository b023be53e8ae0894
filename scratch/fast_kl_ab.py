"""A/B of the candidate-list length KL of the exact-fast search (ISX_FAST_KL_PCT: KL = round32(pct * k / 100 + 32)): time, identity with the fp32 search, fallback rows."""
import sys, time, torch, ctypes
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
def unit(n, d, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, device="cuda", generator=g)
    return x / x.norm(dim=1, keepdim=True)
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ws = torch.empty((ops.cosine_topk_fast_workspace(M, N, D, k, True),), device="cuda", dtype=torch.uint8)
ws0 = torch.empty((ops.cosine_topk_workspace(M, N, D, k),), device="cuda", dtype=torch.uint8)
ref = ops.cosine_topk(Q, G, k, ws=ws0)
fn = lambda: ops.cosine_topk_fast(Q, G, k, gallery_f16=gh, ws=ws)
out = fn(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(8): out = fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 8
f = lib().isx_debug_fast_fallback_rows
f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
print("%.3f ms  equal=%s  fallback rows %d" % (dt * 1e3, torch.equal(ref[1], out[1]) and torch.equal(ref[0], out[0]), f(ws.data_ptr(), M, N, D, k, 1)))
