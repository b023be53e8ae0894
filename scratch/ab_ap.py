"""Sort-free AP at 1 k x 100 k with few / many positives per query (AP_MAXP A/B through ISX_LIB)."""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
def t(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M, N = 1000, 100000
sim = torch.rand(M, N, device="cuda")
for L in (10000, 1000, 250, 110):
    gl = (torch.arange(N, device="cuda") % L).int(); ql = (torch.arange(M, device="cuda") % L).int()
    print("positives per query %5d: %.3f ms" % (N // L, t(lambda: ops.average_precision_sim(sim, ql, gl), n=5 if N // L > 256 else 20)), flush=True)
