import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
def t(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for M in (10000, 2816):
    for N in (1024, 2048, 4096, 8192, 16384, 32768):
        sim = torch.randn(M, N, device="cuda")
        print("M=%d N=%d  k=100: %.3f ms   k=10: %.3f ms   sorted input k=100: %.3f ms" % (M, N, t(lambda: ops.topk_rows(sim, 100)), t(lambda: ops.topk_rows(sim, 10)),
              t(lambda s=sim.sort(dim=1, descending=True).values: ops.topk_rows(s, 100))), flush=True)
