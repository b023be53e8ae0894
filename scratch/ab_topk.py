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
for (M, N) in ((10000, 32768), (1000, 100000), (10000, 8192)):
    sim = torch.randn(M, N, device="cuda")
    print("topk_rows %d x %d k=100: %.3f ms" % (M, N, t(lambda: ops.topk_rows(sim, 100))), flush=True)
