import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
def t(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for N in (32768, 32768 + 64, 32768 + 288, 16384, 16384 + 96):
    sim = torch.randn(10000, N, device="cuda")
    ms = t(lambda: ops.topk_rows(sim, 100)); ms2 = t(lambda: sim.max(dim=1))
    print("N=%d  topk_rows %.3f ms (%.2f TB/s)   torch max %.3f ms (%.2f TB/s)" % (N, ms, sim.numel() * 4e-9 / ms, ms2, sim.numel() * 4e-9 / ms2), flush=True)
