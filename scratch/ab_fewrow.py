"""A/B: few-row score GEMM at M = 1000 vs 1024 and N = 100000 vs 98304 (same grid of 128x128 tiles), interleaved, after warm-up."""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
cases = [(1000, 100000, 2048), (1024, 98304, 2048), (1000, 98304, 2048), (1024, 100000, 2048), (1000, 100000, 464), (1024, 98304, 464)]
bufs = {}
for c in cases:
    M, N, D = c
    bufs[c] = (ops.l2norm_rows(torch.randn(M, D, device="cuda")), ops.l2norm_rows(torch.randn(N, D, device="cuda")), torch.empty(M, N, device="cuda"))
for _ in range(3):
    for c in cases: ops.cosine_sim(bufs[c][0], bufs[c][1], out=bufs[c][2])
torch.cuda.synchronize()
acc = {c: [] for c in cases}
for rep in range(8):
    for c in cases:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): ops.cosine_sim(bufs[c][0], bufs[c][1], out=bufs[c][2])
        b.record(); torch.cuda.synchronize()
        acc[c].append(a.elapsed_time(b) / 5)
for c in cases:
    M, N, D = c
    t = sorted(acc[c])[len(acc[c]) // 2]
    print(c, "%.3f ms  %.1f TF  %.3f of peak" % (t, 2e-9 * M * N * D / t, 2e-9 * M * N * D / t / 157.3))
