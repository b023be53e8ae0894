import sys, os, time
sys.path.insert(0, 'instance-search_amd'); sys.path.insert(0, 'oracle')
import torch
from isx import ops

def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

g = torch.Generator(device="cuda").manual_seed(0)
D = 2048
def unit(n): return ops.l2norm_rows(torch.randn(n, D, device="cuda", generator=g))
for (M, N) in [(256, 10000), (1024, 10000), (4096, 16384), (10000, 32768)]:
    Q, G = unit(M), unit(N)
    out = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: ops.cosine_sim(Q, G, out=out))
    ms_t = timeit(lambda: torch.mm(Q, G.t(), out=out))
    print(f"cosine_sim {M}x{N}x{D}: {ms:.3f} ms  {2*M*N*D/ms/1e9:.1f} TF/s   | torch.mm {ms_t:.3f} ms {2*M*N*D/ms_t/1e9:.1f} TF/s")
M, N, k = 10000, 125000, 100
Q, G = unit(M), unit(N)
ws = torch.empty(ops.cosine_topk_workspace(M, N, D, k), dtype=torch.uint8, device="cuda")
ms = timeit(lambda: ops.cosine_topk(Q, G, k, ws=ws), n=3, w=1)
print(f"cosine_topk {M}x{N}x{D} k={k}: {ms:.2f} ms  {2*M*N*D/ms/1e9:.1f} TF/s  {M*N/ms/1e6:.1f} Gdist/s")
sim = torch.empty(M, 16384, device="cuda").normal_()
ms = timeit(lambda: ops.topk_rows(sim, k))
print(f"topk_rows first-chunk {M}x16384: {ms:.3f} ms  {M*16384*4/ms/1e6:.1f} GB/s")
for B in (256, 1024):
    f = torch.randn(B, 2048, 7, 7, device="cuda").relu_()
    y = torch.empty(B, 2048, device="cuda")
    ms = timeit(lambda: ops.gap_l2(f, out=y), n=20)
    print(f"gap_l2 B={B}: {ms*1000:.1f} us  {(B*2048*49*4+B*2048*4)/ms/1e6:.1f} GB/s")
x = torch.randn(100000, 2048, device="cuda"); y = torch.empty_like(x)
ms = timeit(lambda: ops.l2norm_rows(x, out=y), n=10)
print(f"l2norm_rows 100000x2048: {ms*1000:.1f} us  {2*x.numel()*4/ms/1e6:.1f} GB/s")
sim = torch.randn(1000, 10000, device="cuda")
ms = timeit(lambda: ops.rank_full(sim), n=3)
print(f"rank_full 1000x10000: {ms:.2f} ms")
