import sys, os
sys.path.insert(0, 'instance-search_amd'); sys.path.insert(0, 'oracle')
import torch, numpy as np
from isx import ops
g = torch.Generator(device="cuda").manual_seed(0)
M, N, D, k = 1000, 100000, 2048, 100
G = torch.randn(N, D, device="cuda", generator=g)
Q = torch.randn(M, D, device="cuda", generator=g)
G, Q = ops.l2norm_rows(G), ops.l2norm_rows(Q)
f1 = ops.cosine_sim(Q, G); f2 = ops.cosine_sim(Q, G)
print("gemm repeat equal", torch.equal(f1, f2))
ref = Q @ G.t()
print("gemm vs torch.mm maxdiff", float((f1 - ref).abs().max()))
bad = ((f1 - ref).abs() > 1e-5).nonzero()
print("n bad", bad.shape[0], bad[:8])
for t in range(3):
    s, i = ops.topk_rows(f1, k)
    tt = torch.topk(f1, k, dim=1)
    print("topk_rows vs torch.topk mismatches", int((tt.indices != i).sum()), int((tt.values != s).sum()))
    b = (tt.indices != i).nonzero()
    if b.shape[0]:
        print(" rows:", torch.unique(b[:, 0])[:20].tolist(), "first pos", b[0].tolist())
