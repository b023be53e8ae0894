import sys, os
sys.path.insert(0, 'instance-search_amd'); sys.path.insert(0, 'oracle')
import torch, numpy as np
from isx import ops
import oracle as O
g = torch.Generator(device="cuda").manual_seed(0)
M, N, D, k = 1000, 100000, 2048, 100
G = torch.randn(N, D, device="cuda", generator=g)
Q = torch.randn(M, D, device="cuda", generator=g)
G, Q = ops.l2norm_rows(G), ops.l2norm_rows(Q)
ts, ti = ops.cosine_topk(Q, G, k)
ts_b, ti_b = ops.cosine_topk(Q, G, k)
print("repeat equal", torch.equal(ti, ti_b), torch.equal(ts, ts_b))
rows = [0, 1, 499, 999]
sub = ops.cosine_sim(Q[rows], G)
full = ops.cosine_sim(Q, G)
print("sim rows equal", torch.equal(sub, full[rows]))
s2, i2 = ops.topk_rows(sub, k)
s3, i3 = ops.topk_rows(full, k)
print("topk_rows(full) vs cosine_topk", torch.equal(i3, ti), torch.equal(s3, ts))
print("topk_rows(sub) vs cosine_topk rows", torch.equal(i2, ti[rows]), torch.equal(s2, ts[rows]))
# torch reference
tt = torch.topk(full, k, dim=1)
print("vs torch.topk idx mismatch count", int((tt.indices != ti).sum()), "score mismatch", int((tt.values != ts).sum()))
print("sub vs torch:", int((torch.topk(sub, k, dim=1).indices != i2).sum()))
d = (i2 != ti[rows]).nonzero()
print(d[:10], i2[d[:5,0], d[:5,1]], ti[rows][d[:5,0], d[:5,1]])
bad = (tt.indices != ti).nonzero()
print(bad[:10])
