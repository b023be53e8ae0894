"""Few launches of each hand-written kernel at bench shapes, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops
D = 2048
Q = ops.l2norm_rows(torch.randn(512, D, device="cuda")); G = ops.l2norm_rows(torch.randn(10000, D, device="cuda"))
sim = torch.empty(512, 10000, device="cuda")
for _ in range(3): ops.cosine_sim(Q, G, out=sim)
for _ in range(3): ops.topk_rows(sim, 100)
f = torch.randn(512, 2048, 7, 7, device="cuda").relu_(); y = torch.empty(512, 2048, device="cuda")
for _ in range(3): ops.gap_l2(f, out=y)
Qs = ops.l2norm_rows(torch.randn(10000, D, device="cuda")); Gs = ops.l2norm_rows(torch.randn(32768, D, device="cuda"))
big = torch.empty(10000, 32768, device="cuda")
for _ in range(2): ops.cosine_sim(Qs, Gs, out=big)
torch.cuda.synchronize()
