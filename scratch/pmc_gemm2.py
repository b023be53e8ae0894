"""Workload for rocprofv3 --pmc passes: the shipped fp32 GEMM and hipBLASLt (torch.mm) on 10000 x 32768 x 2048."""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
g = torch.Generator(device="cuda").manual_seed(0)
Q = ops.l2norm_rows(torch.randn(10000, 2048, device="cuda", generator=g))
G = ops.l2norm_rows(torch.randn(32768, 2048, device="cuda", generator=g))
out = torch.empty(10000, 32768, device="cuda")
for _ in range(3):
    ops.cosine_sim(Q, G, out=out)
    torch.mm(Q, G.t(), out=out)
torch.cuda.synchronize()
