import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops
D = 2048; M, N, k = 10000, 125000, 100
Q = ops.l2norm_rows(torch.randn(M, D, device="cuda")); G = ops.l2norm_rows(torch.randn(N, D, device="cuda"))
ws = torch.empty(ops.cosine_topk_workspace(M, N, D, k), dtype=torch.uint8, device="cuda")
for _ in range(2): ops.cosine_topk(Q, G, k, ws=ws)
torch.cuda.synchronize()
