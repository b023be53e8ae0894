import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops, _lib
lib = _lib.lib()
D = 2048
Qs = torch.randn(10000, D, device="cuda"); Gs = torch.randn(32768, D, device="cuda")
big = torch.empty(10000, 32768, device="cuda")
for v in (1, 2):
    pass
    for _ in range(2): ops.cosine_sim(Qs, Gs, out=big)
for _ in range(2): torch.mm(Qs, Gs.t(), out=big)
torch.cuda.synchronize()
