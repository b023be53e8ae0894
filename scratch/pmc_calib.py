"""FETCH_SIZE calibration on a known byte count: M = 128 query rows (one m-tile) against 32768 gallery rows:
every G row is needed exactly once (268.4 MB), Q is 1 MB."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops, _lib
lib = _lib.lib()
D = 2048
Q = torch.randn(128, D, device="cuda"); G = torch.randn(32768, D, device="cuda"); out = torch.empty(128, 32768, device="cuda")
for c in (0, 3):
    lib.isx_debug_set_gemm_cfg(c)
    for _ in range(2): ops.cosine_sim(Q, G, out=out)
torch.cuda.synchronize()
