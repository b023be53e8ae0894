"""Micro-benchmark of the head GEMMs (192 rows, K = 100352, N = 2048): forward (split-K), input gradient, weight gradient from rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_head import _fwd, _dgrad
from isx import dp
M, K, N = 192, 100352, 2048
x = torch.randn(M, K, device="cuda") * 0.01; w = torch.randn(N, K, device="cuda") * 0.01; b = torch.randn(N, device="cuda"); dy = torch.randn(M, N, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / n
print("fwd (incl. transpose + reduce) %.3f ms, dgrad (incl. transpose) %.3f ms, dW from rows %.3f ms; torch: linear %.3f, dy@w %.3f, dy.T@x %.3f"
      % (t(lambda: _fwd(x, w, b)), t(lambda: _dgrad(dy, w)), t(lambda: dp.weight_gradient_from_rows(dy, x)),
         t(lambda: torch.nn.functional.linear(x, w, b)), t(lambda: dy @ w), t(lambda: dy.t() @ x)))
