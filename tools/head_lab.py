"""Times the descriptor head's kernels at the reference shape (192 rows of a mini-batch, Linear(100352 -> 2048), reference model/siamese.py:104-114)
through the C ABI of the library named by ISX_LIB (default: the in-tree build): forward on rows, input gradient, fused weight gradient + SGD.
Prints ms, TFLOP/s and -- for the SGD kernel -- GB/s of its algorithmic bytes (w and momentum read + written).  For A/B builds (tools/build_variant.sh)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402
from isx._lib import check, lib  # noqa: E402

M, K, N = int(os.environ.get("LAB_M", 192)), int(os.environ.get("LAB_K", 100352)), int(os.environ.get("LAB_N", 2048))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g) * 0.01
w = torch.randn(N, K, device="cuda", generator=g) * 0.01
b = torch.randn(N, device="cuda", generator=g)
dy = torch.randn(M, N, device="cuda", generator=g) * 0.01
mom = torch.zeros_like(w)
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


Mp = (M + 63) // 64 * 64
dyT = dy.new_zeros((N, Mp)); dyT[:, :M] = dy.t()
dx = torch.empty(Mp, K, device="cuda")
fl = 2.0 * M * K * N
t_f = timeit(lambda: ops.head_linear(x, w, b))
t_d = timeit(lambda: check(lib().isx_head_linear_dgrad(dyT.data_ptr(), Mp, N, w.data_ptr(), K, dx.data_ptr(), st()), "dgrad"))
sgd = lambda: check(lib().isx_head_sgd_step(dy.data_ptr(), x.data_ptr(), M, N, K, w.data_ptr(), mom.data_ptr(), 0, 1e-6, 0.9, 0.0, 5e-4, 0, st()), "sgd")
t_s = timeit(sgd)
nbytes = 4.0 * N * K * 4
print("%s  fwd_rows %.3f ms %.1f TF | dgrad %.3f ms %.1f TF | sgd_step %.3f ms %.1f TF %.0f GB/s" %
      (os.environ.get("ISX_LIB", "in-tree"), t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9, t_s, fl / t_s / 1e9, nbytes / t_s / 1e6), flush=True)
