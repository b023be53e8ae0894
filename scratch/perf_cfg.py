import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import torch, numpy as np, ctypes
from isx import ops, _lib
import oracle as O
lib = _lib.lib()
def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
D = 2048
Qs = ops.l2norm_rows(torch.randn(130, 2048, device="cuda")); Gs = ops.l2norm_rows(torch.randn(257, 2048, device="cuda"))
want = O.cosine_sim(Qs.cpu().numpy(), Gs.cpu().numpy())
for c in (0, 4):
    lib.isx_debug_set_gemm_cfg(c)
    print("cfg", c, "bit-exact:", np.array_equal(ops.cosine_sim(Qs, Gs).cpu().numpy(), want))
for (M, N) in [(2048, 10000), (1000, 100000), (10000, 32768)]:
    Q = torch.randn(M, D, device="cuda"); G = torch.randn(N, D, device="cuda"); out = torch.empty(M, N, device="cuda")
    res = []
    for c in (0, 4):
        lib.isx_debug_set_gemm_cfg(c)
        ms = timeit(lambda: ops.cosine_sim(Q, G, out=out))
        res.append("%s %.3fms %.1fTF" % ("auto" if c < 0 else "c%d" % c, ms, 2*M*N*D/ms/1e9))
    ms = timeit(lambda: torch.mm(Q, G.t(), out=out))
    print(f"{M}x{N}: " + " | ".join(res) + " | torch.mm %.3fms %.1fTF" % (ms, 2*M*N*D/ms/1e9))
