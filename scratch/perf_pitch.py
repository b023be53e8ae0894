import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops, _lib
lib = _lib.lib()
def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M, N = 10000, 32768
out = torch.empty(M, N, device="cuda")
for D in (2048, 2080, 2112, 1984, 4096, 4128):
    Q = torch.randn(M, D, device="cuda"); G = torch.randn(N, D, device="cuda")
    for v in (1, 2):
        lib.isx_debug_set_gemm_variant(v)
        ms = timeit(lambda: ops.cosine_sim(Q, G, out=out))
        print(f"D={D} v{v}: {ms:.3f} ms {2*M*N*D/ms/1e9:.1f} TF")
    ms = timeit(lambda: torch.mm(Q, G.t(), out=out))
    print(f"D={D} torch.mm: {ms:.3f} ms {2*M*N*D/ms/1e9:.1f} TF")
