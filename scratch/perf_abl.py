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
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Q = torch.randn(M, D, device="cuda"); G = torch.randn(N, D, device="cuda"); out = torch.empty(M, N, device="cuda")
print("D =", D)
names = {3: "ws 1blk full", 11: "ws 1blk loads+stores no dependency", 12: "ws 1blk stores only", 13: "ws 1blk loads only"}
for rnd in range(2):
    for v in (3, 11, 12, 13):
        lib.isx_debug_set_gemm_variant(v)
        ms = timeit(lambda: ops.cosine_sim(Q, G, out=out))
        print(rnd, f"{names[v]:34s} {ms:.3f} ms {2*M*N*D/ms/1e9:.1f} TF")
