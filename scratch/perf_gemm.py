import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import torch, numpy as np
from isx import ops, _lib
lib = _lib.lib()
variants = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['1', '2'])]
def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
D = 2048
def unit(n): return ops.l2norm_rows(torch.randn(n, D, device="cuda", generator=g))
# correctness of each variant vs oracle on a small odd shape
import oracle as O
Qs, Gs = unit(130), unit(257)
want = O.cosine_sim(Qs.cpu().numpy(), Gs.cpu().numpy())
for v in variants:
    lib.isx_debug_set_gemm_variant(v)
    got = ops.cosine_sim(Qs, Gs).cpu().numpy()
    print("variant", v, "bit-exact vs oracle:", np.array_equal(got, want))
shapes = [(512, 10000), (1024, 10000), (4096, 16384), (10000, 32768)]
data = {s: (unit(s[0]), unit(s[1]), torch.empty(s[0], s[1], device="cuda")) for s in shapes}
for rnd in range(3):
    for (M, N) in shapes:
        Q, G, out = data[(M, N)]
        res = []
        for v in variants:
            lib.isx_debug_set_gemm_variant(v)
            ms = timeit(lambda: ops.cosine_sim(Q, G, out=out))
            res.append("v%d %.3f ms %.1f TF" % (v, ms, 2*M*N*D/ms/1e9))
        if rnd == 2:
            ms_t = timeit(lambda: torch.mm(Q, G.t(), out=out))
            res.append("torch.mm %.3f ms %.1f TF" % (ms_t, 2*M*N*D/ms_t/1e9))
        print(rnd, f"{M}x{N}:", " | ".join(res))
