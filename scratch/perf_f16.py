import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops
def timeit(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
D = 2048
for (M, N) in [(130, 257), (1000, 10000), (10000, 32768)]:
    Q = ops.l2norm_rows(torch.randn(M, D, device="cuda")); G = ops.l2norm_rows(torch.randn(N, D, device="cuda"))
    Qh, qn, qa = ops.rows_to_f16(Q); Gh, gn, ga = ops.rows_to_f16(G)
    assert torch.equal(Qh, Q.half()) and torch.allclose(qn, (Q*Q).sum(1), rtol=1e-5)
    s16 = ops.cosine_sim_f16(Qh, Gh)
    ref = (Qh.double() @ Gh.double().t())
    exact = ops.cosine_sim(Q, G)
    print(f"{M}x{N}: |f16gemm - fp64(f16 inputs)| max {float((s16-ref).abs().max()):.2e}   |f16gemm - exact fp32| max {float((s16-exact).abs().max()):.2e}")
    out = torch.empty(M, N, device="cuda")
    ms = timeit(lambda: ops.cosine_sim_f16(Qh, Gh, out=out))
    ms_t = timeit(lambda: torch.mm(Qh, Gh.t()))
    ms_c = timeit(lambda: ops.rows_to_f16(G))
    print(f"   f16 gemm {ms:.3f} ms {2*M*N*D/ms/1e9:.0f} TF | torch.mm half {ms_t:.3f} ms {2*M*N*D/ms_t/1e9:.0f} TF | to_f16(G) {ms_c:.3f} ms")
