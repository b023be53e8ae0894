"""Which vendor kernels does torch.mm pick on the shapes our GEMMs are compared with?  Run under
`rocprofv3 --kernel-trace --stats` (kernel names carry the Tensile tile configuration) and plain (timings)."""
import sys, torch
dev = "cuda"
def rnd(n, d, seed, dt=torch.float32):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return (x / x.norm(dim=1, keepdim=True)).to(dt)
def bench(fn, it):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / it)
    return best
for dt in (torch.float32, torch.float16):
    for M, N, D in ((10000, 32768, 2048), (1024, 10000, 2048), (200704, 256, 64), (200704, 512, 1024), (50176, 2048, 512)):
        Q, G = rnd(M, D, 1, dt), rnd(N, D, 2, dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        ms = bench(lambda: torch.mm(Q, G.t(), out=out), 5 if M * N > 1e8 else 20)
        print("torch.mm %s %6d x %6d x %d: %.3f ms %.1f TF" % (str(dt)[6:], M, N, D, ms, 2.0 * M * N * D / ms * 1e-9), flush=True)
