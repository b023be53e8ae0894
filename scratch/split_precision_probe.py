"""Probe for the split-precision trunk option (NOT built): C = A . B^T with (hi, lo) fp16 operand pairs,
three fp16-MFMA GEMMs, fp32 accumulate -- accuracy against fp64 and time against the fp32-MFMA GEMM, on a trunk-like shape."""
import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 200704, 512, 1024
A = torch.randn(M, K, device=dev, generator=g).relu()
B = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
def split(x):
    s = 2.0 ** (13 - torch.floor(torch.log2(x.abs().max())))
    hi = (x * s).half()
    lo = ((x * s - hi.float()) * 2048.0).half()
    return hi, lo, s
Ah, Al, sa = split(A); Bh, Bl, sb = split(B)
def emu():
    hh = ops.cosine_sim_f16(Ah, Bh)
    hl = ops.cosine_sim_f16(Ah, Bl)
    lh = ops.cosine_sim_f16(Al, Bh)
    return (hh + (hl + lh) * (1.0 / 2048.0)) * (1.0 / (sa * sb))
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
C32 = ops.cosine_sim(A, B)
Ce = emu()
rows = torch.arange(0, M, 997, device=dev)
ref = A[rows].double() @ B.double().t()
scale = ref.abs().mean()
e32 = ((C32[rows].double() - ref).abs().max() / scale).item()
ee = ((Ce[rows].double() - ref).abs().max() / scale).item()
e16 = (((ops.cosine_sim_f16(Ah, Bh)[rows].double() / (sa * sb).double()) - ref).abs().max() / scale).item()
fl = 2.0 * M * N * K
t32 = timeit(lambda: ops.cosine_sim(A, B)); te = timeit(emu); t1 = timeit(lambda: ops.cosine_sim_f16(Ah, Bh))
print(f"shape {M}x{N}x{K}: max |err| / mean|C|:  fp32 MFMA {e32:.2e}   split fp16 x3 {ee:.2e}   single fp16 pass {e16:.2e}")
print(f"time: fp32 MFMA {t32:.3f} ms ({fl/t32/1e9:.0f} TF)   3 fp16 passes + combine {te:.3f} ms ({fl/te/1e9:.0f} TF fp32-equivalent)   one fp16 pass {t1:.3f} ms")
