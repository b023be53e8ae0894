import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for M, N, D in [(1024, 10000, 2048), (512, 10000, 2048), (2048, 10000, 2048), (1024, 125000, 2048), (10000, 32768, 2048), (256, 10000, 2048), (1000, 4096, 2048)]:
    Q = torch.randn(M, D, device="cuda"); G = torch.randn(N, D, device="cuda")
    res = []
    for rep in range(2):
        for i, cfg in enumerate((-1, 0, 1, 2, 3)):
            lib().isx_debug_set_gemm_cfg(cfg)
            t = timeit(lambda: ops.cosine_sim(Q, G), 20 if M * N < 5e7 else 5)
            if rep == 0: res.append(t)
            else: res[i] = min(res[i], t)
    lib().isx_debug_set_gemm_cfg(-1)
    fl = 2.0 * M * N * D
    print(f"{M}x{N}x{D}: auto {res[0]:.3f} | " + " ".join(f"cfg{c} {t:.3f} ({fl/t/1e9:.0f} TF)" for c, t in enumerate(res[1:])), flush=True)
