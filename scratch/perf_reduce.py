import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
dev = "cuda"
B = 1024
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for H, Cin, Cout in [(56, 256, 64), (56, 256, 128), (28, 512, 128)]:
    x = torch.relu(torch.randn(B, Cin, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5
    b = torch.randn(Cout, device=dev)
    res = {}
    for rep in range(2):
        for cfg in (-1, 0, 2, 3, 4):
            lib().isx_debug_set_gemm_cfg(cfg)
            t = timeit(lambda: ops.conv1x1_nhwc(x, w, b, None, True))
            res[cfg] = min(res.get(cfg, 1e9), t)
    lib().isx_debug_set_gemm_cfg(-1)
    byt = 4.0 * B * H * H * (Cin + Cout)
    print(f"H={H} {Cin}->{Cout}: " + " ".join(f"cfg{c} {t:.3f}" for c, t in res.items()) + f" | best {byt/min(res.values())/1e6:.0f} GB/s", flush=True)
