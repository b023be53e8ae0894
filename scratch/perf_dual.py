import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(56, 64, 64, 256, 1), (56, 128, 256, 512, 2), (28, 256, 512, 1024, 2), (14, 512, 1024, 2048, 2)]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for H, K1, K2, Cout, s in shapes:
    Ho = (H - 1) // s + 1
    x = torch.relu(torch.randn(B, K2, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
    t = torch.relu(torch.randn(B, K1, Ho, Ho, device=dev)).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, K1 + K2, device=dev) * (K1 + K2) ** -0.5
    b = torch.randn(Cout, device=dev)
    res = []
    for cfg in (-1, 0, 2, 3, 7):
        lib().isx_debug_set_conv_cfg(cfg)
        res.append(timeit(lambda: ops.conv1x1_dual_nhwc(t, x, w, b, s, True)))
    lib().isx_debug_set_conv_cfg(-1)
    fl = 2.0 * B * Ho * Ho * (K1 + K2) * Cout
    print(f"H={H:3d} K={K1}+{K2} -> {Cout} s={s} | auto {res[0]:6.3f} cfg0 {res[1]:6.3f} cfg2 {res[2]:6.3f} cfg3 {res[3]:6.3f} notail {res[4]:6.3f} | best {fl/min(res)/1e9:6.1f} TF", flush=True)
