import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [  # (H, Cin, Cout, residual)
    (56, 64, 64, False), (56, 64, 256, True), (56, 256, 64, False), (56, 256, 128, False),
    (28, 128, 512, True), (28, 512, 128, False), (28, 512, 256, False), (28, 256, 512, False),
    (14, 256, 1024, True), (14, 1024, 256, False), (14, 1024, 512, False),
    (7, 512, 2048, True), (7, 2048, 512, False), (7, 1024, 2048, False),
]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
tot_a = tot_b = 0
for H, Cin, Cout, res in shapes:
    x = torch.relu(torch.randn(B, Cin, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5
    b = torch.randn(Cout, device=dev)
    r = torch.randn(B, Cout, H, H, device=dev).contiguous(memory_format=torch.channels_last) if res else None
    conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
    conv.weight.data.copy_(w)
    with torch.no_grad():
        ta = timeit(lambda: ops.bias_act_(conv(x), b, r, True))
        tc = timeit(lambda: conv(x))
        best = None
        per = {}
        for rep in range(2):
            for cfg in (-1, 0, 1, 2, 3):
                lib().isx_debug_set_gemm_cfg(cfg)
                tb = timeit(lambda: ops.conv1x1_nhwc(x, w, b, r, True))
                per[cfg] = min(per.get(cfg, 1e9), tb)
        tauto = per[-1]
        for cfg, tb in per.items():
            best = (tb, cfg) if best is None or tb < best[0] else best
        lib().isx_debug_set_gemm_cfg(-1)
        lib().isx_debug_set_conv_cfg(7)
        tgen = min(timeit(lambda: ops.conv1x1_nhwc(x, w, b, r, True)) for _ in range(2))
        lib().isx_debug_set_conv_cfg(-1)
    M = B * H * H
    fl = 2.0 * M * Cin * Cout
    byt = 4.0 * (M * Cin + M * Cout * (2 if res else 1))
    tot_a += ta; tot_b += tauto
    print(f"H={H:3d} {Cin:5d}->{Cout:5d} res={int(res)} | miopen+epi {ta:7.3f} ms (conv {tc:7.3f}) | isx auto {tauto:7.3f} ms best {best[0]:7.3f} (cfg {best[1]}) | {fl/tauto/1e9:6.1f} TF {byt/tauto/1e6:7.1f} GB/s | cfg0..3: " + " ".join("%.3f" % per[c] for c in (0, 1, 2, 3)) + " | no tail %.3f" % tgen, flush=True)
print(f"total miopen+epi {tot_a:.2f} ms, isx {tot_b:.2f} ms")
