import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = [(56, 64, 64, 1), (56, 128, 128, 2), (28, 128, 128, 1), (28, 256, 256, 2), (14, 256, 256, 1), (14, 512, 512, 2), (7, 512, 512, 1)]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for H, Cin, Cout, s in shapes:
    x = torch.relu(torch.randn(B, Cin, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, stride=s, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    w = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
    b = torch.randn(Cout, device=dev)
    with torch.no_grad():
        ta = timeit(lambda: ops.bias_act_(conv(x), b, None, True))
        tc = timeit(lambda: conv(x))
        res = []
        for cfg in (-1, 0, 2, 3, 7):
            lib().isx_debug_set_conv_cfg(cfg)
            res.append(timeit(lambda: ops.conv3x3_nhwc(x, w, b, s, None, True)))
        lib().isx_debug_set_conv_cfg(-1)
        y = ops.conv3x3_nhwc(x, w, b, s, None, True)
        ref = torch.relu(conv(x) + b.view(1, -1, 1, 1))
        err = (y - ref).abs().max().item()
    Ho = (H - 1) // s + 1
    fl = 2.0 * B * Ho * Ho * 9 * Cin * Cout
    print(f"H={H:3d} {Cin:4d}->{Cout:4d} s={s} | miopen+epi {ta:6.3f} (conv {tc:6.3f}) | isx auto {res[0]:6.3f} cfg0 {res[1]:6.3f} cfg2 {res[2]:6.3f} cfg3 {res[3]:6.3f} notail {res[4]:6.3f} | {fl/res[0]/1e9:6.1f} TF | maxerr {err:.2e}", flush=True)
