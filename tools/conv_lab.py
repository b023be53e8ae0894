"""Times the trunk's MFMA-bound convolution shapes (ResNet-50, B = 1024) through the C ABI of the library named by ISX_LIB (default: the in-tree
build): one line per shape and the total.  For A/B builds made with tools/build_variant.sh."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instance-search_amd"))
from isx import ops

B = int(os.environ.get("LAB_B", "1024"))
dev = "cuda"
one = [(56, 256, 64, False), (56, 256, 128, False), (28, 128, 512, True), (28, 512, 128, False), (28, 512, 256, False), (14, 256, 1024, True),
       (14, 1024, 256, False), (14, 1024, 512, False), (7, 512, 2048, True), (7, 2048, 512, False)]
three = [(28, 128, 1), (56, 128, 2), (14, 256, 1), (28, 256, 2), (7, 512, 1), (14, 512, 2)]
dual = [(28, 128, 256, 512, 2), (14, 256, 512, 1024, 2), (7, 512, 1024, 2048, 2)]      # (Ho, K1, K2, Cout, stride)


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n * 1e3)
    return best


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
tot = {"1x1": 0.0, "3x3": 0.0, "dual": 0.0}
fl = {"1x1": 0.0, "3x3": 0.0, "dual": 0.0}
with torch.no_grad():
    for H, Cin, Cout, res in one:
        x = cl(torch.relu(torch.randn(B, Cin, H, H, device=dev))); w = torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5
        b = torch.randn(Cout, device=dev); r = cl(torch.randn(B, Cout, H, H, device=dev)) if res else None
        t = timeit(lambda: ops.conv1x1_nhwc(x, w, b, r, True)); f = 2.0 * B * H * H * Cin * Cout
        tot["1x1"] += t; fl["1x1"] += f
        per = ""
        if os.environ.get("LAB_CFGS"):
            from isx._lib import lib
            for cfg in (0, 2, 3):
                lib().isx_debug_set_gemm_cfg(cfg)
                per += "  cfg%d %.3f" % (cfg, timeit(lambda: ops.conv1x1_nhwc(x, w, b, r, True)))
            lib().isx_debug_set_gemm_cfg(-1)
        print("1x1 H=%3d %5d->%5d res=%d %7.3f ms %6.1f TF%s" % (H, Cin, Cout, res, t, f / t / 1e9, per), flush=True)
    for H, C, s in three:
        x = cl(torch.relu(torch.randn(B, C, H, H, device=dev))); w = cl(torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5).permute(0, 2, 3, 1).contiguous()
        b = torch.randn(C, device=dev)
        t = timeit(lambda: ops.conv3x3_nhwc(x, w, b, s, None, True)); Ho = (H - 1) // s + 1; f = 18.0 * B * Ho * Ho * C * C
        tot["3x3"] += t; fl["3x3"] += f
        per = ""
        if os.environ.get("LAB_CFGS"):
            from isx._lib import lib
            for cfg in (0, 2, 3):
                lib().isx_debug_set_conv_cfg(cfg)
                per += "  cfg%d %.3f" % (cfg, timeit(lambda: ops.conv3x3_nhwc(x, w, b, s, None, True)))
            lib().isx_debug_set_conv_cfg(-1)
        print("3x3 H=%3d %5d s=%d        %7.3f ms %6.1f TF%s" % (H, C, s, t, f / t / 1e9, per), flush=True)
    for Ho, K1, K2, Cout, s in dual:
        tt = cl(torch.relu(torch.randn(B, K1, Ho, Ho, device=dev))); x = cl(torch.relu(torch.randn(B, K2, Ho * s, Ho * s, device=dev)))
        w = torch.randn(Cout, K1 + K2, device=dev) * (K1 + K2) ** -0.5; b = torch.randn(Cout, device=dev)
        t = timeit(lambda: ops.conv1x1_dual_nhwc(tt, x, w, b, s, True)); f = 2.0 * B * Ho * Ho * (K1 + K2) * Cout
        tot["dual"] += t; fl["dual"] += f
        per = ""
        if os.environ.get("LAB_CFGS"):
            from isx._lib import lib
            for cfg in (0, 2, 3):
                lib().isx_debug_set_conv_cfg(cfg)
                per += "  cfg%d %.3f" % (cfg, timeit(lambda: ops.conv1x1_dual_nhwc(tt, x, w, b, s, True)))
            lib().isx_debug_set_conv_cfg(-1)
        print("dual Ho=%3d %4d+%4d->%5d   %7.3f ms %6.1f TF%s" % (Ho, K1, K2, Cout, t, f / t / 1e9, per), flush=True)
print("TOTAL %s: " % os.environ.get("ISX_LIB", "in-tree") + "  ".join("%s %.2f ms (%.1f TF)" % (k, tot[k], fl[k] / tot[k] / 1e9) for k in tot), flush=True)
