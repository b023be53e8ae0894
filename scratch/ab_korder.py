"""A/B of the conv3x3 k-tile visiting order (git apply scratch/conv3x3_korder.patch, build conv.hip + expand.hip with -DISX_CONV3X3_KORDER=1 into a second library, ISX_LIB=<that library> vs the default one): time per ResNet-50 3x3 layer at B = 1024,
max |diff| against torch's conv2d (sanity: the two orders are different fma chains)."""
import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
B = 1024
tot = 0.0
for (C, H, s) in ((256, 14, 1), (128, 28, 1), (512, 7, 1), (128, 56, 2), (256, 28, 2), (512, 14, 2)):
    g = torch.Generator(device="cuda").manual_seed(C + H)
    x = torch.relu(torch.randn(B, C, H, H, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, 3, 3, C, device="cuda", generator=g) * (9 * C) ** -0.5).contiguous()
    b = torch.randn(C, device="cuda", generator=g)
    ms = timeit(lambda: ops.conv3x3_nhwc(x, w, b, s, None, True))
    y = ops.conv3x3_nhwc(x[:8], w, b, s, None, True)
    ref = torch.relu(torch.nn.functional.conv2d(x[:8], w.permute(0, 3, 1, 2), b, stride=s, padding=1))
    Ho = (H - 1) // s + 1
    fl = 2.0 * B * Ho * Ho * 9 * C * C
    n_launch = {(256, 14, 1): 5, (128, 28, 1): 3, (512, 7, 1): 2}.get((C, H, s), 1)
    tot += ms * n_launch
    print("%4d ch @%2d s%d: %.3f ms  %.1f TF  max|d vs torch| %.2e  checksum %.6f" % (C, H, s, ms, fl / ms / 1e9, float((y - ref).abs().max()), float(y.double().sum())), flush=True)
print("3x3 family per step (13 launches): %.2f ms" % tot)
