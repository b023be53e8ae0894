"""A/B timing of the fp32 GEMM family for two builds of libisx: ISX_LIB=/path/libisx.so python scratch/gemm_ab.py"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
def rnd(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(n, d, device=dev, generator=g) * 0.05
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
print("lib:", os.environ.get("ISX_LIB", "default"))
for M, N, D in ((10000, 32768, 2048), (1024, 10000, 2048), (10000, 8192, 2048), (4096, 10000, 2048)):
    Q, G = rnd(M, D, 1), rnd(N, D, 2)
    out = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.cosine_sim(Q, G, out=out), 5 if M * N > 1e8 else 30)
    print("cosine_sim %6d x %6d x %d: %.3f ms %.1f TF" % (M, N, D, ms, 2.0 * M * N * D / ms * 1e-9))
# trunk kernels on 2x2 tiles: dual conv layer 1..3, conv3x3 128->128 and 256->256
B = 256
for (H, K1, K2, Co, s) in ((56, 64, 64, 256, 1), (28, 128, 256, 512, 2), (14, 256, 512, 1024, 2)):
    Hi = H * s
    t = torch.relu(rnd(B * H * H, K1, 3)).view(B, H, H, K1).permute(0, 3, 1, 2)
    x = torch.relu(rnd(B * Hi * Hi, K2, 4)).view(B, Hi, Hi, K2).permute(0, 3, 1, 2)
    w = rnd(Co, K1 + K2, 5); bias = rnd(1, Co, 6).view(-1)
    ms = bench(lambda: ops.conv1x1_dual_nhwc(t, x, w, bias, s, True), 20)
    print("dual %dx%d K=%d+%d -> %d: %.3f ms %.1f TF" % (H, H, K1, K2, Co, ms, 2.0 * B * H * H * (K1 + K2) * Co / ms * 1e-9))
for (H, C) in ((28, 128), (14, 256)):
    x = torch.relu(rnd(B * H * H, C, 7)).view(B, H, H, C).permute(0, 3, 1, 2)
    w = rnd(C * 9, C, 8).view(C, 3, 3, C); bias = rnd(1, C, 9).view(-1)
    ms = bench(lambda: ops.conv3x3_nhwc(x, w, bias, 1, None, True), 20)
    print("conv3x3 %dx%d %d->%d: %.3f ms %.1f TF" % (H, H, C, C, ms, 18.0 * B * H * H * C * C / ms * 1e-9))
