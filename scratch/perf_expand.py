import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
H, Cin = 56, 64
x = torch.relu(torch.randn(B, Cin, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
w2 = (torch.randn(64, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).contiguous()
b2 = torch.randn(64, device=dev)
w3 = torch.randn(256, 64, device=dev) * 0.125
w3t = w3.t().contiguous()
b3 = torch.randn(256, device=dev)
r = torch.randn(B, 256, H, H, device=dev).contiguous(memory_format=torch.channels_last)
for rep in range(2):
    t_a = timeit(lambda: ops.conv3x3_nhwc(x, w2, b2, 1, None, True))
    mid = ops.conv3x3_nhwc(x, w2, b2, 1, None, True)
    t_b = timeit(lambda: ops.conv1x1_nhwc(mid, w3, b3, r, True))
    t_two = timeit(lambda: ops.conv1x1_nhwc(ops.conv3x3_nhwc(x, w2, b2, 1, None, True), w3, b3, r, True))
    t_f = timeit(lambda: ops.conv3x3_expand_nhwc(x, w2, b2, 1, w3t, b3, r, True))
    fl = 2.0 * B * H * H * (9 * Cin * 64 + 64 * 256)
    print(f"3x3 {t_a:.3f} + 1x1 {t_b:.3f} = back to back {t_two:.3f} ms | fused {t_f:.3f} ms = {fl/t_f/1e9:.1f} TF", flush=True)
y1 = ops.conv1x1_nhwc(ops.conv3x3_nhwc(x, w2, b2, 1, None, True), w3, b3, r, True)
y2 = ops.conv3x3_expand_nhwc(x, w2, b2, 1, w3t, b3, r, True)
print("identical", bool(torch.equal(y1, y2)))
# first block of the stage: 3x3 + fused projection GEMM vs the DUAL fused kernel
x2 = torch.relu(torch.randn(B, 64, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
wc = torch.randn(256, 128, device=dev) * 128 ** -0.5
wct = wc.t().contiguous()
for rep in range(2):
    t_two = timeit(lambda: ops.conv1x1_dual_nhwc(ops.conv3x3_nhwc(x, w2, b2, 1, None, True), x2, wc, b3, 1, True))
    t_f = timeit(lambda: ops.conv3x3_expand_dual_nhwc(x, w2, b2, x2, wct, b3, True))
    print(f"dual: back to back {t_two:.3f} ms | fused {t_f:.3f} ms", flush=True)
