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
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(memory_format=torch.channels_last)
w = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
b = torch.randn(64, device=dev)
with torch.no_grad():
    t_old = timeit(lambda: ops.bias_relu_maxpool(conv(x), b))
    t_conv = timeit(lambda: conv(x))
    t_new = timeit(lambda: ops.stem7x7_pool(x, w, b))
    y0 = ops.bias_relu_maxpool(conv(x), b); y1 = ops.stem7x7_pool(x, w, b)
fl = 2.0 * B * 112 * 112 * 147 * 64
print(f"B={B}: miopen conv {t_conv:.3f} + pool = {t_old:.3f} ms | fused stem {t_new:.3f} ms = {fl/t_new/1e9:.1f} TF | maxdiff {(y0-y1).abs().max().item():.2e}")
