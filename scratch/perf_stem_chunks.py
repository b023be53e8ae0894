"""Stem (conv7x7/2 + bias + ReLU + maxpool) in batch chunks small enough for the convolution output to stay in the 256 MB Infinity
Cache between the convolution and the fused epilogue + pooling pass: does it save the 6.6 GB HBM round trip of the 3.3 GB map?"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/instance-search_amd")
import bench
from isx import ops
from isx._lib import lib, check
dev = torch.device("cuda", 0)
B = 1024
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
stem = list(net.features)[1]
img = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
def whole():
    with torch.no_grad():
        return stem(img)
def chunked(cs):
    out = torch.empty((B, 64, 56, 56), device=dev, memory_format=torch.channels_last)
    c = stem.cba
    with torch.no_grad():
        for i in range(0, B, cs):
            y = c.conv(img[i:i + cs])
            o = out[i:i + cs]
            check(lib().isx_bias_relu_maxpool_nhwc(y.data_ptr(), c.bias.data_ptr(), y.shape[0], 112, 112, 64, o.data_ptr(), torch.cuda.current_stream().cuda_stream), "pool")
    return out
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
ref = whole()
print("whole batch : %.3f ms" % timeit(whole))
for cs in (256, 128, 64, 32, 16):
    ms = timeit(lambda: chunked(cs))
    print("chunks of %3d: %.3f ms  identical=%s" % (cs, ms, torch.equal(chunked(cs), ref)), flush=True)
