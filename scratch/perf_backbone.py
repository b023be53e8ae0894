import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import backbones
def timeit(f, n=3, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
net = backbones.resnet50(pretrained=True).cuda().eval()
feats = torch.nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool, net.layer1, net.layer2, net.layer3, net.layer4)
for B in (256, 512, 1024):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad():
        t = timeit(lambda: feats(x)); print(f"fp32 NCHW B={B}: {B/t:.0f} img/s")
        xc = x.to(memory_format=torch.channels_last); fc = feats.to(memory_format=torch.channels_last)
        t = timeit(lambda: fc(xc)); print(f"fp32 NHWC B={B}: {B/t:.0f} img/s")
        feats.to(memory_format=torch.contiguous_format)
for B in (512,):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        t = timeit(lambda: feats(x)); print(f"bf16 autocast NCHW B={B}: {B/t:.0f} img/s")
        xc = x.to(memory_format=torch.channels_last); fc = feats.to(memory_format=torch.channels_last)
        t = timeit(lambda: fc(xc)); print(f"bf16 autocast NHWC B={B}: {B/t:.0f} img/s")
