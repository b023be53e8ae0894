"""Micro-benchmark of the suffix engine (layer4 of ResNet-50, 24 images = one micro-batch): wall time vs GPU time per forward + backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch
import torch.nn as nn
from isx import backbones
from isx.suffix import SuffixEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
net = backbones.resnet50(pretrained=True, seed=0)
seq = nn.Sequential(*net.layer4).cuda().train()
for m in seq.modules():
    if isinstance(m, nn.BatchNorm2d):
        m.eval()
eng = SuffixEngine(list(seq))
x = torch.relu(torch.randn(B, 1024, 14, 14, device="cuda")).contiguous(memory_format=torch.channels_last)
r = torch.randn(B, 2048, 7, 7, device="cuda").contiguous(memory_format=torch.channels_last)
def step(f):
    y = f(x)
    y.backward(r)
for name, f in (("engine", eng), ("torch", seq)):
    for _ in range(5):
        step(f)
    torch.cuda.synchronize()
    n = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(n):
        step(f)
    b.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s B=%d: host enqueue %.3f ms, GPU %.3f ms, wall %.3f ms per fwd+bwd" % (name, B, 1e3 * t_host / n, a.elapsed_time(b) / n, 1e3 * t_all / n))
