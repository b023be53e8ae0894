"""Lab: the 1024-image trunk pass (49 launches) eager vs replayed from a HIP graph (torch.cuda.CUDAGraph capture of the same calls)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from isx import ops  # noqa: E402
from utils.dataset import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
B = 1024
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
img = synthetic_images(64, seed=1234).to(dev).repeat(B // 64, 1, 1, 1).contiguous(memory_format=torch.channels_last)
out = torch.empty((B, 2048), device=dev)


def eager():
    with torch.no_grad():
        ops.gap_l2(net.features(img), out=out)


def timed(f, reps=9):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


eager(); torch.cuda.synchronize()
ref = out.clone()
print("eager   median %.3f ms  min %.3f ms" % timed(eager), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eager()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eager()
torch.cuda.synchronize()
out.zero_()
g.replay(); torch.cuda.synchronize()
print("graph   median %.3f ms  min %.3f ms  identical %s" % (timed(g.replay) + (torch.equal(out, ref),)), flush=True)
print("eager   median %.3f ms  min %.3f ms" % timed(eager), flush=True)
