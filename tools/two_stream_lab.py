"""Lab: does running the trunk as two half batches on two HIP streams hide the tails of its 40-odd launches?
One pass over 1024 resident images: (a) one stream, one batch; (b) two streams, 512 images each; (c) two streams, 4 x 256.
Prints ms per pass (median of 7) and checks the feature maps are identical."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("LAB_B", "1024"))
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
from utils.dataset import synthetic_images  # noqa: E402
img = synthetic_images(64, seed=1234).to(dev).repeat(B // 64, 1, 1, 1).contiguous(memory_format=torch.channels_last)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def one():
    with torch.no_grad():
        return [net.features(img)]


def split(parts):
    outs = [None] * parts
    n = B // parts
    main = torch.cuda.current_stream()
    start = torch.cuda.Event(); start.record()
    with torch.no_grad():
        for p in range(parts):
            s = streams[p % 2]
            s.wait_event(start)
            with torch.cuda.stream(s):
                outs[p] = net.features(img[p * n:(p + 1) * n])
    for s in streams:
        main.wait_stream(s)
    return outs


def timed(f, reps=7):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


ref = torch.cat(one())
for name, f in (("one stream x %d" % B, one), ("two streams x %d" % (B // 2), lambda: split(2)), ("two streams, 4 x %d" % (B // 4), lambda: split(4)),
                ("one stream x %d again" % B, one)):
    out = torch.cat(f())
    torch.cuda.synchronize()
    med, mn = timed(f)
    print("%-28s median %.2f ms  min %.2f ms  identical %s" % (name, med, mn, torch.equal(out, ref)), flush=True)
