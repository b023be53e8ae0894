"""Does a depth-first schedule pay?  Stages 1-2 of the folded NHWC ResNet-50 trunk (activations of 0.8 - 3.3 GB per tensor at 1024 images)
run on sub-batches of S images, so that producer -> consumer tensors stay inside the 256-MiB Infinity Cache; stages 3-4 at the full batch."""
import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
mods = list(net.features)
x = torch.randn(1024, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
# flat list of leaf stages: find the first module whose output is 14 x 14
with torch.no_grad():
    y = x[:2]
    cut = None
    for i, m in enumerate(mods):
        y = m(y)
        print(i, type(m).__name__, tuple(y.shape), flush=True)
        if cut is None and y.shape[-1] == 14: cut = i
front, back = torch.nn.Sequential(*mods[:cut]), torch.nn.Sequential(*mods[cut:])
print("cut at module", cut)
def t(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    full = t(lambda: net.features(x))
    f_front = t(lambda: front(x))
    mid = front(x)
    f_back = t(lambda: back(mid))
    print("full batch: trunk %.2f ms = front %.2f + back %.2f" % (full, f_front, f_back), flush=True)
    ref = back(mid)
    for S in (32, 64, 128, 256, 512):
        def sub():
            return torch.cat([front(x[i:i + S]) for i in range(0, 1024, S)], 0)
        ms = t(sub)
        def sub_nocat():
            for i in range(0, 1024, S): front(x[i:i + S])
        ms2 = t(sub_nocat)
        same = torch.equal(sub(), mid)
        print("S=%4d: front in sub-batches %.2f ms (%.2f without the cat); identical %s" % (S, ms, ms2, same), flush=True)
