"""Does overlapping the HBM-bound front of the trunk (stem, layer 1, layer 2) of one half-batch with the MFMA-bound back (layer 3, layer 4)
of the other half-batch on a second HIP stream pay?  images/s of the fp32 NHWC folded ResNet-50 trunk + gap_l2, B = 1024."""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/instance-search_amd")
import bench
from isx import ops
from utils.dataset import synthetic_images
dev = torch.device("cuda", 0)
B = 1024
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
feats = list(net.features)
print("modules:", [type(m).__name__ for m in feats])
img = synthetic_images(64, seed=1234).to(dev).repeat(B // 64, 1, 1, 1).contiguous().to(memory_format=torch.channels_last)
q = torch.empty((B, 2048), device=dev)
def run(mods, x):
    for m in mods: x = m(x)
    return x
def single():
    with torch.no_grad():
        ops.gap_l2(run(feats, img), out=q)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def piped(split, parts=2):
    front, back = feats[:split], feats[split:]
    n = B // parts
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        prev_ev = None
        for p in range(parts):
            st = s1 if p % 2 == 0 else s2
            with torch.cuda.stream(st):
                if prev_ev is not None: st.wait_event(prev_ev)      # front(p) starts when front(p-1) is done, i.e. beside back(p-1)
                f = run(front, img[p * n:(p + 1) * n])
                prev_ev = torch.cuda.Event(); prev_ev.record(st)
                ops.gap_l2(run(back, f), out=q[p * n:(p + 1) * n])
    cur.wait_stream(s1); cur.wait_stream(s2)
def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
single(); ref = q.clone()
for rep in range(2):
    ms = timeit(single); print("single stream            : %.2f ms  %.0f images/s" % (ms, B / ms * 1e3), flush=True)
    for split in (9, 13):
        for parts in (2, 4):
            ms = timeit(lambda: piped(split, parts))
            ok = torch.equal(q, ref)
            print("split %2d, %d parts        : %.2f ms  %.0f images/s  identical=%s" % (split, parts, ms, B / ms * 1e3, ok), flush=True)
