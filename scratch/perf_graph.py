"""Does capturing the trunk + pooling of one step in a HIP graph shorten the step? (inter-kernel gaps ~10 us x 52 launches)"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/instance-search_amd")
import bench
from isx import ops
dev = torch.device("cuda", 0)
B = 1024
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
x = torch.randn(B, 3, 224, 224, device=dev).to(memory_format=torch.channels_last)
q = torch.empty(B, 2048, device=dev)
def step():
    with torch.no_grad():
        ops.gap_l2(net.features(x), out=q)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
t_eager = timeit(step)
q_ref = q.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    step()
t_graph = timeit(g.replay)
print(f"eager {t_eager:.3f} ms  graph {t_graph:.3f} ms  same output {bool(torch.equal(q, q_ref))}")
t_eager2 = timeit(step)
print(f"eager again {t_eager2:.3f} ms")
