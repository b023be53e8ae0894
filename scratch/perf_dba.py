import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from test import instance_avg as ia
N, D, L = 10000, 2048, 1000
g = torch.Generator(device="cuda").manual_seed(0)
E = ops.l2norm_rows(torch.randn(N, D, device="cuda", generator=g))
ds = [(None, "l%d" % (i % L), "p%d" % i) for i in range(N)]
import inspect
print([n for n, f in inspect.getmembers(ia, inspect.isfunction)])
fn = getattr(ia, "instance_avg", None) or getattr(ia, "instance_average", None)
labels = sorted(set(l for _, l, _ in ds))
out = fn(0, E, ds, labels)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): out = fn(0, E, ds, labels)
torch.cuda.synchronize()
print(f"DBA over {N} x {D} descriptors, {L} labels: {(time.perf_counter()-t)/3*1e3:.1f} ms")
