import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops, _lib
lib = _lib.lib()
def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (256, 1024, 4096):
    f = torch.randn(B, 2048, 7, 7, device="cuda").relu_(); y = torch.empty(B, 2048, device="cuda")
    fc = f.to(memory_format=torch.channels_last)
    byt = B*2048*49*4 + B*2048*4
    res = []
    for kb in (13, 26, 52, 104):
        lib.isx_debug_set_gap_budget(kb * 1024)
        ms = timeit(lambda: ops.gap_l2(f, out=y)); res.append("NCHW %dKB %.1fus %.0fGB/s" % (kb, ms*1e3, byt/ms/1e6))
    ms = timeit(lambda: ops.gap_l2(fc, out=y)); res.append("NHWC %.1fus %.0fGB/s" % (ms*1e3, byt/ms/1e6))
    print(B, " | ".join(res))
