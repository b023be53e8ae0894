import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
from isx._lib import lib
import ctypes
dev = "cuda"; B, H = 1024, 56
M = B * H * H
x = torch.relu(torch.randn(M, 64, device=dev)); w = torch.randn(256, 64, device=dev) * 0.125; b = torch.randn(256, device=dev)
r = torch.randn(M, 256, device=dev); y = torch.empty(M, 256, device=dev)
L = lib()
f = L.isx_conv1x1_nhwc
def run(relu, res):
    st = torch.cuda.current_stream().cuda_stream
    rc = f(ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(M), 64, ctypes.c_void_p(w.data_ptr()), 256, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(r.data_ptr() if res else 0), relu, ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(st))
    assert rc == 0
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for relu, res, name in ((1, True, "full, residual"), (3, True, "no math"), (5, True, "no stores"), (9, True, "no res loads"), (13, True, "no stores, no res loads"), (7, True, "no math, no stores"), (1, False, "full, no residual arg")):
    ms = timeit(lambda: run(relu, res))
    byt = 4.0 * M * (64 + 256 * (2 if res else 1))
    print("%-22s %.3f ms  %.0f GB/s" % (name, ms, byt / ms / 1e6))
