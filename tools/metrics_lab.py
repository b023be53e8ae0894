"""Lab: the evaluation tail of a test main (utils.metrics.retrieval_metrics: similarity, P@1, mAP over the whole gallery) on synthetic unit-norm
descriptors at sizes where host-side loops show.  LAB_M queries x LAB_N gallery rows, LAB_L labels.  Prints wall time + the top of a cProfile."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402
from test import _common as C  # noqa: E402

M, N, L = (int(os.environ.get(k, d)) for k, d in (("LAB_M", "10000"), ("LAB_N", "100000"), ("LAB_L", "10000")))
g = torch.Generator(device="cuda").manual_seed(0)
G = ops.l2norm_rows(torch.randn((N, 2048), device="cuda", generator=g))
Q = ops.l2norm_rows(torch.randn((M, 2048), device="cuda", generator=g))
ref = [(None, "l%06d" % (i % L), "g%d" % i) for i in range(N)]
qry = [(None, "l%06d" % (i % L), "q%d" % i) for i in range(M)]
for rep in range(2):
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter(); pr.enable()
    m = C.retrieval_metrics(Q, G, qry, ref)
    pr.disable(); torch.cuda.synchronize()
    print("run %d: %.3f s  %s" % (rep, time.perf_counter() - t0, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in m.items()}), flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue()[:4000])
