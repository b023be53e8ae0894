"""Times isx_gap_l2 on the trunk output of the bench step (B x 2048 x 7 x 7, channels-last) through the library named by ISX_LIB: us, GB/s of
the algorithmic bytes (409 600 B per image), fraction of 8 TB/s.  For A/B builds (tools/build_variant.sh -DISX_GAP_UNROLL=.. -DISX_GAP_NT=1)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402


def timeit(f, n=50, w=5):
    for _ in range(w):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


out = []
for B in (256, 1024, 4096):
    f = torch.randn(B, 2048, 7, 7, device="cuda").relu_().to(memory_format=torch.channels_last)
    y = torch.empty(B, 2048, device="cuda")
    big = torch.empty(64 * 1024 * 1024, device="cuda")                     # 256 MiB written between launches: the map is not in the Infinity Cache when timed

    def run():
        big.fill_(1.0)
        ops.gap_l2(f, out=y)
    t_all = timeit(run, n=20)
    t_fill = timeit(lambda: big.fill_(1.0), n=20)
    t_hot = timeit(lambda: ops.gap_l2(f, out=y))
    nbytes = B * 2048 * 49 * 4 + B * 2048 * 4
    out.append("B=%d cold %.1f us %.0f GB/s (%.2f) | back to back %.1f us %.0f GB/s (%.2f)" %
               (B, (t_all - t_fill) * 1e3, nbytes / (t_all - t_fill) / 1e6, nbytes / (t_all - t_fill) / 8e9, t_hot * 1e3, nbytes / t_hot / 1e6, nbytes / t_hot / 8e9))
print(os.environ.get("ISX_LIB", "in-tree"), " || ".join(out), flush=True)
