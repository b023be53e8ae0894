"""Lab: the three kernels of the sharded average precision on one config-5 shard (10 k queries x 125 k rows, 10 positives per query and shard),
timed with HIP events: bytes of the score block over the kernel time against the HBM roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402

M, N, D = 8192, 125000, 2048
g = torch.Generator(device="cuda").manual_seed(0)
L = N // 10                                               # N / 10 instances, row i carries label i mod L, descriptor = normalise(centroid + sigma * noise)
SIGMA = float(os.environ.get("LAB_SIGMA", "4.0"))         # 4.0: the survey's discriminating setting (mAP ~ 0.5); 1e9: pure noise (every key above the smallest positive)
glab = (torch.arange(N, dtype=torch.int64) % L).to(torch.int32).cuda()
qlab = (torch.arange(M, dtype=torch.int64) % L).to(torch.int32).cuda()
cent = torch.randn(L, D, device="cuda", generator=g)
G = ops.l2norm_rows(cent[glab.long()] + SIGMA * torch.randn(N, D, device="cuda", generator=g))
Q = ops.l2norm_rows(cent[qlab.long()] + SIGMA * torch.randn(M, D, device="cuda", generator=g))
sim = ops.cosine_sim(Q, G)


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


gb = M * N * 4 / 1e9
t1, (keys, cnt) = timed(lambda: ops.ap_shard_positives(sim, 0, qlab, glab))
t2, hist = timed(lambda: ops.ap_shard_hist(sim, 0, keys))
t3, ap = timed(lambda: ops.ap_from_hist(hist, cnt))
tu, apu = timed(lambda: ops.average_precision_sim(sim, qlab, glab))
print("score block %.2f GB; positives per query %d; sigma %g; mAP %.3f" % (gb, int(cnt.max()), SIGMA, float(apu[apu == apu].mean())))
print("isx_ap_shard_positives %.3f ms (reads the labels per row + the positives' scores; writes %d MB of key slots)" % (t1, keys.numel() * 8 >> 20))
print("isx_ap_shard_hist      %.3f ms = %.2f TB/s of score rows" % (t2, gb / t2))
print("isx_ap_from_hist       %.3f ms" % t3)
print("isx_average_precision_sim (unsharded, one kernel) %.3f ms = %.2f TB/s; identical %s" % (tu, gb / tu, torch.equal(torch.nan_to_num(ap, nan=-7.), torch.nan_to_num(apu, nan=-7.))))
