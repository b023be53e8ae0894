#!/usr/bin/env python3
"""How testable is "|dmAP| <= 1e-4" on a synthetic set?  GPU path only: for an architecture / image structure / set size, the score spread of the
descriptors, P@1 / mAP, and the number of (positive, negative) pairs adjacent in a ranked list whose scores lie within LAB_TIE (default 3e-5, about
2 x the fp32 paths' distance on ResNet-152) of each other in the first LAB_TOP ranks -- every such pair is a rank swap a different summation
order may make, and each moves that query's AP.  Used to size tests/test_gpu_end_to_end.py's ResNet-152 case (docs/rounds/r06.md)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "instance-search_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from test_gpu_end_to_end import _calibrated_weights
    from isx import ops
    from test import _common as C
    from test import classif_finetune_test as T
    from utils.metrics import _average_precisions
    arch = os.environ.get("LAB_ARCH", "resnet152")
    n, q, labels = int(os.environ.get("LAB_N", 1000)), int(os.environ.get("LAB_Q", 200)), int(os.environ.get("LAB_LABELS", 50))
    tie, top = float(os.environ.get("LAB_TIE", 3e-5)), int(os.environ.get("LAB_TOP", 50))
    for struct in [int(s) for s in os.environ.get("LAB_STRUCT", "70,85,92").split(",")]:
        w = _calibrated_weights("classif", labels, "/tmp/lab_w.pth", arch=arch)
        spec = "synthetic:CLICIDE_video_224sq:n=%d:q=%d:labels=%d:struct=%d" % (n, q, labels, struct)
        seen = []
        real = C.evaluate_retrieval
        C.evaluate_retrieval = lambda te, re_, ts, rs, *a, **k: (seen.append((te, re_, ts, rs)) or real(te, re_, ts, rs, *a, **k))
        try:
            torch.manual_seed(0)
            p1, mAP = T.main(spec, arch, w, 0, False, 64, 0)
        finally:
            C.evaluate_retrieval = real
        te, re_, ts, rs = seen[-1]
        sim = ops.cosine_sim(te.float(), re_.float())
        labs = sorted(set(l for _, l, _ in ts) | set(l for _, l, _ in rs))
        ql = torch.tensor([labs.index(l) for _, l, _ in ts]).cuda()
        gl = torch.tensor([labs.index(l) for _, l, _ in rs]).cuda()
        order = sim.sort(dim=1, descending=True, stable=True)
        sc, idx = order.values[:, :top + 1], order.indices[:, :top + 1]
        pos = gl[idx] == ql[:, None]
        near = (sc[:, :-1] - sc[:, 1:]) < tie
        crit = near & (pos[:, :-1] != pos[:, 1:])
        first = (sc[:, 0] - sc[:, 1]) < tie
        print("%s struct=%d %dx%d labels=%d: spread %.3g, P@1 %.4f, mAP %.6f | adjacent pos/neg pairs within %.1e in the top %d: %d (queries touched %d), "
              "top-1/top-2 within it: %d" % (arch, struct, q, n, labels, float(sim.max() - sim.min()), p1, mAP, tie, top, int(crit.sum()),
                                            int(crit.any(1).sum()), int(first.sum())), flush=True)


if __name__ == "__main__":
    main()
