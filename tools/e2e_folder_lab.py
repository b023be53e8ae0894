"""Lab: one evaluation run of the reference's CLI surface (test/classif_finetune_test.main) on a FOLDER of JPEG files at a size where host-side
costs show: LAB_GALLERY gallery images + LAB_QUERIES queries over LAB_LABELS labels, written to a scratch folder first.  Prints the wall time of
the run, the result line, and the 25 most expensive host functions (cProfile, cumulative)."""
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

NG, NQ, NL = (int(os.environ.get(k, d)) for k, d in (("LAB_GALLERY", "20000"), ("LAB_QUERIES", "2000"), ("LAB_LABELS", "2000")))
MAIN, SIZE = os.environ.get("LAB_MAIN", "classif_finetune"), int(os.environ.get("LAB_SIZE", "224"))   # LAB_MAIN=classif_regions LAB_SIZE=448: the region path
tmp = tempfile.mkdtemp(prefix="isx_e2e_")
root = os.path.join(tmp, "CLICIDE_video_224sq" if SIZE == 224 else "CLICIDE_video_448")
os.makedirs(os.path.join(root, "test"))
os.makedirs(os.path.join(tmp, "data"))
for name in ("CLICIDE_224sq", "CLICIDE_448"):
    open(os.path.join(tmp, "data", name + "_train_ms.txt"), "w").write("0.485 0.456 0.406\n0.229 0.224 0.225\n")


def write(job):
    i, path = job
    low = np.random.default_rng(i % 4096).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    im = np.asarray(Image.fromarray(low).resize((SIZE, SIZE), Image.BICUBIC), dtype=np.int16)
    im = np.clip(im + np.random.default_rng(10 ** 6 + i).integers(-12, 13, im.shape), 0, 255).astype(np.uint8)
    Image.fromarray(im).save(path, quality=90)


jobs = [(i, os.path.join(root, "l%05d-%d.jpg" % (i % NL, i // NL))) for i in range(NG)]
jobs += [(NG + i, os.path.join(root, "test", "l%05d-q%d.jpg" % (i % NL, i // NL))) for i in range(NQ)]
from concurrent.futures import ProcessPoolExecutor  # noqa: E402
t0 = time.perf_counter()
with ProcessPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as pool:
    list(pool.map(write, jobs, chunksize=64))
print("wrote %d files in %.1f s" % (len(jobs), time.perf_counter() - t0), flush=True)
os.chdir(tmp)
from test import classif_finetune_test, classif_regions_test, siamese_descriptor_test, siamese_regions_test  # noqa: E402
try:
    for rep in range(2):                       # the second run is the one to read (kernels loaded, decoder processes started)
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr.enable()
        if MAIN == "classif_regions":
            res = classif_regions_test.main(root, "resnet50", "", 0, 0)
        elif MAIN == "siamese_descriptor":
            res = siamese_descriptor_test.main(root, "resnet50", "", 0, 2048, 64, 0)
        elif MAIN == "siamese_regions":
            res = siamese_regions_test.main(root, "resnet50", "", 0, 2048, 6, 0)
        else:
            res = classif_finetune_test.main(root, "resnet50", "", 0, False, 64, 0)
        pr.disable()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("run %d: %.2f s for %d gallery + %d query files (%.0f images/s end to end)  result %s" % (rep, dt, NG, NQ, (NG + NQ) / dt, res), flush=True)
        if rep == 0 and os.environ.get("LAB_COLD"):          # LAB_COLD=1: profile of the FIRST run (code objects, decoder start, allocations)
            break
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue()[:6000])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
