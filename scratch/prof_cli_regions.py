import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo/instance-search_amd")
import torch
from test import classif_regions_test as T
spec = "synthetic:CLICIDE_video_224sq:n=600:q=60:labels=20:size=448"
T.main(spec, "resnet50", "", 0, 0)
torch.cuda.synchronize()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
T.main(spec, "resnet50", "", 0, 0)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t
print("second run wall %.2f s" % dt)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(24); print(s.getvalue()[:4200])
