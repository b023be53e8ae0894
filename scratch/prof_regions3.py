"""torch-profiler split of the two region paths at a chip-filling batch (448x448), and a ResNet-152 global extraction probe.
   python scratch/prof_regions3.py [B]"""
import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from torch.profiler import profile, ProfilerActivity
from isx import backbones, ops
from model.siamese import TuneClassifSub, RegionDescriptorNet, TuneClassif
from model.nn_utils import fold_batch_norm
from train import classif_regions as cr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
def cl(m):
    m = m.eval(); m.features = fold_batch_norm(m.features)
    return m.cuda().to(memory_format=torch.channels_last)
def timed(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
x = torch.randn(B, 3, 448, 448, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    sub = cl(TuneClassifSub(backbones.resnet50(pretrained=True), 464, (7, 7)))
    f1 = lambda: cr._best_location_descriptors(sub(x)[0])
    t = timed(f1); print("classif_regions B=%d: %.1f ms, %.0f images/s" % (B, t * 1e3, B / t))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        f1(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=90))
    del sub
    rd = cl(RegionDescriptorNet(backbones.resnet50(pretrained=True), 6, 2048, (7, 7)))
    f2 = lambda: rd(x)
    t = timed(f2); print("siamese_regions B=%d: %.1f ms, %.0f images/s" % (B, t * 1e3, B / t))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        f2(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=90))
    del rd, x
    torch.cuda.empty_cache()
    for name, bs in (("resnet152", 512), ("resnet50", 512)):
        g = cl(TuneClassif(backbones.MODELS[name](pretrained=True), 464))
        x2 = torch.randn(bs, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
        slab = torch.empty(bs, 2048, device="cuda")
        f3 = lambda: ops.gap_l2(g.features(x2), out=slab)
        t = timed(f3); print("%s global B=%d: %.1f ms, %.0f images/s" % (name, bs, t * 1e3, bs / t))
        if name == "resnet152":
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                f3(); torch.cuda.synchronize()
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=90))
        del g, x2
