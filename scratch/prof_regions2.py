import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from torch.profiler import profile, ProfilerActivity
from isx import backbones
from model.siamese import TuneClassifSub
from model.nn_utils import fold_batch_norm
from train import classif_regions as cr
torch.manual_seed(0)
sub = TuneClassifSub(backbones.resnet50(pretrained=True), 464, (7, 7)).eval()
sub.features = fold_batch_norm(sub.features)
sub = sub.cuda().to(memory_format=torch.channels_last)
x = torch.randn(64, 3, 448, 448, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(2): cr._best_location_descriptors(sub(x)[0])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2): cr._best_location_descriptors(sub(x)[0])
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
