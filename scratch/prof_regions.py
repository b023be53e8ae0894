import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from torch.profiler import profile, ProfilerActivity
from isx import backbones
from model.siamese import RegionDescriptorNet
from model.nn_utils import fold_batch_norm
torch.manual_seed(0)
rd = RegionDescriptorNet(backbones.resnet50(pretrained=True), 6, 2048, (7, 7)).eval()
rd.features = fold_batch_norm(rd.features)
rd = rd.cuda().to(memory_format=torch.channels_last)
x = torch.randn(64, 3, 448, 448, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(2): rd(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2): rd(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=16, max_name_column_width=64))
