import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from isx import backbones, ops
from model.nn_utils import extract_layers, fold_batch_norm
net = backbones.resnet50(pretrained=True).eval()
feats, _, _ = extract_layers(net)
ff = fold_batch_norm(feats).cuda().to(memory_format=torch.channels_last)
x = torch.randn(512, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3): ff(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2): ff(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
