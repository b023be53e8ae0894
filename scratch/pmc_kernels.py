"""Few launches of each hand-written kernel at bench shapes, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from isx import ops
D = 2048
Q = ops.l2norm_rows(torch.randn(1024, D, device="cuda")); G = ops.l2norm_rows(torch.randn(10000, D, device="cuda"))
sim = torch.empty(1024, 10000, device="cuda")
for _ in range(3): ops.cosine_sim(Q, G, out=sim)
for _ in range(3): ops.topk_rows(sim, 100)
f = torch.randn(1024, 2048, 7, 7, device="cuda").relu_().contiguous(memory_format=torch.channels_last); y = torch.empty(1024, 2048, device="cuda")
for _ in range(3): ops.gap_l2(f, out=y)
del f
Qs = ops.l2norm_rows(torch.randn(10000, D, device="cuda")); Gs = ops.l2norm_rows(torch.randn(125000, D, device="cuda"))
big = torch.empty(10000, 32768, device="cuda")
for _ in range(2): ops.cosine_sim(Qs, Gs[:32768], out=big)
del big
gh = ops.gallery_to_f16(Gs)
for _ in range(2): ops.cosine_topk_fast(Qs, Gs, 100, gallery_f16=gh)
del Gs, gh
# trunk convolutions (B = 256 keeps the counter pass short): layer1 expanding conv with residual, layer3 reducing conv
x = torch.randn(256, 64, 56, 56, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
r = torch.randn(256, 256, 56, 56, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 64, device="cuda") * 0.1; b = torch.randn(256, device="cuda")
for _ in range(2): ops.conv1x1_nhwc(x, w, b, r, True)
x2 = torch.randn(256, 1024, 14, 14, device="cuda").relu_().contiguous(memory_format=torch.channels_last)
w2 = torch.randn(256, 1024, device="cuda") * 0.03; b2 = torch.randn(256, device="cuda")
for _ in range(2): ops.conv1x1_nhwc(x2, w2, b2, None, True)
torch.cuda.synchronize()
