"""One 1x1 convolution shape, a few launches: the workload of `rocprofv3 --pmc ... -- python3 tools/one_conv.py` counter passes (LAB_H, LAB_CIN, LAB_COUT, LAB_RES)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402

B, H, Cin, Cout, res = 1024, int(os.environ.get("LAB_H", 28)), int(os.environ.get("LAB_CIN", 128)), int(os.environ.get("LAB_COUT", 512)), os.environ.get("LAB_RES", "1") == "1"
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.relu(torch.randn(B, Cin, H, H, device="cuda")))
w = torch.randn(Cout, Cin, 1, 1, device="cuda") * Cin ** -0.5
b = torch.randn(Cout, device="cuda")
r = cl(torch.randn(B, Cout, H, H, device="cuda")) if res else None
with torch.no_grad():
    for _ in range(4):
        ops.conv1x1_nhwc(x, w, b, r, True)
torch.cuda.synchronize()
