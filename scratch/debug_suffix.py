import copy, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_suffix import _blocks, _rel
from isx.suffix import SuffixEngine
for B in (8, 16, 24):
    which = "layer4"
    seq, cin, hw = _blocks(which)
    ref64 = copy.deepcopy(seq).double()
    eng = SuffixEngine(list(seq))
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.relu(torch.randn(B, cin, hw, hw, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    y = eng(x)
    r = torch.randn(y.shape, device="cuda", generator=g)
    (y * r).sum().backward()
    y64 = ref64(x.double())
    (y64 * r.double()).sum().backward()
    errs = [(_rel(p.grad.double(), q.grad), n) for (n, p), (_, q) in zip(seq.named_parameters(), ref64.named_parameters())]
    print(B, "out %.2e" % _rel(y.double(), y64), ["%s %.1e" % (n, e) for e, n in errs if e > 1e-5])
