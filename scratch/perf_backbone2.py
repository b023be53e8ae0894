import sys, os, time, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
import torch
from torch.nn.utils.fusion import fuse_conv_bn_eval
from isx import backbones
def timeit(f, n=3, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
def fold(seq):
    """fold every Conv2d+BatchNorm2d pair of a ResNet trunk (eval mode)"""
    import torch.nn as nn
    mods = list(seq)
    out = []
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            out.append(fuse_conv_bn_eval(m, mods[i + 1])); i += 2; continue
        if isinstance(m, (backbones.Bottleneck, backbones.BasicBlock)):
            m = copy.deepcopy(m)
            m.conv1 = fuse_conv_bn_eval(m.conv1, m.bn1); m.bn1 = nn.Identity()
            m.conv2 = fuse_conv_bn_eval(m.conv2, m.bn2); m.bn2 = nn.Identity()
            if hasattr(m, 'conv3'):
                m.conv3 = fuse_conv_bn_eval(m.conv3, m.bn3); m.bn3 = nn.Identity()
            if m.downsample is not None:
                m.downsample = nn.Sequential(fuse_conv_bn_eval(m.downsample[0], m.downsample[1]))
        out.append(m); i += 1
    return nn.Sequential(*out)
net = backbones.resnet50(pretrained=True).cuda().eval()
feats = torch.nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool, *net.layer1, *net.layer2, *net.layer3, *net.layer4).to(memory_format=torch.channels_last)
B = 512
x = torch.randn(B, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    t = timeit(lambda: feats(x)); print(f"fp32 NHWC: {B/t:.0f} img/s")
    ff = fold(feats).to(memory_format=torch.channels_last)
    d = (ff(x[:8]) - feats(x[:8])).abs().max().item(); print("fold max abs diff", d, "ref max", feats(x[:8]).abs().max().item())
    t = timeit(lambda: ff(x)); print(f"fp32 NHWC BN-folded: {B/t:.0f} img/s")
    torch.backends.cudnn.benchmark = True
    t = timeit(lambda: ff(x), n=3, w=3); print(f"fp32 NHWC BN-folded + benchmark=True: {B/t:.0f} img/s")
    t = timeit(lambda: feats(x), n=3, w=3); print(f"fp32 NHWC + benchmark=True: {B/t:.0f} img/s")
