import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'instance-search_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from isx import backbones
from foldbn import fold
net = backbones.resnet50(pretrained=True).cuda().eval()
feats = torch.nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool, *net.layer1, *net.layer2, *net.layer3, *net.layer4)
ff = fold(feats).to(memory_format=torch.channels_last)
x = torch.randn(512, 3, 224, 224, device="cuda").to(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3): ff(x)
    torch.cuda.synchronize()
    print("MARK_START", flush=True)
    for _ in range(3): ff(x)
    torch.cuda.synchronize()
