import copy
import torch
from torch.nn.utils.fusion import fuse_conv_bn_eval
from isx import backbones
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
