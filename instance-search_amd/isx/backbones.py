"""Backbone definitions with torchvision-compatible state-dict keys.

The reference builds its nets from `torchvision.models.alexnet / resnet152(pretrained=True)`
(train/classif_finetune.py:113-121).  torchvision is not part of this image and there is
no network for pretrained weights, so the topologies are defined here with the same
attribute names (`conv1 bn1 relu maxpool layer1..4 avgpool fc` / `features classifier`),
hence the same state-dict keys: torchvision or reference-trained checkpoints load as is.
Convolutions run through PyTorch-ROCm (MIOpen); everything after the last conv is libisx.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        # the reference era's torchvision used a fixed AvgPool2d(7) (model/siamese.py:38-44 reads
        # its kernel_size), not an adaptive pool
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.flatten(1))


class AlexNet(nn.Module):
    """torchvision-era AlexNet: `features` (13 modules) + `classifier` (7 modules); same
    topology as the reference's own model/ModelDefinition.py:13-45."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 11, stride=4, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2))
        self.classifier = nn.Sequential(
            nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
            nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(self.features(x).flatten(1))


def _seeded(ctor, seed):
    """No pretrained weights exist offline: `pretrained=True` means seeded default init."""
    if seed is None:
        return ctor()
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        return ctor()


def alexnet(pretrained=False, seed=0, **kw):
    return _seeded(lambda: AlexNet(**kw), seed if pretrained else None)


def resnet18(pretrained=False, seed=0, **kw):
    return _seeded(lambda: ResNet(BasicBlock, [2, 2, 2, 2], **kw), seed if pretrained else None)


def resnet50(pretrained=False, seed=0, **kw):
    return _seeded(lambda: ResNet(Bottleneck, [3, 4, 6, 3], **kw), seed if pretrained else None)


def resnet152(pretrained=False, seed=0, **kw):
    return _seeded(lambda: ResNet(Bottleneck, [3, 8, 36, 3], **kw), seed if pretrained else None)


MODELS = {"alexnet": alexnet, "resnet18": resnet18, "resnet50": resnet50, "resnet152": resnet152}
