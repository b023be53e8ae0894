"""`maxnet` / `Maxnet` / `copyParameters` -- the reference's own AlexNet clone
(model/ModelDefinition.py:13-73): `features` = 13 modules, `classifier` = 7 modules ending in
Linear(4096, nbClass), and a positional weight copy between two such nets."""
import torch.nn as nn

_CONVS = ((3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1))
_POOL_AFTER = (0, 1, 4)


class maxnet(nn.Module):
    def __init__(self, nbClass=464):
        super().__init__()
        feats = []
        for i, (cin, cout, k, s, p) in enumerate(_CONVS):
            feats += [nn.Conv2d(cin, cout, kernel_size=(k, k), stride=(s, s), padding=(p, p)), nn.ReLU(True)]
            if i in _POOL_AFTER:
                feats.append(nn.MaxPool2d((3, 3), stride=(2, 2), dilation=(1, 1)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(
            nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
            nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
            nn.Linear(4096, nbClass))

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))


def Maxnet(nbClass=464):
    return maxnet(nbClass)


def copyParameters(net, modelBase):
    """Position-wise copy: every Conv2d of `features`; every Linear of `classifier` whose
    weight shape matches (the last layer usually does not)."""
    for dst, src in zip(net.features, modelBase.features):
        if type(dst) is nn.Conv2d:
            dst.weight.data = src.weight.data
            dst.bias.data = src.bias.data
    for dst, src in zip(net.classifier, modelBase.classifier):
        if type(dst) is nn.Linear and dst.weight.size() == src.weight.size():
            dst.weight.data = src.weight.data
            dst.bias.data = src.bias.data
