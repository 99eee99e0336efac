"""torchvision architectures the reference's loss networks are built on (TEST INFRASTRUCTURE).

The reference imports `torchvision.models.vgg16` (my_lpips/pretrained_networks.py:100, LPIPS-VGG) and `torchvision.models.resnet101`
(Loss/id_loss.py:3,13, the ArcFace identity network); torchvision is an un-vendored third-party dependency (requirements.txt:27 pins
torchvision==0.13.0) and is NOT installed in the build container.  This file restates the two PUBLISHED architectures as plain
torch.nn modules with torchvision's module / parameter names, so that (a) tools/refshim.py can hand them to the reference's own
PNetLin / IDLoss classes when it generates goldens and (b) the recorded state-dict specs carry the names a real checkpoint has.
    VGG16   "configuration D" of Simonyan & Zisserman: features = [64,64,M,128,128,M,256,256,256,M,512,512,512,M,512,512,512,M],
            every conv 3x3 pad 1 + ReLU(inplace), M = MaxPool2d(2, 2)                                 (torchvision/models/vgg.py)
    ResNet  v1.5 bottleneck network [3, 4, 23, 3]: 7x7/2 stem, BN, ReLU, MaxPool(3, 2, 1); Bottleneck = 1x1 -> 3x3 (carries the
            stride) -> 1x1 (x4) with BN after each, ReLU after the first two and after the residual add; 1x1/stride + BN
            downsample where the shape changes; global average pool; fc                             (torchvision/models/resnet.py)
Parity status of THIS file: unpinned against torchvision itself (absent here); the reference code that runs on top of it
(normalisation, lin layers, pooling, losses) is the reference's own and is what the goldens pin.
"""
import torch
import torch.nn as nn

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class VGG(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features


def vgg16(pretrained=False, **kw):
    layers, cin = [], 3
    for v in VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return VGG(nn.Sequential(*layers))


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

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class ResNet(nn.Module):
    def __init__(self, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        seq = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        seq += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet101(pretrained=False, num_classes=1000, **kw):
    return ResNet([3, 4, 23, 3], num_classes=num_classes)
