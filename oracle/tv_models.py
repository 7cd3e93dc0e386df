"""Plain ``torch.nn`` CPU restatement of the torchvision-0.8.2 graphs the reference instantiates.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Follows the call sites
``/root/reference/neuston_models.py:22-45`` (``get_namebrand_model``): torchvision's
``inception_v3`` with ``fc`` and ``AuxLogits.fc`` replaced for ``num_o_classes`` (:23-26) and
``resnet*`` with ``fc`` replaced (:37-39).  torchvision itself is NOT in /root/reference (pinned at
``requirements/env.hpc.yml:9``) and not installed; the graph below restates its published
architecture and is pinned by the published parameter totals (27,161,264 / 11,689,512 at 1000
classes) and by ``tests/golden/*_keys.json``.

``storage='bf16'`` inserts straight-through bf16 rounding (forward AND backward) at exactly the
points where the HIP path stores a tensor in HBM as bf16, so the two can be compared to tight
tolerance; ``storage='fp32'`` is the reference's own fp32 arithmetic.
"""
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

InceptionOutputs = namedtuple('InceptionOutputs', ['logits', 'aux_logits'])


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class Storage:
    """Rounding policy shared by every module of one oracle model."""

    def __init__(self, kind='fp32'):
        assert kind in ('fp32', 'bf16')
        self.kind = kind

    def act(self, x):
        return _RoundBF16.apply(x) if self.kind == 'bf16' else x

    def weight(self, w):
        # the HIP path multiplies bf16 shadow weights; gradients flow to the fp32 master unrounded
        if self.kind == 'bf16':
            return w + (w.detach().to(torch.bfloat16).to(torch.float32) - w.detach())
        return w


class BasicConv2d(nn.Module):
    """conv(bias=False) -> BatchNorm2d(eps=1e-3) -> ReLU  ([TV] inception.BasicConv2d)."""

    def __init__(self, st, cin, cout, **kw):
        super().__init__()
        self.st = st
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        c = self.conv
        y = F.conv2d(x, self.st.weight(c.weight), None, c.stride, c.padding)
        y = self.st.act(y)                      # raw conv output is stored
        y = F.relu(self.bn(y))
        return self.st.act(y)                   # activated output is stored


def _pool(st, y):
    return st.act(y)


class InceptionA(nn.Module):
    def __init__(self, st, cin, pool_features):
        super().__init__()
        self.st = st
        self.branch1x1 = BasicConv2d(st, cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(st, cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(st, 48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(st, cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(st, 64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(st, 96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(st, cin, pool_features, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(_pool(self.st, F.avg_pool2d(x, 3, 1, 1)))
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, st, cin):
        super().__init__()
        self.st = st
        self.branch3x3 = BasicConv2d(st, cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(st, cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(st, 64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(st, 96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = F.max_pool2d(x, 3, 2)
        return torch.cat([b3, bd, bp], 1)


class InceptionC(nn.Module):
    def __init__(self, st, cin, c7):
        super().__init__()
        self.st = st
        self.branch1x1 = BasicConv2d(st, cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(st, cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(st, c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(st, c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(st, cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(st, c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(st, c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(st, c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(st, c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(st, cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_1(x)
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(bd))))
        bp = self.branch_pool(_pool(self.st, F.avg_pool2d(x, 3, 1, 1)))
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, st, cin):
        super().__init__()
        self.st = st
        self.branch3x3_1 = BasicConv2d(st, cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(st, 192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(st, cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(st, 192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(st, 192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(st, 192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        bp = F.max_pool2d(x, 3, 2)
        return torch.cat([b3, b7, bp], 1)


class InceptionE(nn.Module):
    def __init__(self, st, cin):
        super().__init__()
        self.st = st
        self.branch1x1 = BasicConv2d(st, cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(st, cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(st, 384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(st, 384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(st, cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(st, 448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(st, 384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(st, 384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(st, cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bp = self.branch_pool(_pool(self.st, F.avg_pool2d(x, 3, 1, 1)))
        return torch.cat([b1, b3, bd, bp], 1)


class InceptionAux(nn.Module):
    def __init__(self, st, cin, num_classes):
        super().__init__()
        self.st = st
        self.conv0 = BasicConv2d(st, cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(st, 128, 768, kernel_size=5)
        self.conv1.stddev = 0.01
        self.fc = nn.Linear(768, num_classes)
        self.fc.stddev = 0.001

    def forward(self, x):
        x = _pool(self.st, F.avg_pool2d(x, 5, 3))
        x = self.conv1(self.conv0(x))
        x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
        return self.fc(x)


class Inception3(nn.Module):
    """[TV] ``torchvision.models.inception.Inception3`` (aux_logits=True).

    ``dropout_mask``: optional ``[B,2048]`` tensor of 0/1 keep flags; when given, the train-mode
    dropout is ``x * mask * 2`` (p=0.5) so the HIP path and the oracle see the same Bernoulli draw.
    """

    def __init__(self, num_classes=1000, transform_input=False, storage='fp32'):
        super().__init__()
        st = self.st = Storage(storage)
        self.transform_input = transform_input
        self.Conv2d_1a_3x3 = BasicConv2d(st, 3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(st, 32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(st, 32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(st, 64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(st, 80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(st, 192, 32)
        self.Mixed_5c = InceptionA(st, 256, 64)
        self.Mixed_5d = InceptionA(st, 288, 64)
        self.Mixed_6a = InceptionB(st, 288)
        self.Mixed_6b = InceptionC(st, 768, 128)
        self.Mixed_6c = InceptionC(st, 768, 160)
        self.Mixed_6d = InceptionC(st, 768, 160)
        self.Mixed_6e = InceptionC(st, 768, 192)
        self.AuxLogits = InceptionAux(st, 768, num_classes)
        self.Mixed_7a = InceptionD(st, 768)
        self.Mixed_7b = InceptionE(st, 1280)
        self.Mixed_7c = InceptionE(st, 2048)
        self.fc = nn.Linear(2048, num_classes)
        self.dropout_mask = None
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                std = getattr(m, 'stddev', 0.1)
                nn.init.trunc_normal_(m.weight, 0.0, std, -2 * std, 2 * std)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _transform_input(self, x):
        if self.transform_input:
            c0 = x[:, 0:1] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
            c1 = x[:, 1:2] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
            c2 = x[:, 2:3] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
            x = torch.cat((c0, c1, c2), 1)
        return x

    def forward(self, x):
        st = self.st
        x = st.act(self._transform_input(x))
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, 3, 2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, 3, 2)
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6a(x)
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(x))))
        aux = self.AuxLogits(x) if self.training else None
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
        if self.training:
            if self.dropout_mask is not None:
                x = x * (self.dropout_mask.to(x.dtype) * 2.0)
            else:
                x = F.dropout(x, 0.5, True)
        x = self.fc(x)
        if self.training:
            return InceptionOutputs(x, aux)
        return x


# --------------------------------------------------------------------------------------------- resnet
class _ConvBN(nn.Module):
    """helper holding no parameters itself; resnet keeps conv/bn as siblings for key compatibility."""


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, st, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.st = st
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        st = self.st
        idt = x
        out = st.act(_cv(st, self.conv1, x))
        out = st.act(F.relu(self.bn1(out)))
        out = st.act(_cv(st, self.conv2, out))
        if self.downsample is not None:
            idt = st.act(_cv(st, self.downsample[0], x))
            idt = st.act(self.downsample[1](idt))
        out = F.relu(self.bn2(out) + idt)
        return st.act(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, st, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.st = st
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        st = self.st
        idt = x
        out = st.act(F.relu(self.bn1(st.act(_cv(st, self.conv1, x)))))
        out = st.act(F.relu(self.bn2(st.act(_cv(st, self.conv2, out)))))
        out = st.act(_cv(st, self.conv3, out))
        if self.downsample is not None:
            idt = st.act(_cv(st, self.downsample[0], x))
            idt = st.act(self.downsample[1](idt))
        return st.act(F.relu(self.bn3(out) + idt))


def _cv(st, conv, x):
    return F.conv2d(x, st.weight(conv.weight), None, conv.stride, conv.padding)


class ResNet(nn.Module):
    """[TV] ``torchvision.models.resnet.ResNet`` (BN eps 1e-5, no zero-init-residual)."""

    def __init__(self, block, layers, num_classes=1000, storage='fp32'):
        super().__init__()
        st = self.st = Storage(storage)
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.st, self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        for _ in range(1, n):
            blocks.append(block(self.st, self.inplanes, planes))
        return nn.Sequential(*blocks)

    def forward(self, x):
        st = self.st
        x = st.act(x)
        x = st.act(_cv(st, self.conv1, x))
        x = st.act(F.relu(self.bn1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
        return self.fc(x)


_RESNETS = {'resnet18': (BasicBlock, [2, 2, 2, 2]), 'resnet34': (BasicBlock, [3, 4, 6, 3]),
            'resnet50': (Bottleneck, [3, 4, 6, 3]), 'resnet101': (Bottleneck, [3, 4, 23, 3]),
            'resnet152': (Bottleneck, [3, 8, 36, 3])}


# ------------------------------------------------------------------------------------------ alexnet / vgg / squeezenet / densenet
# (reference call sites neuston_models.py:27-36,40-42; torchvision 0.8.2 alexnet.py, vgg.py, squeezenet.py, densenet.py restated
# from their published architectures; pinned by the published parameter totals in tests/test_oracle_cpu.py)
def _drop(st, x, masks, name, training, p=0.5):
    """nn.Dropout with an optional fixed keep-mask (parity tests); the HIP path stores the result"""
    if not training:
        return x
    if masks is not None and name in masks:
        return st.act(x * masks[name].to(x.dtype).reshape(x.shape) * (1.0 / (1.0 - p)))
    return st.act(F.dropout(x, p, True))


def _cbr(st, conv, x, relu=True):
    """conv + bias (+ReLU) as ONE stored tensor (the HIP epilogue adds the fp32 bias to the fp32 accumulator, then rounds)"""
    y = F.conv2d(x, st.weight(conv.weight), conv.bias, conv.stride, conv.padding)
    return st.act(F.relu(y) if relu else y)


def _lin(st, lin, x, relu=True):
    y = F.linear(x, st.weight(lin.weight), lin.bias)
    return st.act(F.relu(y) if relu else y)


class AlexNet(nn.Module):
    def __init__(self, num_classes=1000, storage='fp32'):
        super().__init__()
        self.st = Storage(storage)
        self.dropout_masks = None
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(True), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(True), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(True), nn.MaxPool2d(3, 2))
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Linear(4096, num_classes))

    def forward(self, x):
        st, f, c = self.st, self.features, self.classifier
        x = st.act(x)
        x = F.max_pool2d(_cbr(st, f[0], x), 3, 2)
        x = F.max_pool2d(_cbr(st, f[3], x), 3, 2)
        x = _cbr(st, f[6], x)
        x = _cbr(st, f[8], x)
        x = F.max_pool2d(_cbr(st, f[10], x), 3, 2)
        x = torch.flatten(self.avgpool(x), 1)
        x = _drop(st, x, self.dropout_masks, 'classifier.drop0', self.training)
        x = _lin(st, c[1], x)
        x = _drop(st, x, self.dropout_masks, 'classifier.drop1', self.training)
        x = _lin(st, c[4], x)
        return c[6](x)


_VGG_CFG = {'11': [64, 'M', 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
            '13': [64, 64, 'M', 128, 128, 'M', 256, 256, 'M', 512, 512, 'M', 512, 512, 'M'],
            '16': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M'],
            '19': [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']}


class VGG(nn.Module):
    def __init__(self, cfg, batch_norm, num_classes=1000, storage='fp32'):
        super().__init__()
        self.st = Storage(storage)
        self.dropout_masks = None
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers.append(nn.Conv2d(cin, v, 3, padding=1))
                if batch_norm:
                    layers.append(nn.BatchNorm2d(v))
                layers.append(nn.ReLU(True))
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        st, c = self.st, self.classifier
        x = st.act(x)
        mods = list(self.features)
        k = 0
        while k < len(mods):
            m = mods[k]
            if isinstance(m, nn.MaxPool2d):
                x = F.max_pool2d(x, 2, 2)
                k += 1
            elif k + 1 < len(mods) and isinstance(mods[k + 1], nn.BatchNorm2d):
                # conv (+bias) -> BN -> ReLU: the raw conv output and the activation are both stored
                y = st.act(F.conv2d(x, st.weight(m.weight), m.bias, m.stride, m.padding))
                x = st.act(F.relu(mods[k + 1](y)))
                k += 3
            else:
                x = _cbr(st, m, x)
                k += 2
        x = torch.flatten(self.avgpool(x), 1)
        x = _lin(st, c[0], x)
        x = _drop(st, x, self.dropout_masks, 'classifier.drop0', self.training)
        x = _lin(st, c[3], x)
        x = _drop(st, x, self.dropout_masks, 'classifier.drop1', self.training)
        return c[6](x)


class Fire(nn.Module):
    def __init__(self, st, cin, sq, e1, e3):
        super().__init__()
        self.st = st
        self.squeeze = nn.Conv2d(cin, sq, 1)
        self.squeeze_activation = nn.ReLU(True)
        self.expand1x1 = nn.Conv2d(sq, e1, 1)
        self.expand1x1_activation = nn.ReLU(True)
        self.expand3x3 = nn.Conv2d(sq, e3, 3, padding=1)
        self.expand3x3_activation = nn.ReLU(True)

    def forward(self, x):
        s = _cbr(self.st, self.squeeze, x)
        return torch.cat([_cbr(self.st, self.expand1x1, s), _cbr(self.st, self.expand3x3, s)], 1)


class SqueezeNet11(nn.Module):
    def __init__(self, num_classes=1000, storage='fp32'):
        super().__init__()
        st = self.st = Storage(storage)
        self.dropout_masks = None
        self.num_classes = num_classes
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 3, 2), nn.ReLU(True), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(st, 64, 16, 64, 64), Fire(st, 128, 16, 64, 64), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(st, 128, 32, 128, 128), Fire(st, 256, 32, 128, 128), nn.MaxPool2d(3, 2, ceil_mode=True),
            Fire(st, 256, 48, 192, 192), Fire(st, 384, 48, 192, 192), Fire(st, 384, 64, 256, 256), Fire(st, 512, 64, 256, 256))
        final_conv = nn.Conv2d(512, num_classes, 1)
        self.classifier = nn.Sequential(nn.Dropout(0.5), final_conv, nn.ReLU(True), nn.AdaptiveAvgPool2d((1, 1)))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                if m is final_conv:
                    nn.init.normal_(m.weight, 0.0, 0.01)
                else:
                    nn.init.kaiming_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        st, f = self.st, self.features
        x = st.act(x)
        x = F.max_pool2d(_cbr(st, f[0], x), 3, 2, ceil_mode=True)
        x = f[4](f[3](x))
        x = F.max_pool2d(x, 3, 2, ceil_mode=True)
        x = f[7](f[6](x))
        x = F.max_pool2d(x, 3, 2, ceil_mode=True)
        x = f[12](f[11](f[10](f[9](x))))
        x = _drop(st, x, self.dropout_masks, 'classifier.0', self.training)
        x = _cbr(st, self.classifier[1], x)
        return torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)


class _DenseLayer(nn.Module):
    def __init__(self, st, cin, growth, bn_size):
        super().__init__()
        self.st = st
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        st = self.st
        x = torch.cat(feats, 1)
        t = st.act(F.relu(self.norm1(x)))
        t = st.act(_cv(st, self.conv1, t))
        t = st.act(F.relu(self.norm2(t)))
        return st.act(_cv(st, self.conv2, t))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, st, nl, cin, bn_size, growth):
        super().__init__()
        for i in range(nl):
            self.add_module('denselayer%d' % (i + 1), _DenseLayer(st, cin + i * growth, growth, bn_size))

    def forward(self, x):
        feats = [x]
        for _name, layer in self.items():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, st, cin, cout):
        super().__init__()
        self.st = st
        self.add_module('norm', nn.BatchNorm2d(cin))
        self.add_module('relu', nn.ReLU(True))
        self.add_module('conv', nn.Conv2d(cin, cout, 1, bias=False))
        self.add_module('pool', nn.AvgPool2d(2, 2))

    def forward(self, x):
        st = self.st
        t = st.act(F.relu(self.norm(x)))
        t = st.act(_cv(st, self.conv, t))
        return st.act(F.avg_pool2d(t, 2, 2))


class DenseNet(nn.Module):
    def __init__(self, growth, blocks, c0, bn_size=4, num_classes=1000, storage='fp32'):
        super().__init__()
        st = self.st = Storage(storage)
        self.features = nn.Sequential()
        self.features.add_module('conv0', nn.Conv2d(3, c0, 7, 2, 3, bias=False))
        self.features.add_module('norm0', nn.BatchNorm2d(c0))
        self.features.add_module('relu0', nn.ReLU(True))
        self.features.add_module('pool0', nn.MaxPool2d(3, 2, 1))
        nf = c0
        for i, nl in enumerate(blocks):
            self.features.add_module('denseblock%d' % (i + 1), _DenseBlock(st, nl, nf, bn_size, growth))
            nf += nl * growth
            if i != len(blocks) - 1:
                self.features.add_module('transition%d' % (i + 1), _Transition(st, nf, nf // 2))
                nf //= 2
        self.features.add_module('norm5', nn.BatchNorm2d(nf))
        self.classifier = nn.Linear(nf, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        st, f = self.st, self.features
        x = st.act(x)
        x = st.act(_cv(st, f.conv0, x))
        x = st.act(F.relu(f.norm0(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        for name, m in f.named_children():
            if name.startswith(('denseblock', 'transition')):
                x = m(x)
        x = st.act(F.relu(f.norm5(x)))
        x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
        return self.classifier(x)


_DENSENETS = {'densenet121': (32, (6, 12, 24, 16), 64), 'densenet161': (48, (6, 12, 36, 24), 96),
              'densenet169': (32, (6, 12, 32, 32), 64), 'densenet201': (32, (6, 12, 48, 32), 64)}


def get_namebrand_model(model_name, num_o_classes, pretrained=False, storage='fp32'):
    """Oracle twin of ``/root/reference/neuston_models.py:22-45`` for every backbone family the reference accepts.

    ``pretrained=True`` only switches on inception's ``transform_input`` (what torchvision does when it
    loads ImageNet weights); no weights are downloaded -- callers load an explicit ``state_dict``.
    """
    if model_name == 'inception_v3':
        model = Inception3(1000, transform_input=bool(pretrained), storage=storage)
        model.AuxLogits.fc = nn.Linear(model.AuxLogits.fc.in_features, num_o_classes)
        model.fc = nn.Linear(model.fc.in_features, num_o_classes)
    elif model_name in _RESNETS:
        block, layers = _RESNETS[model_name]
        model = ResNet(block, layers, 1000, storage=storage)
        model.fc = nn.Linear(model.fc.in_features, num_o_classes)
    elif model_name == 'alexnet':
        model = AlexNet(1000, storage=storage)
        model.classifier[6] = nn.Linear(model.classifier[6].in_features, num_o_classes)
    elif model_name == 'squeezenet':
        model = SqueezeNet11(1000, storage=storage)
        model.classifier[1] = nn.Conv2d(512, num_o_classes, kernel_size=(1, 1), stride=(1, 1))
        model.num_classes = num_o_classes
    elif model_name.startswith('vgg') and model_name[3:5] in _VGG_CFG and model_name[5:] in ('', '_bn'):
        model = VGG(_VGG_CFG[model_name[3:5]], model_name.endswith('_bn'), 1000, storage=storage)
        model.classifier[6] = nn.Linear(model.classifier[6].in_features, num_o_classes)
    elif model_name in _DENSENETS:
        growth, blocks, c0 = _DENSENETS[model_name]
        model = DenseNet(growth, blocks, c0, 4, 1000, storage=storage)
        model.classifier = nn.Linear(model.classifier.in_features, num_o_classes)
    else:
        raise KeyError("model unknown!")
    return model
