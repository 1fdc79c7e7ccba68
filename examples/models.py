"""Networks for the BASELINE.json configurations, written for qsparse-style conversion.

* ``MnistNet``: the CNN of the reference's MNIST example (architecture of examples/mnist.py:17-44).
* ``resnet18`` / ``resnet50``: standard residual networks with ONE ``nn.ReLU`` module per call site --
  ``convert`` keys operators on module instances, so a shared ``self.relu`` would receive a single
  PruneLayer and fail as soon as two sites disagree on the channel count.
* ``convert_pq``: the reference's ``--pq`` recipe (examples/mnist.py:192-199): channel-prune and
  tensor-wise quantize every ReLU output but the last, quantize conv/linear weights and the input.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import qsparse_amd as qs


class MnistNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv_part = nn.Sequential(nn.Conv2d(1, 32, 3, 1), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 64, 3, 1),
                                       nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2), nn.Dropout(0.25))
        self.linear_part = nn.Sequential(nn.Flatten(), nn.Linear(9216, 128), nn.BatchNorm1d(128), nn.ReLU(),
                                         nn.Dropout(0.5), nn.Linear(128, 10))

    def forward(self, x):
        return F.log_softmax(self.linear_part(self.conv_part(x)), dim=1)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu2 = nn.ReLU()
        self.down = None
        if stride != 1 or cin != planes:
            self.down = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu2(out + (x if self.down is None else self.down(x)))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1):
        super().__init__()
        cout = planes * 4
        self.conv1, self.bn1, self.relu1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes), nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # stride on the 3x3 (v1.5)
        self.bn2, self.relu2 = nn.BatchNorm2d(planes), nn.ReLU()
        self.conv3, self.bn3, self.relu3 = nn.Conv2d(planes, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        out = self.relu1(self.bn1(self.conv1(x)))
        out = self.relu2(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu3(out + (x if self.down is None else self.down(x)))


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=10, cifar_stem=True, width=64):
        super().__init__()
        if cifar_stem:
            self.stem = nn.Sequential(nn.Conv2d(3, width, 3, 1, 1, bias=False), nn.BatchNorm2d(width), nn.ReLU())
        else:
            self.stem = nn.Sequential(nn.Conv2d(3, width, 7, 2, 3, bias=False), nn.BatchNorm2d(width), nn.ReLU(),
                                      nn.MaxPool2d(3, 2, 1))
        cin, stages = width, []
        for i, n in enumerate(layers):
            planes = width * 2 ** i
            for j in range(n):
                stages.append(block(cin, planes, stride=(1 if i == 0 or j > 0 else 2)))
                cin = planes * block.expansion
        self.stages = nn.Sequential(*stages)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        return self.fc(torch.flatten(self.pool(self.stages(self.stem(x))), 1))


def resnet18(num_classes=10, cifar_stem=True, width=64):
    return ResNet(BasicBlock, [2, 2, 2, 2], num_classes, cifar_stem, width)


def resnet50(num_classes=1000, cifar_stem=False, width=64):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes, cifar_stem, width)


def convert_pq(model, sparsity=0.75, bits=4, prune_start=2, prune_interval=2, repetition=2, quant_timeout=1, log=False,
               fuse=True):
    """the reference's --pq recipe with explicit schedule arguments"""
    model = qs.convert(model, qs.prune(sparsity=sparsity, dimensions={1}, start=prune_start, interval=prune_interval,
                                       repetition=repetition),
                       activation_layers=[nn.ReLU], excluded_activation_layer_indexes=[(nn.ReLU, [-1])], log=log, fuse=fuse)
    model = qs.convert(model, qs.quantize(bits=bits, channelwise=-1, timeout=quant_timeout), activation_layers=[nn.ReLU],
                       weight_layers=[nn.Conv2d, nn.Linear], input=True, log=log, fuse=fuse)
    return model


class TokenBlock(nn.Module):
    """the MLP half of a transformer encoder block on token-major (B, T, C) activations: LayerNorm -> Linear -> act -> Linear,
    residual.  One activation module per site."""

    def __init__(self, dim, hidden, act=nn.GELU):
        super().__init__()
        self.norm, self.fc1, self.act, self.fc2 = nn.LayerNorm(dim), nn.Linear(dim, hidden), act(), nn.Linear(hidden, dim)

    def forward(self, x):
        return x + self.fc2(self.act(self.fc1(self.norm(x))))


class TokenNet(nn.Module):
    """patch embedding (a strided convolution, flattened to tokens) -> `depth` TokenBlocks -> mean over tokens -> classifier"""

    def __init__(self, num_classes=10, dim=64, hidden=128, depth=2, patch=4, act=nn.GELU):
        super().__init__()
        self.embed = nn.Conv2d(3, dim, patch, patch)
        self.blocks = nn.Sequential(*[TokenBlock(dim, hidden, act) for _ in range(depth)])
        self.norm, self.head = nn.LayerNorm(dim), nn.Linear(dim, num_classes)

    def forward(self, x):
        t = self.embed(x).flatten(2).transpose(1, 2).contiguous()      # (B, T, C)
        return self.head(self.norm(self.blocks(t)).mean(1))


def convert_pq_tokens(model, act=nn.GELU, sparsity=0.5, bits=4, prune_start=2, prune_interval=2, repetition=2, quant_timeout=1,
                      log=False, fuse=True):
    """the --pq recipe on token-major activations: the hidden activations of every block pruned along their LAST dim
    (`dimensions={2}`: (B, T, hidden) -> hidden channels) and quantized tensor-wise, Linear / Conv2d weights quantized"""
    model = qs.convert(model, qs.prune(sparsity=sparsity, dimensions={2}, start=prune_start, interval=prune_interval,
                                       repetition=repetition), activation_layers=[act], log=log, fuse=fuse)
    model = qs.convert(model, qs.quantize(bits=bits, channelwise=-1, timeout=quant_timeout), activation_layers=[act],
                       weight_layers=[nn.Conv2d, nn.Linear], log=log, fuse=fuse)
    return model
