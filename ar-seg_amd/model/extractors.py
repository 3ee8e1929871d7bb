"""Dilated (stride-8) ResNet-18 feature extractor -- mirror of ``model/extractors.py:35-66,108-158,340-358``."""
from __future__ import annotations


from torch import nn

from .. import _lib, ops
from ..packing import PackedConv
from ._common import HipModule


def conv3x3(in_planes, out_planes, stride=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


class BasicBlock(HipModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride=stride, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, stride=1, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def _pack(self, device):
        pk = {"c1": PackedConv.from_modules(self.conv1, self.bn1, _lib.ACT_RELU, device=device),
              "c2": PackedConv.from_modules(self.conv2, self.bn2, _lib.ACT_RELU, device=device)}   # ReLU after the residual add
        if self.downsample is not None:
            pk["ds"] = PackedConv.from_modules(self.downsample[0], self.downsample[1], _lib.ACT_NONE, device=device)
        return pk

    def forward_nhwc(self, x, out=None):
        pk = self.packed()
        y = ops.conv2d(x, pk["c1"])
        res = ops.conv2d(x, pk["ds"]) if "ds" in pk else x
        return ops.conv2d(y, pk["c2"], residual=res, out=out)


class ResNet(HipModule):
    def __init__(self, block, layers=(3, 4, 23, 3), input_channel=3):
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(input_channel, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)
        # (no weight initialisation here: inference only, parameters always come from a state_dict)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        # as in the reference (extractors.py:139), the first block of a layer is NOT dilated
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*layers)

    def _pack(self, device):
        return {"stem": PackedConv.from_modules(self.conv1, self.bn1, _lib.ACT_RELU, device=device)}

    def forward_nhwc(self, x4, out_x=None):
        """x4: NHWC input padded to 4 channels.  Returns (layer4 out, layer3 out); ``out_x`` optionally
        receives the layer4 output in place (e.g. a channel slice of the pyramid-pooling concat buffer)."""
        x = ops.conv2d(x4, self.packed()["stem"])
        x = ops.maxpool3x3s2(x)
        for blk in self.layer1:
            x = blk.forward_nhwc(x)
        for blk in self.layer2:
            x = blk.forward_nhwc(x)
        for blk in self.layer3:
            x = blk.forward_nhwc(x)
        x_3 = x
        n4 = len(self.layer4)
        for i, blk in enumerate(self.layer4):
            x = blk.forward_nhwc(x, out=out_x if i == n4 - 1 else None)
        return x, x_3

    def forward(self, x):
        self._check_inference()
        N, C, H, W = x.shape
        f, f3 = self.forward_nhwc(ops.frame_to_nhwc4(x, H, W))
        return ops.as_nchw(f), ops.as_nchw(f3)


def resnet18(pretrained=True, input_channel=3):
    """``pretrained`` is accepted for signature compatibility (extractors.py:340); the ImageNet download it
    triggers in the reference is irrelevant once a checkpoint is loaded and impossible offline."""
    if input_channel != 3:
        raise NotImplementedError("only RGB input is on the hot path")
    return ResNet(BasicBlock, [2, 2, 2, 2], input_channel=input_channel)
