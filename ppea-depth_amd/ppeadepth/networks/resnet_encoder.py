"""ResNet-18 pose encoder (reference: networks/resnet_encoder.py:25-72, 367-409).

torchvision is not a dependency: the ResNet-18 definition (BasicBlock x [2,2,2,2], torchvision
attribute names so `pose_encoder.encoder.*` state_dict keys match) is stated here.
"""
import numpy as np
import torch
import torch.nn as nn


CHANNELS_LAST = True


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNetMultiImageInput(nn.Module):
    """ResNet-18 trunk whose first conv takes num_input_images * 3 channels."""

    def __init__(self, layers=(2, 2, 2, 2), num_classes=1000, num_input_images=1):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1, **kwargs):
        super().__init__()
        if num_layers != 18:
            raise NotImplementedError("the pose network of the hot path is ResNet-18")
        if pretrained:
            raise RuntimeError("ImageNet ResNet-18 weights need a network fetch (resnet_encoder.py:63-71); "
                               "load them through load_state_dict, or pass --weights_init scratch")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetMultiImageInput(num_input_images=num_input_images)
        for name, p in self.encoder.named_parameters():
            if "fc" in name:
                p.requires_grad = False

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        if CHANNELS_LAST and x.is_cuda:
            # MIOpen's implicit-GEMM convolutions are NHWC-native: with a channels_last input every activation of the
            # trunk stays NHWC and no conv pays a layout round trip on its activations (the weights keep their
            # storage -- they are views into the optimizer's flat buffer -- and are re-laid-out per call, which is small)
            x = x.contiguous(memory_format=torch.channels_last)
        self.features = [e.relu(e.bn1(e.conv1(x)))]
        self.features.append(e.layer1(e.maxpool(self.features[-1])))
        self.features.append(e.layer2(self.features[-1]))
        self.features.append(e.layer3(self.features[-1]))
        self.features.append(e.layer4(self.features[-1]))
        return self.features
