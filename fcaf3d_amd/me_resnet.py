"""Sparse ResNet backbone — host-side mirror of mmdet3d/models/backbones/me_resnet.py (ResNetBase :8-99,
MEResNet3D :102-123): same class names, constructor arguments, sub-module names (conv1.{0..3},
layer{1..4}.{j}.{conv1,norm1,conv2,norm2,downsample.{0,1}}) and parameter shapes, on HIP operators."""
from torch import nn

from . import nn as MEnn
from .registry import BACKBONES


class ResNetBase(nn.Module):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, n_outs):
        super().__init__()
        self.n_outs = n_outs
        self.inplanes = self.INIT_DIM
        # stem: conv k3 s2 -> instance norm -> ReLU -> max-pool k2 s2   (tensor stride 1 -> 2 -> 4)
        self.conv1 = nn.Sequential(
            MEnn.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=3, stride=2, dimension=3),
            MEnn.MinkowskiInstanceNorm(self.inplanes),
            MEnn.MinkowskiReLU(inplace=True),
            MEnn.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3))
        for i in range(min(n_outs, 4)):
            setattr(self, f'layer{i + 1}', self._make_layer(self.BLOCK, self.PLANES[i], self.LAYERS[i], stride=2))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, MEnn.MinkowskiConvolution):
                MEnn.kaiming_normal_(m.kernel, mode='fan_out', nonlinearity='relu')
            if isinstance(m, MEnn.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                MEnn.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                                          dimension=3),
                MEnn.MinkowskiBatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=downsample, dimension=3)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, stride=1, dilation=dilation, dimension=3))
        return nn.Sequential(*layers)

    def forward(self, x):
        outs = []
        x = MEnn.run_sequential(self.conv1, x)
        for i in range(min(self.n_outs, 4)):
            x = getattr(self, f'layer{i + 1}')(x)
            outs.append(x)
        return outs


@BACKBONES.register_module()
class MEResNet3D(ResNetBase):
    _ARCH = {14: (MEnn.BasicBlock, (1, 1, 1, 1)), 18: (MEnn.BasicBlock, (2, 2, 2, 2)),
             34: (MEnn.BasicBlock, (3, 4, 6, 3)), 50: (MEnn.Bottleneck, (4, 3, 6, 3)),
             101: (MEnn.Bottleneck, (3, 4, 23, 3))}

    def __init__(self, in_channels, depth, n_outs=4):
        if depth not in self._ARCH:
            raise ValueError(f'invalid depth={depth}')
        self.BLOCK, self.LAYERS = self._ARCH[depth]
        super().__init__(in_channels, n_outs)
