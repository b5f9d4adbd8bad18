"""nn.Module layer over the HIP operators, with the class names, constructor arguments and PARAMETER
NAMES/SHAPES of the MinkowskiEngine modules the reference instantiates, so
that the reference's module graphs (me_resnet.py, fcaf3d_neck_with_head.py) and released checkpoints
map one to one:
    MinkowskiConvolution.kernel (K,Cin,Cout) | (Cin,Cout) for k=1,s=1 ; .bias (1,Cout)
    MinkowskiGenerativeConvolutionTranspose.kernel (8,Cin,Cout)
    MinkowskiBatchNorm.bn.{weight,bias,running_mean,running_var,num_batches_tracked}
    MinkowskiInstanceNorm.{weight,bias} (1,C)
"""
import math

import torch
from torch import nn

from . import _lib as L
from . import functional as Fn
from .sparse import SparseTensor


class MinkowskiConvolution(nn.Module):
    """ME.MinkowskiConvolution(in, out, kernel_size, stride=1, dilation=1, bias=False, dimension=3)."""

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        self.kernel_volume = kernel_size ** 3
        if self.kernel_volume == 1 and stride == 1:
            self.kernel = nn.Parameter(torch.empty(in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.empty(self.kernel_volume, in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME default: U(-1/sqrt(Cin*k^3), +) for the kernel and the bias (Appendix A.3)
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def forward(self, x, want_stats=False):
        """want_stats (r5): a training-mode BatchNorm consumes the result — the launch also leaves the statistics table of its
        result (Fn.sparse_conv) on the returned tensor (`.stats`), where MinkowskiBatchNorm finds it"""
        grad_on = self.training and torch.is_grad_enabled()
        want_stats = want_stats and grad_on and self.bias is None
        stats = None
        if self.kernel_volume == 1 and self.stride == 1:
            f = Fn.sparse_conv(x.F, self.kernel.unsqueeze(0), None, x.F.shape[0], want_stats=want_stats)
            out_map = x.cmap
        else:
            out_map = x.cmap.strided(self.stride)
            km = x.cmap.kernel_map(out_map, self.kernel_size)
            f = Fn.sparse_conv(x.F, self.kernel, km, out_map.n, grad_on, want_stats=want_stats)
        if want_stats:
            f, stats = f
        if self.bias is not None:
            f = f + self.bias
        out = SparseTensor(f, coordinate_map_key=out_map)
        if stats is not None:
            out.stats = (stats, 1)
        return out


class MinkowskiGenerativeConvolutionTranspose(nn.Module):
    """ME.MinkowskiGenerativeConvolutionTranspose(in, out, kernel_size=2, stride=2): every input voxel
    emits its 8 children, out[8i+k] = in[i] @ kernel[k]  ==  one dense GEMM (N,Cin) x (Cin, 8*Cout)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3 and kernel_size == 2 and stride == 2 and not bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel = nn.Parameter(torch.empty(8, in_channels, out_channels))
        stdv = 1.0 / math.sqrt(in_channels * 8)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)

    def forward(self, x, want_stats=False):
        out_map = x.cmap.generate()
        w = self.kernel.permute(1, 0, 2).reshape(1, self.in_channels, 8 * self.out_channels)
        want_stats = want_stats and self.training and torch.is_grad_enabled()
        f = Fn.sparse_conv(x.F, w, None, x.F.shape[0], want_stats=want_stats)
        stats = None
        if want_stats:
            f, stats = f
        out = SparseTensor(f.reshape(-1, self.out_channels), coordinate_map_key=out_map)
        if stats is not None:
            out.stats = (stats, 8)              # the (n, 8 C) GEMM result viewed as (8 n, C): 8 column groups per channel
        return out


class _Act(nn.Module):
    act = None

    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        f = torch.relu(x.F) if self.act == 'relu' else torch.nn.functional.elu(x.F)
        return SparseTensor(f, coordinate_map_key=x.cmap)


class MinkowskiReLU(_Act):
    act = 'relu'


class MinkowskiELU(_Act):
    act = 'elu'


class MinkowskiBatchNorm(nn.Module):
    """ME.MinkowskiBatchNorm: nn.BatchNorm1d over the (N_total, C) feature matrix of this GPU's batch."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x, act=None, residual=None):
        bn = self.bn
        if self.training:
            part, groups = getattr(x, 'stats', None) or (None, 1)
            y, _ = Fn.bn_train(x.F, bn.weight, bn.bias, residual, bn.eps, act, bn.momentum, bn.running_mean,
                               bn.running_var, bn.num_batches_tracked, part=part, groups=groups)
        else:
            C = x.F.shape[1]
            stats = (bn.running_mean.reshape(1, C).contiguous(), bn.running_var.reshape(1, C).contiguous(),
                     torch.ones(1, device=x.F.device))
            y, _ = Fn.norm_act(x.F, bn.weight, bn.bias, residual=residual, eps=bn.eps, act=act, stats=stats)
        return SparseTensor(y, coordinate_map_key=x.cmap)


class MinkowskiInstanceNorm(nn.Module):
    """ME.MinkowskiInstanceNorm(C): per scene, per channel, biased variance, eps 1e-8 (Appendix A.6)."""

    def __init__(self, num_features):
        super().__init__()
        self.eps = 1e-8
        self.weight = nn.Parameter(torch.ones(1, num_features))
        self.bias = nn.Parameter(torch.zeros(1, num_features))

    def forward(self, x, act=None):
        y, _ = Fn.norm_act(x.F, self.weight, self.bias, seg=x.cmap.coords, nseg=x.cmap.batch_size, eps=self.eps, act=act)
        return SparseTensor(y, coordinate_map_key=x.cmap)


class MinkowskiMaxPooling(nn.Module):
    def __init__(self, kernel_size, stride=1, dilation=1, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.kernel_size, self.stride = kernel_size, stride

    def forward(self, x):
        out_map = x.cmap.strided(self.stride)
        km = x.cmap.kernel_map(out_map, self.kernel_size)
        return SparseTensor(Fn.max_pool(x.F, km), coordinate_map_key=out_map)


class MinkowskiPruning(nn.Module):
    def forward(self, x, mask, expect_n=None):
        from .sparse import compact_mask
        kept = compact_mask(mask, expect_n)
        if kept.numel() == x.F.shape[0]:
            return x
        return SparseTensor(Fn.gather_rows(x.F, kept), coordinate_map_key=x.cmap.pruned(kept))


_NORMS = (MinkowskiBatchNorm, MinkowskiInstanceNorm)
_CONVS = (MinkowskiConvolution, MinkowskiGenerativeConvolutionTranspose)


def run_sequential(seq, x):
    """nn.Sequential forward that fuses  norm -> activation  pairs into one kernel pass."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _NORMS) and i + 1 < len(mods) and isinstance(mods[i + 1], _Act):
            x = m(x, act=mods[i + 1].act)
            i += 2
        elif isinstance(m, _CONVS) and i + 1 < len(mods) and isinstance(mods[i + 1], MinkowskiBatchNorm):
            x = m(x, want_stats=True)                  # the BatchNorm behind it takes its statistics from this launch's epilogue
            i += 1
        else:
            x = m(x)
            i += 1
    return x


class BasicBlock(nn.Module):
    """MinkowskiEngine.modules.resnet_block.BasicBlock (imported at me_resnet.py:3), Appendix A.7."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x if self.downsample is None else run_sequential(self.downsample, x)
        out = self.norm1(self.conv1(x, want_stats=True), act='relu')
        out = self.conv2(out, want_stats=True)
        assert out.cmap is residual.cmap
        return self.norm2(out, act='relu', residual=residual.F)      # relu(bn(conv) + residual), one pass


class Bottleneck(nn.Module):
    """MinkowskiEngine.modules.resnet_block.Bottleneck (depth 50/101 at me_resnet.py:114-119)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=3):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * 4, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * 4, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x if self.downsample is None else run_sequential(self.downsample, x)
        out = self.norm1(self.conv1(x), act='relu')
        out = self.norm2(self.conv2(out), act='relu')
        out = self.conv3(out)
        return self.norm3(out, act='relu', residual=residual.F)


def kaiming_normal_(tensor, mode='fan_out', nonlinearity='relu'):
    """ME.utils.kaiming_normal_ for a (K,Cin,Cout) kernel: fan_out = Cout*K, fan_in = Cin*K (A.3)."""
    K = tensor.shape[0] if tensor.dim() == 3 else 1
    fan = (tensor.shape[-1] if mode == 'fan_out' else tensor.shape[-2]) * K
    gain = math.sqrt(2.0) if nonlinearity == 'relu' else 1.0
    with torch.no_grad():
        return tensor.normal_(0, gain / math.sqrt(fan))
