"""Drop-in for `src/eoe/models/cbam.py:7-107` (CBAM: channel gate + spatial gate) on the HIP path.  Same module tree
and parameter / buffer names (`ChannelGate.mlp.{1,3}.{weight,bias}`, `SpatialGate.spatial.conv.weight`,
`SpatialGate.spatial.bn.*`; BatchNorm momentum 0.01, `cbam.py:14`).  The torch.nn modules are parameter CONTAINERS;
the gates run as the kernels of `eoe_amd/csrc/cbam.hip` on fp32 NHWC activations."""
import torch.nn as nn

from .. import ops_resnet


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class BasicConv(nn.Module):
    """parameter container of conv (no bias) + BatchNorm (cbam.py:7-23); only the (2 -> 1, 7x7, no relu) instance of the
    spatial gate is on the hot path"""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, relu=True, bn=True, bias=False):
        super().__init__()
        if (in_planes, out_planes, kernel_size, stride, padding, relu, bn, bias) != (2, 1, 7, 1, 3, False, True, False):
            raise NotImplementedError("only the SpatialGate BasicConv(2, 1, 7, padding=3, relu=False) runs on the HIP path")
        self.out_channels = out_planes
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride=stride, padding=padding, bias=bias)
        self.bn = nn.BatchNorm2d(out_planes, eps=1e-5, momentum=0.01, affine=True)
        self.relu = None


class ChannelGate(nn.Module):
    def __init__(self, gate_channels, reduction_ratio=16, pool_types=('avg', 'max')):
        super().__init__()
        if tuple(pool_types) != ('avg', 'max'):
            raise NotImplementedError("the reference only instantiates pool_types ['avg', 'max'] (cbam.py:96)")
        self.gate_channels = gate_channels
        self.mlp = nn.Sequential(Flatten(), nn.Linear(gate_channels, gate_channels // reduction_ratio), nn.ReLU(),
                                 nn.Linear(gate_channels // reduction_ratio, gate_channels))
        self.pool_types = list(pool_types)

    def forward(self, x):      # x: fp32 NHWC
        l1, l3 = self.mlp[1], self.mlp[3]
        return ops_resnet.ChannelGateFunction.apply(x, l1.weight, l1.bias, l3.weight, l3.bias)


class SpatialGate(nn.Module):
    def __init__(self):
        super().__init__()
        self.spatial = BasicConv(2, 1, 7, stride=1, padding=3, relu=False)

    def forward(self, x):      # x: fp32 NHWC
        bn = self.spatial.bn
        return ops_resnet.SpatialGateFunction.apply(x, self.spatial.conv.weight, bn.weight, bn.bias, bn.running_mean,
                                                    bn.running_var, bn.num_batches_tracked,
                                                    (self.training, bn.eps, bn.momentum))


class CBAM(nn.Module):
    def __init__(self, gate_channels, reduction_ratio=16, pool_types=('avg', 'max'), no_spatial=False):
        super().__init__()
        self.ChannelGate = ChannelGate(gate_channels, reduction_ratio, pool_types)
        self.no_spatial = no_spatial
        if not no_spatial:
            self.SpatialGate = SpatialGate()

    def forward(self, x):
        x = self.ChannelGate(x)
        return x if self.no_spatial else self.SpatialGate(x)
