"""Drop-in for `src/eoe/models/resnet.py:25-152` (WideResNet = ResNet-18 layout + CBAM in every BasicBlock): same
constructor (rep_dim, clf), module tree, parameter / buffer names and initialisation (`resnet.py:50-66`).  The reference
accepts 224x224 inputs only (`view(-1, 3, 224, 224)`, `AvgPool2d(7)`; resnet.py:86,38); the extra `res` argument is this
build's own generalisation for BASELINE.json's "WideResNet backbone, 32x32" configuration: the same layers on a res x res
input, the final (res/32)^2 map averaged whole (at 224 that IS AvgPool2d(7)).
The torch.nn modules are parameter/buffer CONTAINERS; the forward runs im2col + MFMA GEMM convolutions with fused
BatchNorm/ReLU (`eoe_amd/csrc/conv.hip`) and the CBAM / pooling / residual kernels of `eoe_amd/csrc/cbam.hip`, on fp32
NHWC activations.  `WideResNet50Pretrained` (`resnet.py:8-21`, torchvision) is out of scope (unused by any runner)."""
import torch
import torch.nn as nn
import torch.nn.init as init

from .. import ops, ops_resnet
from .cbam import CBAM


def conv3x3(in_planes, out_planes, stride=1):
    """3x3 convolution with padding (resnet.py:152)"""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


_ONLY16 = __import__("os").environ.get("EOE_ONLY16", "1") != "0"      # conv1 -> bn1 -> relu writes its 16-bit copy only (0: A/B)


def _conv_bn(x, conv, bn, training, slope, is_image=False, normalize=None, want16=False, pool=1, passthrough=False, only16=False):
    """conv (no bias) -> BatchNorm2d -> ReLU (slope 0) or nothing (slope 1); fp32 NHWC in/out (only16: the output is consumed through its
    16-bit copy alone -- the fp32 tensor is not written)"""
    mean, std = normalize if (is_image and normalize is not None) else (None, None)
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    cfg = (training, bn.eps, bn.momentum, pool, is_image, mean, std, False, (k, k, s, p), slope, want16, passthrough, only16)
    return ops.conv_bn_act_pool(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, cfg)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_cbam=False):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.cbam = CBAM(planes, 16) if use_cbam else None

    def forward(self, x):      # x: fp32 NHWC (resnet.py:130-149)
        # identity shortcut: conv1 hands its input through as the junction's shortcut operand, so that the shortcut's
        # gradient reaches conv1's backward and is accumulated by its dgrad GEMM (no separate gradient add)
        # (down-sampling blocks: the 1x1 convolution of the shortcut reads the handed-through input, so ITS input gradient is the
        # one that arrives at conv1's backward and is accumulated by conv1's col2im)
        fuse = x.requires_grad and torch.is_grad_enabled()
        out = _conv_bn(x, self.conv1, self.bn1, self.training, 0.0, want16=True, passthrough=fuse, only16=_ONLY16)       # feeds conv2 only
        residual = x
        if fuse:
            out, residual = out
        out = _conv_bn(out, self.conv2, self.bn2, self.training, 1.0)
        if self.downsample is not None:
            residual = _conv_bn(residual, self.downsample[0], self.downsample[1], self.training, 1.0)
        if self.cbam is not None and not self.cbam.no_spatial:
            # channel gate, spatial gate and the residual junction as one unit (neither the channel-gated tensor nor the spatial gate's
            # input gradient is ever written; ops_resnet.FUSE_CBAM = False: the two-unit form of round 2, for A/B and tests)
            if ops_resnet.FUSE_CBAM:
                return ops_resnet.cbam_junction(out, residual, self.cbam, self.training)
            out = self.cbam.ChannelGate(out)
            sg = self.cbam.SpatialGate.spatial
            return ops_resnet.spatial_gate_add_relu(out, residual, sg.conv.weight, sg.bn, self.training)
        if self.cbam is not None:
            out = self.cbam(out)
        return ops_resnet.add_relu(out, residual)


class WideResNet(nn.Module):

    def __init__(self, rep_dim=256, clf=False, res=224):
        super().__init__()
        if res % 32 != 0 or res < 32:
            raise ValueError(f"WideResNet: res = {res} must be a positive multiple of 32 (five stride-2 stages)")
        self.res = res
        self.inplanes = 64
        self.clf = clf
        self.rep_dim = rep_dim
        att_type = 'CBAM'
        layers = [2, 2, 2, 2]
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.avgpool = nn.AvgPool2d(7)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.bam1, self.bam2, self.bam3 = None, None, None
        self.layer1 = self._make_layer(BasicBlock, 64, layers[0], att_type=att_type)
        self.layer2 = self._make_layer(BasicBlock, 128, layers[1], stride=2, att_type=att_type)
        self.layer3 = self._make_layer(BasicBlock, 256, layers[2], stride=2, att_type=att_type)
        self.layer4 = self._make_layer(BasicBlock, 512, layers[3], stride=2, att_type=att_type)
        self.fc = nn.Linear(512 * BasicBlock.expansion, self.rep_dim)
        if self.clf:
            self.linear = nn.Linear(self.rep_dim, 1)
            init.kaiming_normal_(self.linear.weight)
        # initialisation rule of resnet.py:54-66, keyed on the state_dict names
        init.kaiming_normal_(self.fc.weight)
        with torch.no_grad():
            for key, t in self.state_dict().items():
                leaf = key.split('.')[-1]
                if leaf == "weight":
                    if "conv" in key:
                        init.kaiming_normal_(t, mode='fan_out')
                    if "bn" in key:
                        t[...] = 0 if "SpatialGate" in key else 1
                elif leaf == "bias":
                    t[...] = 0
        self.normalize = None

    def set_normalize(self, mean, std):
        """fold the trainer's per-channel Normalize (`ad_trainer.py:413-425`) into the stem's patch extraction"""
        if mean is None:
            self.normalize = None
        else:
            dev = self.conv1.weight.device
            self.normalize = (torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous(),
                              torch.as_tensor(std, dtype=torch.float32, device=dev).contiguous())

    def _make_layer(self, block, planes, blocks, stride=1, att_type=None):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, use_cbam=att_type == 'CBAM')]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, use_cbam=att_type == 'CBAM'))
        return nn.Sequential(*layers)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("eoe_amd.WideResNet runs on the GPU only (no CPU fallback)")
        x = x.view(-1, 3, self.res, self.res)
        # all 16-bit weight copies that the optimiser step made stale, in one launch (the 19 block / down-sampling convolutions)
        ops.refresh_conv_weight_copies([m.weight for layer in (self.layer1, self.layer2, self.layer3, self.layer4)
                                        for m in layer.modules() if isinstance(m, nn.Conv2d) and m.weight.shape[1] % 8 == 0
                                        and m.weight.shape[2] != 7])
        mp = self.maxpool
        # conv1 -> bn1 -> relu -> maxpool in one unit: the 112x112x64 activation is never written; fp32 NHWC from here
        x = _conv_bn(x, self.conv1, self.bn1, self.training, 0.0, is_image=True, normalize=self.normalize, want16=True,
                     pool=(mp.kernel_size, mp.stride, mp.padding))
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        x = ops_resnet.GlobalAvgPoolFunction.apply(x)
        x = ops.linear(x, self.fc.weight, self.fc.bias)
        return ops.linear(x, self.linear.weight, self.linear.bias) if self.clf else x
