"""CPU fp32 restatement of the encoders on the hot path (test infrastructure; see oracle/__init__.py).

Parameter names and shapes equal the reference's, so a reference `state_dict` loads unchanged:
  * VisualTransformer  -- `src/eoe/models/clip_official/clip/model.py:202-236` (+ ResidualAttentionBlock
    :167-188, LayerNorm :153-159, QuickGELU :162-164, nn.MultiheadAttention packed in_proj);
  * CNN32              -- `src/eoe/models/cnn.py:44-86`;
  * CustomNet head     -- `src/eoe/models/custom_base.py:6-51` (feature_model + final_linear(->256 | ->1)).
All arithmetic is written out with elementary torch-CPU ops (no nn.MultiheadAttention / nn.LayerNorm /
nn.BatchNorm) so that it is a restatement, not a call into the same library routine.
"""
import math
from collections import OrderedDict
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fill as _fill


def layer_norm(x, weight, bias, eps=1e-5):
    # model.py:153-159: LayerNorm computed in fp32 over the last dim, biased variance
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * weight + bias


def quick_gelu(x):
    # model.py:162-164
    return x * torch.sigmoid(1.702 * x)


class _OutProj(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Attn(nn.Module):
    """packed-projection multi-head self-attention (what nn.MultiheadAttention(d, h) computes, model.py:171)"""

    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _OutProj(d)

    def forward(self, x):  # x: N x L x D
        n, l, d = x.shape
        h, dh = self.heads, d // self.heads
        qkv = x @ self.in_proj_weight.t() + self.in_proj_bias
        q, k, v = qkv.split(d, dim=-1)
        q = q.reshape(n, l, h, dh).permute(0, 2, 1, 3)
        k = k.reshape(n, l, h, dh).permute(0, 2, 1, 3)
        v = v.reshape(n, l, h, dh).permute(0, 2, 1, 3)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        p = torch.softmax(s, dim=-1)
        o = (p @ v).permute(0, 2, 1, 3).reshape(n, l, d)
        return o @ self.out_proj.weight.t() + self.out_proj.bias


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))

    def forward(self, x):
        return layer_norm(x, self.weight, self.bias)


class _Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o)) if bias else None

    def forward(self, x):
        y = x @ self.weight.t()
        return y + self.bias if self.bias is not None else y


class ResidualAttentionBlock(nn.Module):
    # model.py:167-188
    def __init__(self, d, heads):
        super().__init__()
        self.attn = _Attn(d, heads)
        self.ln_1 = _LN(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", _Lin(d, 4 * d)), ("c_proj", _Lin(4 * d, d))]))
        self.ln_2 = _LN(d)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        x = x + self.mlp.c_proj(quick_gelu(self.mlp.c_fc(self.ln_2(x))))
        return x


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


class VisualTransformer(nn.Module):
    # model.py:202-236; (batch-major token order: the reference permutes to LND internally, model.py:227,229,
    # which does not change any value)
    def __init__(self, input_resolution=224, patch_size=32, width=768, layers=12, heads=12, output_dim=512):
        super().__init__()
        self.input_resolution, self.patch_size, self.output_dim = input_resolution, patch_size, output_dim
        self.width, self.layers, self.heads = width, layers, heads
        grid = input_resolution // patch_size
        self.conv1 = _Conv(3, width, patch_size, bias=False)
        self.class_embedding = nn.Parameter(torch.empty(width))
        self.positional_embedding = nn.Parameter(torch.empty(grid * grid + 1, width))
        self.ln_pre = _LN(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = _LN(width)
        self.proj = nn.Parameter(torch.empty(width, output_dim))

    def forward(self, x):
        n = x.shape[0]
        p, g = self.patch_size, self.input_resolution // self.patch_size
        # conv with kernel == stride == patch is a GEMM over non-overlapping patches (model.py:220-222)
        patches = x.reshape(n, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(n, g * g, 3 * p * p)
        tok = patches @ self.conv1.weight.reshape(self.width, -1).t()
        cls = self.class_embedding.reshape(1, 1, -1).expand(n, 1, self.width)
        x = torch.cat([cls, tok], dim=1) + self.positional_embedding          # model.py:223-224
        x = self.ln_pre(x)                                                     # :225
        x = self.transformer(x)                                                # :227-229
        x = self.ln_post(x[:, 0, :])                                           # :231
        return x @ self.proj                                                   # :233-234


class ClipViTNet(nn.Module):
    """CustomNet (custom_base.py:6-51) with feature_model = VisualTransformer, feature_dim = its output_dim:
    forward = final_linear(feature_model(x).flatten(1)) (:45-51); final_linear -> 256 (HSC) or 1 (clf)."""

    def __init__(self, clf=False, freeze=False, layers=12, width=768, heads=12, output_dim=512,
                 input_resolution=224, patch_size=32, prediction_head=True):
        super().__init__()
        self.feature_model = VisualTransformer(input_resolution, patch_size, width, layers, heads, output_dim)
        self.feature_dim, self.clf, self.freeze, self.prediction_head = output_dim, clf, freeze, prediction_head
        if prediction_head:
            self.final_linear = _Lin(output_dim, 1 if clf else 256)

    def freeze_parts(self):
        if self.freeze:
            for p in self.feature_model.parameters():
                p.requires_grad_(False)
            return True
        return False

    def forward(self, x):
        f = self.feature_model(x)
        return self.final_linear(f.flatten(1)) if self.prediction_head else f


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-4):
    """BatchNorm over all dims but 1 (what nn.BatchNorm1d/2d do): biased variance for normalisation,
    unbiased for the running estimate; updates the running buffers in place when training."""
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mu = x.mean(dim=dims)
        var = ((x - mu.reshape(shape)) ** 2).mean(dim=dims)
        cnt = x.numel() // x.shape[1]
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(momentum * mu)
            running_var.mul_(1 - momentum).add_(momentum * var * (cnt / max(cnt - 1, 1)))
    else:
        mu, var = running_mean, running_var
    y = (x - mu.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps)
    if weight is not None:
        y = y * weight.reshape(shape) + bias.reshape(shape)
    return y


class _BN(nn.Module):
    def __init__(self, c, eps, affine):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c)) if affine else None
        self.bias = nn.Parameter(torch.zeros(c)) if affine else None
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        if self.training:
            self.num_batches_tracked += 1
        return batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                          0.1, self.eps)


class CNN32(nn.Module):
    # cnn.py:44-86
    def __init__(self, rep_dim=256, bias=False, clf=False, grayscale=False):
        super().__init__()
        self.clf, self.grayscale, self.rep_dim = clf, grayscale, rep_dim
        cin = 1 if grayscale else 3
        self.conv1 = _Conv(cin, 32, 5, bias)
        self.bn2d1 = _BN(32, 1e-4, bias)
        self.conv2 = _Conv(32, 64, 5, bias)
        self.bn2d2 = _BN(64, 1e-4, bias)
        self.conv3 = _Conv(64, 128, 5, bias)
        self.bn2d3 = _BN(128, 1e-4, bias)
        self.fc1 = _Lin(128 * 4 * 4, 512, bias)
        self.bn1d1 = _BN(512, 1e-4, bias)
        self.fc2 = _Lin(512, rep_dim, bias)
        if clf:
            self.linear = _Lin(rep_dim, 1)

    @staticmethod
    def _conv(x, c):
        return F.conv2d(x, c.weight, c.bias, stride=1, padding=2)

    def forward(self, x):
        x = x.reshape(-1, 1 if self.grayscale else 3, 32, 32)
        x = F.max_pool2d(F.leaky_relu(self.bn2d1(self._conv(x, self.conv1)), 0.01), 2, 2)
        x = F.max_pool2d(F.leaky_relu(self.bn2d2(self._conv(x, self.conv2)), 0.01), 2, 2)
        x = F.max_pool2d(F.leaky_relu(self.bn2d3(self._conv(x, self.conv3)), 0.01), 2, 2)
        x = x.reshape(x.shape[0], -1)
        x = F.leaky_relu(self.bn1d1(self.fc1(x)), 0.01)
        x = self.fc2(x)
        return self.linear(x) if self.clf else x


class CNN28(nn.Module):
    # cnn.py:5-41 (1-channel 28x28 variant: two conv blocks, FC 1568 -> 64 -> rep_dim)
    def __init__(self, rep_dim=32, bias=False, clf=False):
        super().__init__()
        self.clf, self.rep_dim = clf, rep_dim
        self.conv1 = _Conv(1, 16, 5, bias)
        self.bn2d1 = _BN(16, 1e-4, bias)
        self.conv2 = _Conv(16, 32, 5, bias)
        self.bn2d2 = _BN(32, 1e-4, bias)
        self.fc1 = _Lin(32 * 7 * 7, 64, bias)
        self.bn1d1 = _BN(64, 1e-4, bias)
        self.fc2 = _Lin(64, rep_dim, bias)
        if clf:
            self.linear = _Lin(rep_dim, 1)

    def forward(self, x):
        x = x.reshape(-1, 1, 28, 28)
        x = F.max_pool2d(F.leaky_relu(self.bn2d1(CNN32._conv(x, self.conv1)), 0.01), 2, 2)
        x = F.max_pool2d(F.leaky_relu(self.bn2d2(CNN32._conv(x, self.conv2)), 0.01), 2, 2)
        x = x.reshape(x.shape[0], -1)
        x = F.leaky_relu(self.bn1d1(self.fc1(x)), 0.01)
        x = self.fc2(x)
        return self.linear(x) if self.clf else x


# ---------------------------------------------------------------------------------------------------------
# deterministic initialisation (shapes and scales of the reference's own init, values from oracle.fill)
# ---------------------------------------------------------------------------------------------------------
def init_std(name: str, shape, width: int = 768, layers: int = 12) -> float:
    """std used for the deterministic fill of a parameter: CLIP's own scales where it defines them
    (model.py:207-217 scale = width**-0.5; model.py:312-319 per-block stds), else a fan-in rule."""
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    if name.endswith("attn.in_proj_weight"):
        return attn_std
    if name.endswith("attn.out_proj.weight") or name.endswith("mlp.c_proj.weight"):
        return proj_std
    if name.endswith("mlp.c_fc.weight"):
        return fc_std
    if name.endswith("class_embedding") or name.endswith("positional_embedding") or name.endswith(".proj") \
            or name == "proj":
        return width ** -0.5
    if name.endswith("bias"):
        return 0.02
    if "ln_" in name and name.endswith("weight") or ("bn" in name and name.endswith("weight")):
        return 0.05  # around 1.0, see deterministic_init
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return (1.0 / max(fan_in, 1)) ** 0.5


def deterministic_init(model: nn.Module, tag: str = "w", width: int = 768, layers: int = 12) -> nn.Module:
    """fill every parameter of `model` with oracle.fill values (norm scales around 1, biases small but
    non-zero so that every gradient path is exercised); running BN buffers keep their defaults."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            std = init_std(name, tuple(p.shape), width, layers)
            is_scale = (("ln_" in name) or ("bn" in name)) and name.endswith("weight")
            arr = _fill.fill(f"{tag}/{name}", tuple(p.shape), std=std, mean=1.0 if is_scale else 0.0)
            p.copy_(torch.from_numpy(arr))
    return model


# ---------------------------------------------------------------------------------------------------------
# WideResNet = ResNet-18 layout + CBAM in every BasicBlock, 224x224 only
# (`src/eoe/models/resnet.py:25-152`, `src/eoe/models/cbam.py:7-107`)
# ---------------------------------------------------------------------------------------------------------
class _BasicConv(nn.Module):
    # cbam.py:7-23 (conv without bias + BatchNorm momentum 0.01, no relu in the spatial gate)
    def __init__(self):
        super().__init__()
        self.conv = _Conv(2, 1, 7, bias=False)
        self.bn = _BN(1, 1e-5, True)

    def forward(self, x):
        y = F.conv2d(x, self.conv.weight, None, stride=1, padding=3)
        bn = self.bn
        if bn.training:
            bn.num_batches_tracked += 1
        return batch_norm(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, 0.01, bn.eps)


class _ChannelGate(nn.Module):
    # cbam.py:31-66: shared MLP on the global average pool and the global max pool, summed, sigmoid, scale
    def __init__(self, c, r=16):
        super().__init__()
        self.mlp = nn.Sequential(OrderedDict([("1", _Lin(c, c // r)), ("3", _Lin(c // r, c))]))

    def _mlp(self, v):
        return self.mlp[1](torch.relu(self.mlp[0](v)))

    def forward(self, x):
        att = self._mlp(x.mean(dim=(2, 3))) + self._mlp(x.amax(dim=(2, 3)))
        return x * torch.sigmoid(att)[:, :, None, None]


class _SpatialGate(nn.Module):
    # cbam.py:76-92: cat(channel max, channel mean) -> 7x7 conv (2->1) + BN -> sigmoid -> scale
    def __init__(self):
        super().__init__()
        self.spatial = _BasicConv()

    def forward(self, x):
        comp = torch.cat([x.amax(dim=1, keepdim=True), x.mean(dim=1, keepdim=True)], dim=1)
        return x * torch.sigmoid(self.spatial(comp))


class _CBAM(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ChannelGate = _ChannelGate(c)
        self.SpatialGate = _SpatialGate()

    def forward(self, x):
        return self.SpatialGate(self.ChannelGate(x))


class _Downsample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("0", _Conv(cin, cout, 1, bias=False))
        self.add_module("1", _BN(cout, 1e-5, True))


class BasicBlock(nn.Module):
    # resnet.py:112-149
    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.stride = stride
        self.conv1 = _Conv(inplanes, planes, 3, bias=False)
        self.bn1 = _BN(planes, 1e-5, True)
        self.conv2 = _Conv(planes, planes, 3, bias=False)
        self.bn2 = _BN(planes, 1e-5, True)
        self.downsample = _Downsample(inplanes, planes) if downsample else None
        self.cbam = _CBAM(planes)

    def forward(self, x):
        out = torch.relu(self.bn1(F.conv2d(x, self.conv1.weight, None, stride=self.stride, padding=1)))
        out = self.bn2(F.conv2d(out, self.conv2.weight, None, stride=1, padding=1))
        res = x
        if self.downsample is not None:
            res = getattr(self.downsample, "1")(F.conv2d(x, getattr(self.downsample, "0").weight, None, stride=self.stride))
        return torch.relu(self.cbam(out) + res)


class WideResNet(nn.Module):
    # resnet.py:25-109.  `res` != 224 is the build's own generalisation (BASELINE.json config 2 names a 32 x 32 WideResNet; the
    # reference hard-codes view(-1, 3, 224, 224) and AvgPool2d(7), resnet.py:86,38): same layers, the input viewed as
    # res x res, the final res/32 x res/32 map averaged whole.
    def __init__(self, rep_dim=256, clf=False, res=224):
        super().__init__()
        assert res % 32 == 0
        self.clf, self.rep_dim, self.res = clf, rep_dim, res
        self.conv1 = _Conv(3, 64, 7, bias=False)
        self.bn1 = _BN(64, 1e-5, True)
        cfg = [(64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)]
        for i, (cin, planes, stride) in enumerate(cfg, start=1):
            ds = stride != 1 or cin != planes
            setattr(self, f"layer{i}", nn.Sequential(BasicBlock(cin, planes, stride, ds), BasicBlock(planes, planes)))
        self.fc = _Lin(512, rep_dim)
        if clf:
            self.linear = _Lin(rep_dim, 1)

    def forward(self, x):
        x = x.reshape(-1, 3, self.res, self.res)
        x = torch.relu(self.bn1(F.conv2d(x, self.conv1.weight, None, stride=2, padding=3)))
        x = F.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        x = F.avg_pool2d(x, self.res // 32).reshape(x.shape[0], -1)
        x = self.fc(x)
        return self.linear(x) if self.clf else x
