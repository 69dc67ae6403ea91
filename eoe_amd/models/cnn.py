"""Drop-ins for `src/eoe/models/cnn.py` -- CNN32 (:44-86) and its 1-channel 28x28 sibling CNN28 (:5-41).  CNN32: same constructor (rep_dim, bias, clf, grayscale), parameter and
buffer names (`conv{1,2,3}`, `bn2d{1,2,3}`, `fc1`, `bn1d1`, `fc2`, `linear`; BatchNorm eps 1e-4, affine = bias), same
initialisation (xavier_normal with the leaky_relu gain, overwritten by `reset_parameters()` under the trainer's
`weight_reset`, `ad_trainer.py:31-34,237-239`).  The torch.nn modules are parameter/buffer CONTAINERS: the forward
runs im2col + MFMA GEMM + fused BatchNorm/LeakyReLU/MaxPool HIP kernels (`eoe_amd/csrc/conv.hip`)."""
import torch
import torch.nn as nn

from .. import ops


class CNN32(nn.Module):
    """some CNN architecture for 32x32 images (cnn.py:44)"""

    def __init__(self, rep_dim=256, bias=False, clf=False, grayscale=False):
        super().__init__()
        if grayscale:
            raise NotImplementedError("grayscale CNN32 is not used by the hot path (train_cifar.py:44)")
        self.clf, self.grayscale, self.rep_dim = clf, grayscale, rep_dim
        self.pool = nn.MaxPool2d(2, 2)
        gain = nn.init.calculate_gain("leaky_relu")
        self.conv1 = nn.Conv2d(3, 32, 5, bias=bias, padding=2)
        nn.init.xavier_normal_(self.conv1.weight, gain=gain)
        self.bn2d1 = nn.BatchNorm2d(32, eps=1e-04, affine=bias)
        self.conv2 = nn.Conv2d(32, 64, 5, bias=bias, padding=2)
        nn.init.xavier_normal_(self.conv2.weight, gain=gain)
        self.bn2d2 = nn.BatchNorm2d(64, eps=1e-04, affine=bias)
        self.conv3 = nn.Conv2d(64, 128, 5, bias=bias, padding=2)
        nn.init.xavier_normal_(self.conv3.weight, gain=gain)
        self.bn2d3 = nn.BatchNorm2d(128, eps=1e-04, affine=bias)
        self.fc1 = nn.Linear(128 * 4 * 4, 512, bias=bias)
        nn.init.xavier_normal_(self.fc1.weight, gain=gain)
        self.bn1d1 = nn.BatchNorm1d(512, eps=1e-04, affine=bias)
        self.fc2 = nn.Linear(512, self.rep_dim, bias=bias)
        nn.init.xavier_normal_(self.fc2.weight)
        if self.clf:
            self.linear = nn.Linear(self.rep_dim, 1)
        self.normalize = None

    def set_normalize(self, mean, std):
        if mean is None:
            self.normalize = None
        else:
            dev = self.conv1.weight.device
            self.normalize = (torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous(),
                              torch.as_tensor(std, dtype=torch.float32, device=dev).contiguous())

    def _layer(self, x, conv, bn, is_image, flat_out):
        mean, std = self.normalize if (is_image and self.normalize is not None) else (None, None)
        # 5x5 stride 1 pad 2, LeakyReLU(0.01), MaxPool 2; a 16-bit copy of the output feeds the next conv's implicit GEMM
        cfg = (self.training, bn.eps, bn.momentum, 2, is_image, mean, std, flat_out, (5, 5, 1, 2), 0.01, not flat_out)
        return ops.conv_bn_act_pool(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    bn.num_batches_tracked, cfg)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("eoe_amd.CNN32 runs on the GPU only (no CPU fallback)")
        x = x.view(-1, 3, 32, 32)
        if ops._implicit_conv:          # the three weight packs in one launch (conv1 gathers from the channel-padded NHWC8 image)
            ops.refresh_conv_weight_copies([(self.conv1.weight, 8), self.conv2.weight, self.conv3.weight])
        x = self._layer(x, self.conv1, self.bn2d1, True, False)
        x = self._layer(x, self.conv2, self.bn2d2, False, False)
        x = self._layer(x, self.conv3, self.bn2d3, False, True)          # NCHW-flattened [n, 2048] (cnn.py:83)
        x = ops.linear(x, self.fc1.weight, self.fc1.bias)
        bn = self.bn1d1
        x = ops.BnActFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                    (self.training, bn.eps, bn.momentum))
        x = ops.linear(x, self.fc2.weight, self.fc2.bias)
        return ops.linear(x, self.linear.weight, self.linear.bias) if self.clf else x


class CNN28(nn.Module):
    """some CNN architecture for 28x28 images (cnn.py:5-41): conv5x5(1->16) + BN + LeakyReLU + pool, conv5x5(16->32) + BN + LeakyReLU +
    pool, FC 1568 -> 64 + BN1d + LeakyReLU, FC 64 -> rep_dim [-> 1]; same names, eps 1e-4, affine = bias, same initialisation.
    Runs on the CNN32 kernels: materialised patches for the 1-channel first layer, the narrow-channel implicit GEMM for the second."""

    def __init__(self, rep_dim=32, bias=False, clf=False):
        super().__init__()
        self.clf, self.rep_dim = clf, rep_dim
        self.pool = nn.MaxPool2d(2, 2)
        gain = nn.init.calculate_gain("leaky_relu")
        self.conv1 = nn.Conv2d(1, 16, 5, bias=bias, padding=2)
        nn.init.xavier_normal_(self.conv1.weight, gain=gain)
        self.bn2d1 = nn.BatchNorm2d(16, eps=1e-04, affine=bias)
        self.conv2 = nn.Conv2d(16, 32, 5, bias=bias, padding=2)
        nn.init.xavier_normal_(self.conv2.weight, gain=gain)
        self.bn2d2 = nn.BatchNorm2d(32, eps=1e-04, affine=bias)
        self.fc1 = nn.Linear(32 * 7 * 7, 64, bias=bias)
        nn.init.xavier_normal_(self.fc1.weight, gain=gain)
        self.bn1d1 = nn.BatchNorm1d(64, eps=1e-04, affine=bias)
        self.fc2 = nn.Linear(64, self.rep_dim, bias=bias)
        nn.init.xavier_normal_(self.fc2.weight)
        if self.clf:
            self.linear = nn.Linear(self.rep_dim, 1)
        self.normalize = None

    set_normalize = CNN32.set_normalize
    _layer = CNN32._layer

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("eoe_amd.CNN28 runs on the GPU only (no CPU fallback)")
        x = x.view(-1, 1, 28, 28)
        x = self._layer(x, self.conv1, self.bn2d1, True, False)
        x = self._layer(x, self.conv2, self.bn2d2, False, True)          # NCHW-flattened [n, 1568] (cnn.py:36)
        x = ops.linear(x, self.fc1.weight, self.fc1.bias)
        bn = self.bn1d1
        x = ops.BnActFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                    (self.training, bn.eps, bn.momentum))
        x = ops.linear(x, self.fc2.weight, self.fc2.bias)
        return ops.linear(x, self.linear.weight, self.linear.bias) if self.clf else x
