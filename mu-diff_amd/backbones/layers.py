"""DDPM-style initialisers, conv factories, NIN and the sinusoidal timestep embedding - the five
symbols of the reference's backbones/layers.py that the default NCSN++ config instantiates
(default_init:92, ddpm_conv1x1:104, ddpm_conv3x3:122, get_timestep_embedding:465, NIN:496).
The NCSNv1/v2 legacy blocks of that file are never instantiated by the sampling path and are not
rebuilt."""
import numpy as np
import torch
import torch.nn as nn

from mudiff_hip import ops
from mudiff_hip.ops import View


def get_act(config):
    name = config.nonlinearity.lower()
    if name == 'elu':
        return nn.ELU()
    if name == 'relu':
        return nn.ReLU()
    if name == 'lrelu':
        return nn.LeakyReLU(negative_slope=0.2)
    if name == 'swish':
        return nn.SiLU()
    raise NotImplementedError('activation function does not exist!')


def variance_scaling(scale, mode, distribution, in_axis=1, out_axis=0, dtype=torch.float32, device='cpu'):
    """JAX-style variance scaling (layers.py:58-89 of the reference)."""
    def init(shape, dtype=dtype, device=device):
        receptive = np.prod(shape) / shape[in_axis] / shape[out_axis]
        fan_in, fan_out = shape[in_axis] * receptive, shape[out_axis] * receptive
        denom = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2}[mode]
        variance = scale / denom
        if distribution == 'normal':
            return torch.randn(*shape, dtype=dtype, device=device) * np.sqrt(variance)
        if distribution == 'uniform':
            return (torch.rand(*shape, dtype=dtype, device=device) * 2. - 1.) * np.sqrt(3 * variance)
        raise ValueError('invalid distribution for variance scaling initializer')
    return init


def default_init(scale=1.):
    return variance_scaling(1e-10 if scale == 0 else scale, 'fan_avg', 'uniform')


def ddpm_conv1x1(in_planes, out_planes, stride=1, bias=True, init_scale=1., padding=0):
    conv = nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, padding=padding, bias=bias)
    conv.weight.data = default_init(init_scale)(conv.weight.data.shape)
    nn.init.zeros_(conv.bias)
    return conv


def ddpm_conv3x3(in_planes, out_planes, stride=1, bias=True, dilation=1, init_scale=1., padding=1):
    conv = nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=padding, dilation=dilation, bias=bias)
    conv.weight.data = default_init(init_scale)(conv.weight.data.shape)
    nn.init.zeros_(conv.bias)
    return conv


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """[sin | cos] positional embedding (layers.py:465-479 of the reference) - HIP kernel."""
    assert len(timesteps.shape) == 1
    return ops.timestep_embedding(timesteps.to(torch.int64), embedding_dim, float(max_positions))


class NIN(nn.Module):
    """Per-pixel linear map x @ W[in,out] + b (layers.py:496-505 of the reference): a 1x1 GEMM on the
    matrix cores."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(default_init(scale=init_scale)((in_dim, num_units)), requires_grad=True)
        self.b = nn.Parameter(torch.zeros(num_units), requires_grad=True)
        self._packed = None

    def forward(self, x):
        xv = View.from_nchw(x)
        key = (self.W._version, self.W.data_ptr())
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_matrix_in_out(self.W))
        out = ops.conv(xv, self._packed[1], 1, self.W.shape[1], mfma=True, bias=self.b.detach())
        return out.to_nchw()
