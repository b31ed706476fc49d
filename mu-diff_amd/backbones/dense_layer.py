"""`dense` / `conv2d` factories (reference backbones/dense_layer.py:63-82): plain nn.Linear /
nn.Conv2d parameter holders with fan-avg uniform initialisation.  Inside the generators their weights
are consumed by the HIP dense / convolution kernels."""
import math

import torch
import torch.nn as nn


def variance_scaling_init_(tensor, scale):
    """U(-b, b), b = sqrt(3*scale / fan_avg-as-computed-by-the-reference).  The reference's helper maps
    every mode other than 'fan_in' to fan_out (dense_layer.py:33), so 'fan_avg' means fan_out here."""
    scale = 1e-10 if scale == 0 else scale
    fan_out = tensor.shape[0] * (tensor[0][0].numel() if tensor.dim() > 2 else 1)
    bound = math.sqrt(3.0 * scale / max(1.0, fan_out))
    with torch.no_grad():
        return tensor.uniform_(-bound, bound)


def dense(in_channels, out_channels, init_scale=1.):
    lin = nn.Linear(in_channels, out_channels)
    variance_scaling_init_(lin.weight, scale=init_scale)
    nn.init.zeros_(lin.bias)
    return lin


def conv2d(in_planes, out_planes, kernel_size=(3, 3), stride=1, dilation=1, padding=1, bias=True, padding_mode='zeros',
           init_scale=1.):
    conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                     bias=bias, padding_mode=padding_mode)
    variance_scaling_init_(conv.weight, scale=init_scale)
    if bias:
        nn.init.zeros_(conv.bias)
    return conv
