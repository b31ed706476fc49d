"""StyleGAN2-style FIR resampling front ends (reference backbones/up_or_down_sampling.py) over the
gfx950 up-FIR-down kernel.  Public functions keep the reference's names, arguments and NCHW tensors."""
import numpy as np
import torch
import torch.nn as nn

from mudiff_hip import ops
from mudiff_hip.ops import View
from utils.op import upfirdn2d


def _setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    assert k.ndim == 2 and k.shape[0] == k.shape[1]
    return k


def fir_params(mode, k=None, factor=2, gain=1, conv_k=3):
    """(kernel2d, up, down, (pad0, pad1)) for 'up' (upsample_2d:200-229), 'down' (downsample_2d:232-262)
    and 'conv_down' (the FIR stage of conv_downsample_2d:149-183)."""
    if k is None:
        k = [1] * factor
    if mode == 'up':
        kk = _setup_kernel(k) * (gain * (factor ** 2))
        p = kk.shape[0] - factor
        return kk, factor, 1, ((p + 1) // 2 + factor - 1, p // 2)
    if mode == 'naive_up':       # nearest-neighbour repeat (naive_upsample_2d:64-68): ones(f,f), zero-stuffed input, pad (f-1, 0)
        return np.ones((factor, factor), np.float32), factor, 1, (factor - 1, 0)
    if mode == 'naive_down':     # box mean (naive_downsample_2d:71-74)
        return np.full((factor, factor), 1.0 / (factor * factor), np.float32), 1, factor, (0, 0)
    kk = _setup_kernel(k) * gain
    if mode == 'down':
        p = kk.shape[0] - factor
        return kk, 1, factor, ((p + 1) // 2, p // 2)
    if mode == 'conv_down':
        p = (kk.shape[0] - factor) + (conv_k - 1)
        return kk, 1, 1, ((p + 1) // 2, p // 2)
    raise ValueError(mode)


_DEV_KERNELS = {}


def _device_kernel(kk, device):
    """FIR taps as a cached device tensor (no host->device copy per call, so hipGraph capture works)."""
    key = (kk.tobytes(), kk.shape, str(device))
    t = _DEV_KERNELS.get(key)
    if t is None:
        t = torch.tensor(kk, device=device, dtype=torch.float32)
        _DEV_KERNELS[key] = t
    return t


def upsample_2d(x, k=None, factor=2, gain=1):
    kk, up, down, pad = fir_params('up', k, factor, gain)
    return upfirdn2d(x, torch.tensor(kk, device=x.device, dtype=x.dtype), up=up, pad=pad)


def downsample_2d(x, k=None, factor=2, gain=1):
    kk, up, down, pad = fir_params('down', k, factor, gain)
    return upfirdn2d(x, torch.tensor(kk, device=x.device, dtype=x.dtype), down=down, pad=pad)


def conv_downsample_2d(x, w, k=None, factor=2, gain=1):
    """FIR (padded once) then the strided convolution with OIHW weights `w`, no bias."""
    kk, _, _, pad = fir_params('conv_down', k, factor, gain, conv_k=w.shape[-1])
    xf = upfirdn2d(x, torch.tensor(kk, device=x.device, dtype=x.dtype), pad=pad)
    out = ops.conv(View.from_nchw(xf), ops.direct_weight(w), w.shape[-1], w.shape[0], mfma=False, stride=factor, pad=0)
    return out.to_nchw()


def naive_upsample_2d(x, factor=2):
    """Nearest-neighbour repeat (reference :64-68) on the same up-FIR-down kernel: exact (one tap of weight 1)."""
    kk, up, down, pad = fir_params('naive_up', factor=factor)
    return upfirdn2d(x, _device_kernel(kk, x.device), up=up, pad=pad)


def naive_downsample_2d(x, factor=2):
    """factor x factor box mean (reference :71-74)."""
    kk, up, down, pad = fir_params('naive_down', factor=factor)
    return upfirdn2d(x, _device_kernel(kk, x.device), down=down, pad=pad)


def resample_view(x: View, direction, fir, fir_kernel):
    """Parameter-free x2 resampling of an NHWC view (Upsample / Downsample with_conv=False): the single-channel image
    pyramids go through the planes kernel (NHWC == planes when C == 1), feature maps through the NHWC kernel."""
    kk, up, down, pad = fir_params(direction if fir else 'naive_' + direction, fir_kernel)
    if x.C % 4 == 0:
        return ops.fir_nhwc(x, kk, up, down, pad)[0]
    assert x.ld == x.C, 'planes path needs a dense view'
    t = x.base.reshape(x.B, x.H, x.W, x.C)
    planes = t.reshape(x.B, 1, x.H, x.W) if x.C == 1 else t.permute(0, 3, 1, 2).contiguous()
    r = upfirdn2d(planes, _device_kernel(kk, x.device), up=up, down=down, pad=pad)
    return View.from_nchw(r)


def padded_strided_conv(x: View, kk, pad, weights, ks, cout, bias, res, out_scale, out):
    """FIR `kk` with padding `pad` (an identity tap = pure zero padding), then a stride-2 pad-0 ks x ks conv whose
    epilogue adds bias / residual and rescales.  `weights(mfma)` returns the packed or direct-layout weights."""
    if x.C % 4 == 0 and x.C >= 8 and ks == 3 and x.H % 2 == 0 and x.W % 2 == 0:
        # matrix-core path: the stride-2 pad-0 conv of the padded (H+1)x(W+1) image is the odd-position subset
        # of the stride-1 pad-1 conv (4x the MFMA work, still ~5x faster than the direct kernel)
        xf, _ = ops.fir_nhwc(x, kk, 1, 1, pad)
        return ops.conv(xf, weights(True), 3, cout, mfma=True, bias=bias, res=res, out_scale=out_scale, out=out, sub2=True)
    if x.C % 4 == 0:
        xf, _ = ops.fir_nhwc(x, kk, 1, 1, pad)
    else:   # image pyramid with few channels: planes kernel
        assert x.ld == x.C
        t = x.base.reshape(x.B, x.H, x.W, x.C)
        planes = t.reshape(x.B, 1, x.H, x.W) if x.C == 1 else t.permute(0, 3, 1, 2).contiguous()
        xf = View.from_nchw(upfirdn2d(planes, _device_kernel(np.ascontiguousarray(kk, np.float32), x.device), pad=pad))
    return ops.conv(xf, weights(False), ks, cout, mfma=False, stride=2, pad=0, bias=bias, res=res, out_scale=out_scale, out=out)


class Conv2d(nn.Module):
    """Conv2d with optional FIR down-sampling (reference up_or_down_sampling.py:28-61).  `up=True` is
    dead code in the reference (its upsample_conv_2d raises on torch tensors)."""

    def __init__(self, in_ch, out_ch, kernel, up=False, down=False, resample_kernel=(1, 3, 3, 1), use_bias=True, kernel_init=None):
        super().__init__()
        assert not (up and down)
        assert kernel >= 1 and kernel % 2 == 1
        self.weight = nn.Parameter(torch.zeros(out_ch, in_ch, kernel, kernel))
        if kernel_init is not None:
            self.weight.data = kernel_init(self.weight.data.shape)
        if use_bias:
            self.bias = nn.Parameter(torch.zeros(out_ch))
        self.up, self.down, self.resample_kernel, self.kernel, self.use_bias = up, down, resample_kernel, kernel, use_bias
        self._prep = None

    def _weights(self, mfma=False):
        key = (self.weight._version, self.weight.data_ptr(), mfma)
        if self._prep is None or self._prep[0] != key:
            with torch.no_grad():
                self._prep = (key, ops.pack_conv_weight(self.weight) if mfma else ops.direct_weight(self.weight))
        return self._prep[1]

    def run(self, x: View, res: View = None, out_scale=1.0, out: View = None):
        """NHWC entry used by the generators; residual add and rescale fused into the conv epilogue."""
        if self.up:
            raise NotImplementedError('Conv2d(up=True) is unreachable in the reference (upsample_conv_2d raises)')
        bias = self.bias.detach() if self.use_bias else None
        if not self.down:
            return ops.conv(x, self._weights(), self.kernel, self.weight.shape[0], mfma=False, bias=bias, res=res, out_scale=out_scale, out=out)
        kk, up, down, pad = fir_params('conv_down', self.resample_kernel, conv_k=self.kernel)
        return padded_strided_conv(x, kk, pad, self._weights, self.kernel, self.weight.shape[0], bias, res, out_scale, out)

    def forward(self, x):
        return self.run(View.from_nchw(x)).to_nchw()
