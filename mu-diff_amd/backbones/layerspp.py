"""NCSN++ building blocks with AdaGN (reference backbones/layerspp.py), MI355X-native.

Every block keeps the reference's class name, constructor signature, parameter names/shapes and
`forward` signature (NCHW in, NCHW out), but owns no arithmetic: `forward` wraps `run`, which works on
NHWC views and enqueues fused HIP kernels (mudiff_hip.ops):

  ResnetBlockBigGANpp_Adagn  GN stats -> [FIR up/down of silu(adagn(x)) and x in one pass] ->
                             conv3x3(prologue = AdaGN affine + SiLU, epilogue = + bias + Dense_0(silu(temb)))
                             -> GN stats -> [1x1 skip conv] ->
                             conv3x3(prologue = AdaGN + SiLU, epilogue = + bias + skip, * 1/sqrt2)
  AttnBlockpp                GN stats -> fused q|k|v 1x1 GEMM (GN affine in the prologue) -> QK^T GEMM
                             (scale in the epilogue) -> row softmax -> PV GEMM -> NIN_3 GEMM (+x, *1/sqrt2)
  ConvFeatBlock / ConvBlock / ConvBlock_GAP   head blocks (C_in = 1 direct conv, GN, conv3x3)

The normalised / activated tensors are never written to HBM; concatenations are channel-slice views.
"""
import numpy as np
import torch
import torch.nn as nn

from mudiff_hip import ops
from mudiff_hip.ops import (ACT_NONE, ACT_SILU, INV_SQRT2, PRO_AFFINE, PRO_AFFINE_SILU, View)

from . import dense_layer, layers, up_or_down_sampling

conv1x1 = layers.ddpm_conv1x1
conv3x3 = layers.ddpm_conv3x3
NIN = layers.NIN
default_init = layers.default_init
dense = dense_layer.dense

FIR_K = (1, 3, 3, 1)


def _groups(c):
    return min(c // 4, 32)


def use_mfma(cin, cout):
    """Matrix-core implicit GEMM unless C_in is too small to feed it (the C_in == 1 head convolutions are
    pure store streams): those go to the exact direct kernel (as do the C_out <= 4 image-space output convs, which
    have their own LDS-tiled kernel there: ConvParam)."""
    return cin % 4 == 0 and cin >= 8


def _apply_affine(x: View, sc, sh):
    """Standalone normalise-only form (the blocks never materialise it): identity FIR with the affine
    prologue."""
    return ops.fir_nhwc(x, [[1.0]], 1, 1, (0, 0), pro=(sc, sh, PRO_AFFINE))[0]


class _Prepared:
    """Caches device-side re-packed weights; rebuilt when a parameter is replaced or modified."""

    def _key(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def prepared(self):
        key = self._key()
        cache = self.__dict__.get('_prep_cache')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, self._prepare())
            self.__dict__['_prep_cache'] = cache
        return cache[1]


class ConvParam:
    """A conv weight prepared for one of the two kernels."""

    def __init__(self, conv: nn.Conv2d = None, weight=None, bias=None):
        """conv: an nn.Conv2d; or weight [O,I,k,k] (+ bias) of a stride-1 'same' convolution (several convs merged into one)."""
        if conv is None:
            from types import SimpleNamespace
            conv = SimpleNamespace(weight=weight, bias=bias, stride=(1, 1), padding=(weight.shape[2] // 2,) * 2)
        w = conv.weight.detach()
        self.cout, self.cin, self.ks = w.shape[0], w.shape[1], w.shape[2]
        # C_out <= 4 3x3 convs (the image-space output / pyramid convs) have a dedicated exact kernel on the direct path
        tail = self.ks == 3 and self.cout <= 4 and self.cin % 16 == 0 and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
        self.mfma = use_mfma(self.cin, self.cout) and self.ks in (1, 3) and not tail
        self.w = ops.pack_conv_weight(w) if self.mfma else ops.direct_weight(w)
        self.bias = conv.bias.detach().contiguous() if conv.bias is not None else None
        self._w32, self._w8, self.w_exp = w, None, 0       # the fp32 weight stays referenced: other arithmetic plans are packed on first use

    def fp8x(self):
        """The operand packed for MUD_PREC_FP8X (fp16 hi planes + e4m3 images at this layer's own exponent), made on first use."""
        if self._w8 is None:
            self.w_exp = ops.fp8x_weight_exponent(self._w32)
            self._w8 = ops.pack_conv_weight(self._w32, prec=ops.PREC_FP8X, w_exp=self.w_exp)
        return self._w8

    def plan(self, x, pro=None, skip=None, sub2=False):
        """The arithmetic plan this launch would run with (ops.choose_prec)."""
        if not self.mfma or self.ks != 3:
            return ops.PREC_16X3
        return ops.choose_prec(x, self.cout, pro[2] if pro is not None else ops.PRO_NONE, skip=skip is not None, sub2=sub2)

    def __call__(self, x, **kw):
        if self.plan(x, kw.get('pro'), kw.get('skip'), kw.get('sub2', False)) == ops.PREC_FP8X:
            return ops.conv(x, self.fp8x(), self.ks, self.cout, mfma=True, bias=self.bias, prec=ops.PREC_FP8X, w_exp=self.w_exp, **kw)
        return ops.conv(x, self.w, self.ks, self.cout, mfma=self.mfma, bias=self.bias, **kw)


class AdaptiveGroupNorm(nn.Module):
    """gamma, beta = Linear(style); gamma * GroupNorm(x) + beta (reference layerspp.py:37-54).
    Inside the blocks only `style` is used (its output feeds the conv prologue); `forward` is the
    standalone NCHW form."""

    def __init__(self, num_groups, in_channel, style_dim):
        super().__init__()
        self.norm = nn.GroupNorm(num_groups, in_channel, affine=False, eps=1e-6)
        self.style = dense(style_dim, in_channel * 2)
        self.style.bias.data[:in_channel] = 1
        self.style.bias.data[in_channel:] = 0
        self.num_groups, self.in_channel = num_groups, in_channel

    def scale_shift(self, x: View, style_out):
        c = self.in_channel
        return ops.gn_lazy(x, self.num_groups, style_out[:, :c], style_out[:, c:])

    def forward(self, input, style):
        xv = View.from_nchw(input)
        st = ops.dense(style, self.style.weight.detach(), self.style.bias.detach())
        sc, sh = self.scale_shift(xv, st)
        return _apply_affine(xv, sc, sh).to_nchw()


class GroupNorm_Conv(nn.Module):
    """Affine-free GroupNorm (reference layerspp.py:56-65)."""

    def __init__(self, num_groups, in_channel):
        super().__init__()
        self.norm = nn.GroupNorm(num_groups, in_channel, affine=False, eps=1e-6)
        self.num_groups = num_groups

    def scale_shift(self, x: View):
        return ops.gn_lazy(x, self.num_groups)

    def forward(self, input):
        xv = View.from_nchw(input)
        sc, sh = self.scale_shift(xv)
        return _apply_affine(xv, sc, sh).to_nchw()


class AttnBlockpp(nn.Module, _Prepared):
    """Single-head self-attention over the H*W positions (reference layerspp.py:98-137)."""

    def __init__(self, channels, skip_rescale=False, init_scale=0.):
        super().__init__()
        self.GroupNorm_0 = nn.GroupNorm(num_groups=min(channels // 4, 32), num_channels=channels, eps=1e-6)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.skip_rescale = skip_rescale
        self.channels = channels

    def _prepare(self):
        wqkv = torch.cat([self.NIN_0.W, self.NIN_1.W, self.NIN_2.W], dim=1).contiguous()    # [C, 3C]
        return dict(wqkv=ops.pack_matrix_in_out(wqkv),
                    bqkv=torch.cat([self.NIN_0.b, self.NIN_1.b, self.NIN_2.b]).contiguous(),
                    wo=ops.pack_matrix_in_out(self.NIN_3.W), bo=self.NIN_3.b.detach().contiguous(),
                    gamma=self.GroupNorm_0.weight.detach().contiguous(), beta=self.GroupNorm_0.bias.detach().contiguous())

    def run(self, x: View, out: View = None):
        p = self.prepared()
        c, n = self.channels, x.H * x.W
        gn = ops.gn_lazy(x, self.GroupNorm_0.num_groups, p['gamma'], p['beta'])
        sc, sh = (gn, None) if isinstance(gn, ops.LazyGN) else gn
        qkv = ops.conv(x, p['wqkv'], 1, 3 * c, mfma=True, pro=(sc, sh, PRO_AFFINE), bias=p['bqkv'])     # [B,H,W,3C]
        if ops.attention_supported(c):      # fused flash-style kernel: the N x N score matrix is never materialised
            h = ops.attention(qkv, c, float(int(c) ** (-0.5)))
            return ops.conv(h, p['wo'], 1, c, mfma=True, bias=p['bo'], res=x, out_scale=INV_SQRT2 if self.skip_rescale else 1.0, out=out)
        q = View(qkv.base, x.B, 1, n, c, 3 * c, 0)
        # scores S[b,i,j] = q_i . k_j / sqrt(C): B operand = K rows ("co" = key index), per sample
        kp = ops.pack_weights(qkv.base, 0, 1, 3 * c, 1, c, n, nbatch=x.B, src_bstride=n * 3 * c, src_offset=c)
        s = ops.conv(q, kp, 1, n, mfma=True, out_scale=float(int(c) ** (-0.5)), w_bstride=kp.shape[1])   # [B,1,N,N]
        ops.softmax_rows_(s.base, n)
        # h[b,i,:] = sum_j P[i,j] v_j: B operand = V ("ci" = key index)
        vp = ops.pack_weights(qkv.base, 0, 3 * c, 1, 1, n, c, nbatch=x.B, src_bstride=n * 3 * c, src_offset=2 * c)
        h = ops.conv(s, vp, 1, c, mfma=True, w_bstride=vp.shape[1])                                      # [B,1,N,C]
        h = View(h.base, x.B, x.H, x.W, c)
        return ops.conv(h, p['wo'], 1, c, mfma=True, bias=p['bo'], res=x, out_scale=INV_SQRT2 if self.skip_rescale else 1.0, out=out)

    def forward(self, x):
        return self.run(View.from_nchw(x)).to_nchw()


class Upsample(nn.Module):
    """x2 up-sampling (reference layerspp.py:141-173).  Reachable in the reference only as the parameter-free image-pyramid
    up-sampler of progressive='output_skip' with fir=True (FIR x2).  The other forms are constructed like the reference
    does (same parameters) but cannot run there either: fir=False passes 'nearest' as F.interpolate's scale_factor (:164,
    ValueError) and fir=True/with_conv=True ends in upsample_conv_2d, which raises - `forward` raises the same way."""

    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if not fir:
            if with_conv:
                self.Conv_0 = conv3x3(in_ch, out_ch)
        elif with_conv:
            self.Conv2d_0 = up_or_down_sampling.Conv2d(in_ch, out_ch, kernel=3, up=True, resample_kernel=fir_kernel,
                                                       use_bias=True, kernel_init=default_init())
        self.fir, self.with_conv, self.fir_kernel, self.out_ch = fir, with_conv, fir_kernel, out_ch

    def run(self, x: View):
        assert self.fir and not self.with_conv
        return up_or_down_sampling.resample_view(x, 'up', True, self.fir_kernel)

    def forward(self, x):
        if not self.fir:
            raise ValueError("only one of size or scale_factor should be defined (the reference's Upsample(fir=False) fails "
                             'with this error at layerspp.py:164)')
        if self.with_conv:
            return self.Conv2d_0(x)          # raises, like the reference's upsample_conv_2d
        return up_or_down_sampling.upsample_2d(x, self.fir_kernel, factor=2)


class Downsample(nn.Module):
    """x2 down-sampling (reference layerspp.py:176-210).  with_conv=True is the input-pyramid branch of
    progressive_input='residual': FIR + strided 3x3 conv (fir=True) or zero-pad right/bottom + stride-2 3x3 conv
    (fir=False).  with_conv=False is the parameter-free pyramid of progressive_input='input_skip': FIR /2 or 2x2 mean."""

    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if not fir:
            if with_conv:
                self.Conv_0 = conv3x3(in_ch, out_ch, stride=2, padding=0)
        elif with_conv:
            self.Conv2d_0 = up_or_down_sampling.Conv2d(in_ch, out_ch, kernel=3, down=True, resample_kernel=fir_kernel,
                                                       use_bias=True, kernel_init=default_init())
        self.fir, self.fir_kernel, self.with_conv, self.out_ch = fir, fir_kernel, with_conv, out_ch
        self._prep = None

    def _naive_weights(self, mfma):
        w = self.Conv_0.weight
        key = (w._version, w.data_ptr(), mfma)
        if self._prep is None or self._prep[0] != key:
            with torch.no_grad():
                self._prep = (key, ops.pack_conv_weight(w) if mfma else ops.direct_weight(w))
        return self._prep[1]

    def run(self, x: View, res: View = None, out_scale=1.0, out: View = None):
        if self.with_conv and self.fir:
            return self.Conv2d_0.run(x, res=res, out_scale=out_scale, out=out)
        if self.with_conv:      # F.pad(x, (0,1,0,1)) + conv3x3(stride 2, padding 0): an identity "FIR" that only pads
            return up_or_down_sampling.padded_strided_conv(x, np.ones((1, 1), np.float32), (0, 1), self._naive_weights, 3,
                                                           self.Conv_0.weight.shape[0], self.Conv_0.bias.detach(), res, out_scale, out)
        assert res is None and out is None
        return up_or_down_sampling.resample_view(x, 'down', self.fir, self.fir_kernel)

    def forward(self, x):
        return self.run(View.from_nchw(x)).to_nchw()


class ResnetBlockBigGANpp_Adagn(nn.Module, _Prepared):
    """BigGAN-style residual block with two AdaGN layers (reference layerspp.py:261-324)."""

    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, zemb_dim=None, up=False, down=False, dropout=0.1, fir=False,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if not isinstance(act, nn.SiLU):
            raise NotImplementedError('only the SiLU activation of the reference generators is built')
        self.GroupNorm_0 = AdaptiveGroupNorm(min(in_ch // 4, 32), in_ch, zemb_dim)
        self.up, self.down, self.fir, self.fir_kernel = up, down, fir, fir_kernel
        self.Conv_0 = conv3x3(in_ch, out_ch)
        if temb_dim is not None:
            self.Dense_0 = nn.Linear(temb_dim, out_ch)
            self.Dense_0.weight.data = default_init()(self.Dense_0.weight.shape)
            nn.init.zeros_(self.Dense_0.bias)
        self.GroupNorm_1 = AdaptiveGroupNorm(min(out_ch // 4, 32), out_ch, zemb_dim)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = conv3x3(out_ch, out_ch, init_scale=init_scale)
        if in_ch != out_ch or up or down:
            self.Conv_2 = conv1x1(in_ch, out_ch)
        self.skip_rescale, self.act, self.in_ch, self.out_ch = skip_rescale, act, in_ch, out_ch
        self.dropout = dropout

    def _prepare(self):
        d = dict(c0=ConvParam(self.Conv_0), c1=ConvParam(self.Conv_1))
        if hasattr(self, 'Conv_2'):
            d['c2'] = ConvParam(self.Conv_2)
            if self.up:
                # FIR up-sampling and the 1x1 skip conv are both linear and commute: run the conv at LOW resolution (4x
                # fewer pixels) without its bias, up-sample the result, and add the bias (a constant) in Conv_1's epilogue
                d['c2'].bias = None
                d['c1'].bias = (self.Conv_1.bias.detach() + self.Conv_2.bias.detach()).contiguous()
        return d

    def run(self, x: View, style0, style1, tbias, out: View = None, arena=None):
        """style0 [B,2*in_ch] / style1 [B,2*out_ch]: outputs of GroupNorm_{0,1}.style(zemb);
        tbias [B,out_ch] or None: Dense_0(silu(temb)).  With an `arena`, Conv_0's epilogue also accumulates
        the GroupNorm_1 statistics (and `x.stats`, if its producer filled them, replaces the GroupNorm_0 pass)."""
        if self.dropout and self.training:
            raise NotImplementedError('dropout > 0 in training mode is not part of the inference path')
        p = self.prepared()
        fused = False
        gn0 = self.GroupNorm_0.scale_shift(x, style0)
        sc0, sh0 = (gn0, None) if isinstance(gn0, ops.LazyGN) else gn0
        if self.up:
            kk, up, down, pad = up_or_down_sampling.fir_params('up' if self.fir else 'naive_up', self.fir_kernel)
            h_in, _ = ops.fir_nhwc(x, kk, up, down, pad, pro=(sc0, sh0, PRO_AFFINE_SILU), want_h=True, want_x=False)
            x_skip, _ = ops.fir_nhwc(p['c2'](x), kk, up, down, pad)          # = Conv_2(FIR(x)) minus its bias (see _prepare)
            h = p['c0'](h_in, bias2=tbias, arena=arena)
        elif self.down:
            kk, up, down, pad = up_or_down_sampling.fir_params('down' if self.fir else 'naive_down', self.fir_kernel)
            h_in, x_skip = ops.fir_nhwc(x, kk, up, down, pad, pro=(sc0, sh0, PRO_AFFINE_SILU), want_h=True, want_x=True)
            h = p['c0'](h_in, bias2=tbias, arena=arena)
        else:
            x_skip = x
            fused = 'c2' in p and p['c0'].mfma and p['c2'].mfma and ops.fused_skip_ok(x, self.out_ch, PRO_AFFINE_SILU)
            if fused:       # Conv_0 and the 1x1 skip Conv_2 read the same x: one launch stages it once and writes both
                x_skip = View.empty(x.B, x.H, x.W, self.out_ch, x.device)
                h = p['c0'](x, pro=(sc0, sh0, PRO_AFFINE_SILU), bias2=tbias, arena=arena, skip=(p['c2'].w, p['c2'].bias, x_skip))
            else:
                h = p['c0'](x, pro=(sc0, sh0, PRO_AFFINE_SILU), bias2=tbias, arena=arena)
        gn1 = self.GroupNorm_1.scale_shift(h, style1)
        sc1, sh1 = (gn1, None) if isinstance(gn1, ops.LazyGN) else gn1
        if 'c2' in p and not self.up and not fused:
            x_skip = p['c2'](x_skip)
        return p['c1'](h, pro=(sc1, sh1, PRO_AFFINE_SILU), res=x_skip, out_scale=INV_SQRT2 if self.skip_rescale else 1.0, out=out)

    def forward(self, x, temb=None, zemb=None):
        st0 = ops.dense(zemb, self.GroupNorm_0.style.weight.detach(), self.GroupNorm_0.style.bias.detach())
        st1 = ops.dense(zemb, self.GroupNorm_1.style.weight.detach(), self.GroupNorm_1.style.bias.detach())
        tb = None
        if temb is not None:
            tb = ops.dense(temb, self.Dense_0.weight.detach(), self.Dense_0.bias.detach(), act_in=ACT_SILU)
        return self.run(View.from_nchw(x), st0, st1, tb).to_nchw()


class ConvFeatBlock(nn.Module, _Prepared):
    """conv3x3 -> GroupNorm -> SiLU -> conv3x3 (reference layerspp.py:394-423)."""

    def __init__(self, act, in_ch=None, out_ch=None, zemb_dim=256):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.group_norm = GroupNorm_Conv(min(out_ch // 4, 32), out_ch)
        self.act = act
        self.conv2 = conv3x3(out_ch, out_ch)

    def _prepare(self):
        return dict(c1=ConvParam(self.conv1), c2=ConvParam(self.conv2))

    def run(self, x: View, out: View = None, arena=None):
        p = self.prepared()
        h = p['c1'](x, arena=arena)
        gn = self.group_norm.scale_shift(h)
        sc, sh = (gn, None) if isinstance(gn, ops.LazyGN) else gn
        return p['c2'](h, pro=(sc, sh, PRO_AFFINE_SILU), out=out)

    def forward(self, x):
        return self.run(View.from_nchw(x)).to_nchw()


class ConvBlock(nn.Module, _Prepared):
    """conv3x3 -> AdaGN(style) -> SiLU -> conv3x3 (reference layerspp.py:426-455)."""

    def __init__(self, act, in_ch=None, out_ch=None, zemb_dim=256):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.group_norm = AdaptiveGroupNorm(min(out_ch // 4, 32), out_ch, zemb_dim)
        self.act = act
        self.conv2 = conv3x3(out_ch, out_ch)

    def _prepare(self):
        return dict(c1=ConvParam(self.conv1), c2=ConvParam(self.conv2))

    def run(self, x: View, style_out, out: View = None, arena=None, cache: dict = None):
        """`cache` (a dict owned by the generator's loop cache): conv1(x) and its GroupNorm sums depend on the condition image
        alone, so inside a sampling loop they are computed once; only the style-dependent half is redone."""
        p = self.prepared()
        if cache is not None and 'h' in cache:
            h = cache['h']
            if arena is not None:
                h.stats = arena.take(h.B, h.C)
                if h.stats is not None:
                    h.stats.copy_(cache['stats'])
        else:
            h = p['c1'](x, arena=arena)
            if cache is not None:
                cache.update(h=h, stats=None if h.stats is None else h.stats.clone())
        gn = self.group_norm.scale_shift(h, style_out)
        sc, sh = (gn, None) if isinstance(gn, ops.LazyGN) else gn
        return p['c2'](h, pro=(sc, sh, PRO_AFFINE_SILU), out=out)

    def forward(self, x, style=None):
        st = ops.dense(style, self.group_norm.style.weight.detach(), self.group_norm.style.bias.detach())
        return self.run(View.from_nchw(x), st).to_nchw()


class ConvBlock_GAP(nn.Module, _Prepared):
    """ConvFeatBlock -> global average pool -> dense: the pseudo-target style vector of G2
    (reference layerspp.py:458-501; its first-call debug print is not reproduced)."""

    def __init__(self, act, in_ch=None, out_ch=None, zemb_dim=256):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.group_norm = GroupNorm_Conv(min(out_ch // 4, 32), out_ch)
        self.act = act
        self.conv2 = conv3x3(out_ch, out_ch)
        self.adaptive_gap = nn.AdaptiveAvgPool2d(1)
        self.fc = dense(out_ch, zemb_dim)

    def _prepare(self):
        return dict(c1=ConvParam(self.conv1), c2=ConvParam(self.conv2))

    def run(self, x: View, arena=None):
        p = self.prepared()
        h = p['c1'](x, arena=arena)
        gn = self.group_norm.scale_shift(h)
        sc, sh = (gn, None) if isinstance(gn, ops.LazyGN) else gn
        h = p['c2'](h, pro=(sc, sh, PRO_AFFINE_SILU))
        gap = ops.channel_mean(h)
        assert gap.shape[1] == self.fc.in_features, f'GAP vector {gap.shape[1]} != fc.in_features {self.fc.in_features}'
        return ops.dense(gap, self.fc.weight.detach(), self.fc.bias.detach())

    def forward(self, x):
        return self.run(View.from_nchw(x))


class GaussianFourierProjection(nn.Module):
    """Gaussian Fourier embedding of noise levels (reference layerspp.py:68-77); W is a frozen parameter."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def run_log(self, t):
        """Embedding of log(t) - the form the generators use (ncsnpp_generator_adagn_feat.py:288-289) - in one kernel."""
        return ops.fourier_embedding(t, self.W)

    def forward(self, x):
        return ops.fourier_embedding(torch.exp(x.float()), self.W)


class Combine(nn.Module, _Prepared):
    """conv1x1 of the image pyramid, then sum with / concatenation to the feature map (reference layerspp.py:80-95)."""

    def __init__(self, dim1, dim2, method='cat'):
        super().__init__()
        self.Conv_0 = conv1x1(dim1, dim2)
        self.method = method
        if method not in ('cat', 'sum'):
            raise ValueError(f'Method {method} not recognized.')

    def _prepare(self):
        return dict(c=ConvParam(self.Conv_0))

    def run(self, x: View, y: View, out: View = None):
        """sum: out = conv(x) + y (one launch, y enters the epilogue).  cat: `out` is the [.., 2*dim2] concat buffer whose
        second half `y` already occupies (its producer wrote it there); the conv fills the first half."""
        p = self.prepared()
        if self.method == 'sum':
            return p['c'](x, res=y, out=out)
        d = self.Conv_0.weight.shape[0]
        if out is None:
            out = View.empty(y.B, y.H, y.W, 2 * d, y.device)
            out.slice(d, d).tensor().copy_(y.tensor())          # standalone use only (plumbing copy)
        else:
            assert y.base is out.base and y.c0 == out.c0 + d, 'Combine(cat): y must already live in the second half of `out`'
        p['c'](x, out=out.slice(0, d))
        return out

    def forward(self, x, y):
        return self.run(View.from_nchw(x), View.from_nchw(y)).to_nchw()


def _unbuildable(name, why):
    class _Unbuildable(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name}: {why}')
    _Unbuildable.__name__ = name
    return _Unbuildable


# The generators of the reference cannot be constructed with these block types (resblock_type='ddpm' /
# 'biggan_oneadagn' hit an UnboundLocalError on `ConvBlock`, ncsnpp_generator_adagn_feat.py:177-180), so no checkpoint and
# no caller can exist for them; tests/golden/make_golden.py records the failure.
ResnetBlockDDPMpp_Adagn = _unbuildable('ResnetBlockDDPMpp_Adagn', "unreachable: the reference's generators fail to construct with resblock_type='ddpm'")
ResnetBlockBigGANpp_Adagn_one = _unbuildable('ResnetBlockBigGANpp_Adagn_one', "unreachable: the reference's generators fail to construct with resblock_type='biggan_oneadagn'")
