"""The time-conditioned critic of MU-Diff (reference backbones/discriminator.py), inference forward on the
MI355X kernels: `Discriminator_large(nc, ngf, t_emb_dim, act)` / `Discriminator_small(...)` with the reference's
constructor signature, parameter names and `forward(x, t, x_t) -> (logit[B], mid_feat[B, 8*ngf, H/8, W/8])`
(Discriminator_small returns the logit only, like the reference).

DownConvBlock (reference :39-99) = lrelu -> conv3x3 (+ dense_t1(t_emb)) -> lrelu -> [FIR /2 of both branches] ->
conv3x3 ; skip = 1x1(no bias) ; (out + skip)/sqrt2 - here: LeakyReLU folded into the conv prologue / epilogue,
time-embedding bias, skip add and rescale into the epilogues, FIR on the gfx950 resampler.  The minibatch-stddev
channel (:246-254) is constant over space, so its contribution to `final_conv` is a Cin = 1 convolution that enters
the 512-channel matrix-core convolution as its residual instead of being concatenated.
Training (gradients, R1 penalty) is out of scope: forward only, under no_grad."""
import math

import numpy as np
import torch
import torch.nn as nn

from mudiff_hip import ops
from mudiff_hip.ops import ACT_LRELU, ACT_NONE, ACT_SIGMOID, INV_SQRT2, PRO_LRELU, View

from . import dense_layer, layers, layerspp, up_or_down_sampling

dense = dense_layer.dense
conv2d = dense_layer.conv2d
get_sinusoidal_positional_embedding = layers.get_timestep_embedding


def _check_act(act):
    if not (isinstance(act, nn.LeakyReLU) and abs(act.negative_slope - 0.2) < 1e-12):
        raise NotImplementedError('the MI355X critic implements the reference default act=nn.LeakyReLU(0.2) only')


class TimestepEmbedding(nn.Module):
    def __init__(self, embedding_dim, hidden_dim, output_dim, act=nn.LeakyReLU(0.2)):
        super().__init__()
        _check_act(act)
        self.embedding_dim, self.output_dim, self.hidden_dim = embedding_dim, output_dim, hidden_dim
        self.main = nn.Sequential(dense(embedding_dim, hidden_dim), act, dense(hidden_dim, output_dim))

    def forward(self, temp, act_out=ACT_NONE):
        temb = get_sinusoidal_positional_embedding(temp, self.embedding_dim)
        temb = ops.dense(temb, self.main[0].weight.detach(), self.main[0].bias.detach(), act_out=ACT_LRELU)
        return ops.dense(temb, self.main[2].weight.detach(), self.main[2].bias.detach(), act_out=act_out)


class DownConvBlock(nn.Module, layerspp._Prepared):
    def __init__(self, in_channel, out_channel, kernel_size=3, padding=1, t_emb_dim=128, downsample=False, act=nn.LeakyReLU(0.2),
                 fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        _check_act(act)
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError('DownConvBlock: only the 3x3 / pad 1 form used by the critics is built')
        self.fir_kernel, self.downsample = fir_kernel, downsample
        self.conv1 = nn.Sequential(conv2d(in_channel, out_channel, kernel_size, padding=padding))
        self.conv2 = nn.Sequential(conv2d(out_channel, out_channel, kernel_size, padding=padding, init_scale=0.))
        self.dense_t1 = dense(t_emb_dim, out_channel)
        self.act = act
        self.skip = nn.Sequential(conv2d(in_channel, out_channel, 1, padding=0, bias=False))

    def _prepare(self):
        return dict(c1=layerspp.ConvParam(self.conv1[0]), c2=layerspp.ConvParam(self.conv2[0]), sk=layerspp.ConvParam(self.skip[0]))

    def run(self, x: View, t_emb):
        p = self.prepared()
        tb = ops.dense(t_emb, self.dense_t1.weight.detach(), self.dense_t1.bias.detach())
        out = p['c1'](x, pro=(None, None, PRO_LRELU), bias2=tb, act=ACT_LRELU)      # act(conv1(act(x)) + dense_t1(t))
        if self.downsample:
            kk, up, down, pad = up_or_down_sampling.fir_params('down', self.fir_kernel)
            out, _ = ops.fir_nhwc(out, kk, up, down, pad)
            x, _ = ops.fir_nhwc(x, kk, up, down, pad)
        skip = p['sk'](x)
        return p['c2'](out, res=skip, out_scale=INV_SQRT2)

    def forward(self, input, t_emb):
        return self.run(View.from_nchw(input), t_emb).to_nchw()


class _DiscriminatorBase(nn.Module, layerspp._Prepared):
    def _prepare(self):
        c = self.final_conv.weight.shape[1] - 1
        w = self.final_conv.weight.detach()
        return dict(start=layerspp.ConvParam(self.start_conv),
                    fmain=ops.pack_conv_weight(w[:, :c].contiguous()) if layerspp.use_mfma(c, w.shape[0]) else ops.direct_weight(w[:, :c].contiguous()),
                    fmain_mfma=layerspp.use_mfma(c, w.shape[0]),
                    fstd=ops.direct_weight(w[:, c:].contiguous()), fbias=self.final_conv.bias.detach().contiguous())

    def _head(self, out: View, p):
        """minibatch stddev -> final_conv -> act -> spatial sum -> end_linear (reference :246-263)."""
        B, c = out.B, out.C
        group = min(B, self.stddev_group)
        if B % group:
            raise ValueError(f'batch {B} is not divisible by the minibatch-stddev group {group} (the reference view() fails too)')
        s = ops.minibatch_stddev(out, group)                                                   # [B]
        const = View(s.repeat_interleave(out.H * out.W).contiguous(), B, out.H, out.W, 1)      # the constant stddev plane (tiny)
        std_part = ops.conv(const, p['fstd'], 3, self.final_conv.weight.shape[0], mfma=False)  # its share of final_conv, incl. zero padding
        h = ops.conv(out, p['fmain'], 3, self.final_conv.weight.shape[0], mfma=p['fmain_mfma'], bias=p['fbias'], res=std_part, act=ACT_LRELU)
        pooled = ops.channel_mean(h)                                                           # mean; the sum is hw * mean
        hw = float(h.H * h.W)
        return ops.dense(pooled, (self.end_linear.weight.detach() * hw).contiguous(), self.end_linear.bias.detach()).view(-1)

    def _input(self, x, x_t):
        ops.require_gpu(x, x_t)
        return View.from_nchw(torch.cat((x.detach().float(), x_t.detach().float()), dim=1))    # 2-channel input image (plumbing copy)


class Discriminator_small(_DiscriminatorBase):
    """Reference :101-172 (CIFAR-sized critic; not instantiated by MU-Diff's engines, kept for API completeness)."""

    def __init__(self, nc=3, ngf=64, t_emb_dim=128, act=nn.LeakyReLU(0.2)):
        super().__init__()
        _check_act(act)
        self.act = act
        self.t_embed = TimestepEmbedding(embedding_dim=t_emb_dim, hidden_dim=t_emb_dim, output_dim=t_emb_dim, act=act)
        self.start_conv = conv2d(nc, ngf * 2, 1, padding=0)
        self.conv1 = DownConvBlock(ngf * 2, ngf * 2, t_emb_dim=t_emb_dim, act=act)
        self.conv2 = DownConvBlock(ngf * 2, ngf * 4, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv3 = DownConvBlock(ngf * 4, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv4 = DownConvBlock(ngf * 8, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.final_conv = conv2d(ngf * 8 + 1, ngf * 8, 3, padding=1, init_scale=0.)
        self.end_linear = dense(ngf * 8, 1)
        self.stddev_group, self.stddev_feat = 4, 1

    def forward(self, x, t, x_t):
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            p = self.prepared()
            t_embed = self.t_embed(t, act_out=ACT_LRELU)
            h = p['start'](self._input(x, x_t))
            for blk in (self.conv1, self.conv2, self.conv3, self.conv4):
                h = blk.run(h, t_embed)
            return self._head(h, p)


class Discriminator_large(_DiscriminatorBase):
    """Reference :175-263: six FIR-downsampling blocks; also returns conv3's output as `mid_feat` (the input of the
    uncertainty-map `att_conv` in engine/train.py:954-962)."""

    def __init__(self, nc=1, ngf=32, t_emb_dim=128, act=nn.LeakyReLU(0.2)):
        super().__init__()
        _check_act(act)
        self.act = act
        self.t_embed = TimestepEmbedding(embedding_dim=t_emb_dim, hidden_dim=t_emb_dim, output_dim=t_emb_dim, act=act)
        self.start_conv = conv2d(nc, ngf * 2, 1, padding=0)
        self.conv1 = DownConvBlock(ngf * 2, ngf * 4, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv2 = DownConvBlock(ngf * 4, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv3 = DownConvBlock(ngf * 8, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv4 = DownConvBlock(ngf * 8, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv5 = DownConvBlock(ngf * 8, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.conv6 = DownConvBlock(ngf * 8, ngf * 8, t_emb_dim=t_emb_dim, downsample=True, act=act)
        self.final_conv = conv2d(ngf * 8 + 1, ngf * 8, 3, padding=1)
        self.end_linear = dense(ngf * 8, 1)
        self.stddev_group, self.stddev_feat = 4, 1

    def forward(self, x, t, x_t):
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            p = self.prepared()
            t_embed = self.t_embed(t, act_out=ACT_LRELU)
            h = p['start'](self._input(x, x_t))
            h = self.conv1.run(h, t_embed)
            h = self.conv2.run(h, t_embed)
            mid = self.conv3.run(h, t_embed)
            h = self.conv4.run(mid, t_embed)
            h = self.conv5.run(h, t_embed)
            h = self.conv6.run(h, t_embed)
            return self._head(h, p), mid.to_nchw()


def uncertainty_map(att_conv, mid_feat, size):
    """Uncertainty / attention map of the critic's mid feature (reference engine/train.py:957-959, `att_conv =
    conv2d(64*8, 1, 1, padding=0)` at :466): sigmoid(att_conv(mid_feat)) up-sampled bilinearly (align_corners=False)
    to `size` = the image's (H, W).  -> [B,1,H,W].  The 1x1 conv + sigmoid run as one MFMA-conv launch, the resize as one
    HIP kernel."""
    ops.require_gpu(mid_feat)
    key = (att_conv.weight._version, att_conv.weight.data_ptr(), None if att_conv.bias is None else att_conv.bias._version)
    cache = att_conv.__dict__.get('_mud_param')
    if cache is None or cache[0] != key:
        cache = (key, layerspp.ConvParam(att_conv))
        att_conv.__dict__['_mud_param'] = cache
    with torch.no_grad(), torch.autocast('cuda', enabled=False):
        m = cache[1](View.from_nchw(mid_feat.detach().float()), act=ACT_SIGMOID)        # [B,h,w,1]
        return ops.resize_bilinear(m.to_nchw(), size)
