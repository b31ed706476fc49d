"""CPU oracle for MU-Diff's dual-generator reverse-diffusion sampling path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mu-diff_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
there only as the checker / timed CPU baseline.

What it is: a from-scratch, *functional* (state_dict-in, tensor-out) PyTorch-CPU fp32
restatement of the reference's hot path.  Every function cites the reference file:line it
follows (paths relative to the reference checkout).  Layout is the reference's: NCHW fp32.

Pinning: the reference has no tests and no golden vectors.  The oracle is
pinned by outputs of the reference itself run in the build container:
``tests/golden/make_golden.py`` imports the reference's ``backbones`` (safe-import recipe,
SURVEY.md section 8c), compares every function here against it (about 230 comparisons - schedules, posterior, FIR,
blocks, small and full-size models, 13 alternate configurations, critic, uncertainty map, volume-pipeline pieces - all
bit-exact on torch 2.10 CPU) and stores the reference's outputs as fixtures under ``tests/golden/``,
which ``tests/test_oracle_golden.py`` re-checks on every CPU run (the reference itself never
travels to the GPU box).
Third-party arithmetic (ATen CPU conv / group_norm / softmax) is whatever torch build is
installed (2.10.0 here; the reference pins 2.4.1) - "parity unpinned by the reference, pinned
by our goldens".
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = float(np.sqrt(2.0))


# --------------------------------------------------------------------------------------
# configuration (attribute bag read by the reference constructors,
# backbones/ncsnpp_generator_adagn_feat.py:59-84; defaults = BASELINE config 2 /
# demo.ipynb cell 3)
# --------------------------------------------------------------------------------------
def default_config(**overrides):
    cfg = dict(
        num_timesteps=4, beta_min=0.1, beta_max=20.0, centered=True, use_geometric=False,
        num_channels=1, num_channels_dae=64, n_mlp=3, ch_mult=[1, 2, 4], num_res_blocks=2,
        attn_resolutions=(16,), dropout=0.0, resamp_with_conv=True, conditional=True,
        fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan',
        progressive='none', progressive_input='residual', progressive_combine='sum',
        embedding_type='positional', fourier_scale=16.0, not_use_tanh=False,
        image_size=256, nz=100, z_emb_dim=256, t_emb_dim=256,
    )
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


def _check_supported(cfg):
    """What the reference itself can build and run (probed by tests/golden/make_golden.py::golden_variants)."""
    assert cfg.resblock_type.lower() == 'biggan', "the reference's own constructor fails for 'ddpm' / 'biggan_oneadagn'"
    assert cfg.progressive.lower() in ('none', 'output_skip'), "progressive='residual' fails inside the reference"
    assert cfg.progressive_input.lower() in ('residual', 'input_skip', 'none')
    assert cfg.progressive_combine.lower() in ('sum', 'cat')
    assert cfg.embedding_type.lower() in ('positional', 'fourier')
    assert list(cfg.fir_kernel) == [1, 3, 3, 1]
    assert cfg.fir or cfg.progressive.lower() == 'none', "Upsample(fir=False) raises in the reference (layerspp.py:164)"


# --------------------------------------------------------------------------------------
# L3: schedules and posterior (engine/test.py:48-177, engine/train.py:246-281)
# --------------------------------------------------------------------------------------
def var_func_vp(t, beta_min, beta_max):
    """engine/test.py:48-51."""
    log_mean_coeff = -0.25 * t ** 2 * (beta_max - beta_min) - 0.5 * t * beta_min
    return 1.0 - torch.exp(2.0 * log_mean_coeff)


def var_func_geometric(t, beta_min, beta_max):
    """engine/test.py:54-55."""
    return beta_min * ((beta_max / beta_min) ** t)


def get_time_schedule(cfg):
    """engine/test.py:66-72 (float64 linspace squeezed into [1e-3, 1])."""
    n = cfg.num_timesteps
    t = torch.from_numpy(np.arange(0, n + 1, dtype=np.float64) / n)
    return t * (1.0 - 1e-3) + 1e-3


def get_sigma_schedule(cfg):
    """engine/test.py:75-97: float64 variance -> betas, 1e-8 prepended, cast to fp32."""
    t = get_time_schedule(cfg)
    var = (var_func_geometric if cfg.use_geometric else var_func_vp)(t, cfg.beta_min, cfg.beta_max)
    alpha_bars = 1.0 - var
    betas = 1 - alpha_bars[1:] / alpha_bars[:-1]
    betas = torch.cat((torch.tensor(1e-8)[None], betas)).type(torch.float32)
    return betas ** 0.5, torch.sqrt(1 - betas), betas


class PosteriorCoefficients:
    """engine/test.py:101-123."""

    def __init__(self, cfg):
        _, _, betas = get_sigma_schedule(cfg)
        self.betas = betas.type(torch.float32)[1:]
        self.alphas = 1 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, 0)
        self.alphas_cumprod_prev = torch.cat((torch.ones(1, dtype=torch.float32), self.alphas_cumprod[:-1]), 0)
        self.posterior_variance = self.betas * (1 - self.alphas_cumprod_prev) / (1 - self.alphas_cumprod)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = torch.rsqrt(self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1 / self.alphas_cumprod - 1)
        self.posterior_mean_coef1 = self.betas * torch.sqrt(self.alphas_cumprod_prev) / (1 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1 - self.alphas_cumprod_prev) * torch.sqrt(self.alphas) / (1 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(self.posterior_variance.clamp(min=1e-20))


class DiffusionCoefficients:
    """engine/train.py:246-253."""

    def __init__(self, cfg):
        self.sigmas, self.a_s, _ = get_sigma_schedule(cfg)
        self.a_s_prev = self.a_s.clone()
        self.a_s_prev[-1] = 1
        self.a_s_cum = torch.cumprod(self.a_s, dim=0)
        self.sigmas_cum = torch.sqrt(1.0 - self.a_s_cum ** 2)


def extract(table, t, shape):
    """engine/test.py:58-63: gather + reshape to [B,1,1,...]."""
    return torch.gather(table, 0, t).reshape(shape[0], *([1] * (len(shape) - 1)))


def q_sample(coeff, x_start, t, noise):
    """engine/train.py:256-266 with the noise injected."""
    return extract(coeff.a_s_cum, t, x_start.shape) * x_start + extract(coeff.sigmas_cum, t, x_start.shape) * noise


def q_sample_pairs(coeff, x_start, t, noise_inner, noise_outer):
    """engine/train.py:269-281.  The reference draws `noise` (:276, here noise_outer) and then
    q_sample draws its own (:277, here noise_inner)."""
    x_t = q_sample(coeff, x_start, t, noise_inner)
    x_tp1 = extract(coeff.a_s, t + 1, x_start.shape) * x_t + extract(coeff.sigmas, t + 1, x_start.shape) * noise_outer
    return x_t, x_tp1


def sample_posterior(coef, x_0, x_t, t, noise):
    """engine/test.py:126-147 with the noise injected."""
    mean = extract(coef.posterior_mean_coef1, t, x_t.shape) * x_0 + extract(coef.posterior_mean_coef2, t, x_t.shape) * x_t
    log_var = extract(coef.posterior_log_variance_clipped, t, x_t.shape)
    nonzero = 1 - (t == 0).type(torch.float32)
    return mean + nonzero[:, None, None, None] * torch.exp(0.5 * log_var) * noise


def sample_posterior_combine(coef, x_0_1, x_0_2, x_t, t, noise):
    """engine/test.py:150-177 with the noise injected."""
    c1 = extract(coef.posterior_mean_coef1, t, x_t.shape)
    c2 = extract(coef.posterior_mean_coef2, t, x_t.shape)
    mean = ((c1 * x_0_1 + c2 * x_t) + (c1 * x_0_2 + c2 * x_t)) / 2
    log_var = extract(coef.posterior_log_variance_clipped, t, x_t.shape)
    nonzero = 1 - (t == 0).type(torch.float32)
    return mean + nonzero[:, None, None, None] * torch.exp(0.5 * log_var) * noise


# --------------------------------------------------------------------------------------
# L0/L1: FIR resampling (utils/op/upfirdn2d.py:201-242, backbones/up_or_down_sampling.py)
# --------------------------------------------------------------------------------------
def setup_kernel(k):
    """backbones/up_or_down_sampling.py:186-193."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """utils/op/upfirdn2d.py:170-181 / 201-242: zero-stuff by `up`, pad (negative = crop), true
    convolution with `kernel`, keep every `down`-th sample.  Same pad on both axes."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    z = x.new_zeros(n * c, 1, h * up, w * up)
    z[:, :, ::up, ::up] = x.reshape(n * c, 1, h, w)
    z = F.pad(z, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    z = F.conv2d(z, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw))
    z = z[:, :, ::down, ::down]
    return z.reshape(n, c, z.shape[2], z.shape[3])


def upsample_2d(x, k=(1, 3, 3, 1), factor=2, gain=1):
    """backbones/up_or_down_sampling.py:200-229."""
    kk = setup_kernel(k) * (gain * factor ** 2)
    p = kk.shape[0] - factor
    return upfirdn2d(x, torch.tensor(kk, dtype=x.dtype), up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k=(1, 3, 3, 1), factor=2, gain=1):
    """backbones/up_or_down_sampling.py:232-262."""
    kk = setup_kernel(k) * gain
    p = kk.shape[0] - factor
    return upfirdn2d(x, torch.tensor(kk, dtype=x.dtype), down=factor, pad=((p + 1) // 2, p // 2))


def conv_downsample_2d(x, w, k=(1, 3, 3, 1), factor=2, gain=1):
    """backbones/up_or_down_sampling.py:149-183: FIR then strided conv, padded once."""
    kk = setup_kernel(k) * gain
    p = (kk.shape[0] - factor) + (w.shape[-1] - 1)
    x = upfirdn2d(x, torch.tensor(kk, dtype=x.dtype), pad=((p + 1) // 2, p // 2))
    return F.conv2d(x, w, stride=factor, padding=0)


# --------------------------------------------------------------------------------------
# L1 blocks
# --------------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_positions=10000):
    """backbones/layers.py:465-479."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1))
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


def _conv(sd, p, x, padding=1):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], padding=padding)


def _groups(c):
    return min(c // 4, 32)


def z_transform(sd, z, n_mlp):
    """PixelNorm + mapping MLP, backbones/ncsnpp_generator_adagn_feat.py:44-49, 271-277."""
    h = z / torch.sqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(n_mlp + 1):
        h = F.silu(_lin(sd, f'z_transform.{1 + 2 * i}', h))
    return h


def adagn(sd, p, x, style):
    """AdaptiveGroupNorm, backbones/layerspp.py:37-54."""
    c = x.shape[1]
    s = _lin(sd, p + '.style', style)[:, :, None, None]
    gamma, beta = s[:, :c], s[:, c:]
    return gamma * F.group_norm(x, _groups(c), eps=1e-6) + beta


def nin(sd, p, x):
    """backbones/layers.py:496-505: per-pixel x @ W[in,out] + b."""
    return torch.einsum('bchw,cd->bdhw', x, sd[p + '.W']) + sd[p + '.b'][None, :, None, None]


def naive_upsample_2d(x, factor=2):
    """backbones/up_or_down_sampling.py:64-68: nearest-neighbour repeat."""
    return x.repeat_interleave(factor, dim=2).repeat_interleave(factor, dim=3)


def naive_downsample_2d(x, factor=2):
    """backbones/up_or_down_sampling.py:71-74: mean over factor x factor boxes."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h // factor, factor, w // factor, factor).mean(dim=(3, 5))


def resblock(sd, p, x, temb, zemb, up=False, down=False, skip_rescale=True, fir=True):
    """ResnetBlockBigGANpp_Adagn.forward, backbones/layerspp.py:292-324."""
    in_ch = x.shape[1]
    out_ch = sd[p + '.Conv_0.weight'].shape[0]
    h = F.silu(adagn(sd, p + '.GroupNorm_0', x, zemb))
    if up:
        h, x = (upsample_2d(h), upsample_2d(x)) if fir else (naive_upsample_2d(h), naive_upsample_2d(x))
    elif down:
        h, x = (downsample_2d(h), downsample_2d(x)) if fir else (naive_downsample_2d(h), naive_downsample_2d(x))
    h = _conv(sd, p + '.Conv_0', h)
    if temb is not None:
        h = h + _lin(sd, p + '.Dense_0', F.silu(temb))[:, :, None, None]
    h = F.silu(adagn(sd, p + '.GroupNorm_1', h, zemb))
    h = _conv(sd, p + '.Conv_1', h)
    if in_ch != out_ch or up or down:
        x = _conv(sd, p + '.Conv_2', x, padding=0)
    return (x + h) / SQRT2 if skip_rescale else x + h


def attn_block(sd, p, x, skip_rescale=True):
    """AttnBlockpp.forward, backbones/layerspp.py:111-137."""
    b, c, hh, ww = x.shape
    h = F.group_norm(x, _groups(c), sd[p + '.GroupNorm_0.weight'], sd[p + '.GroupNorm_0.bias'], eps=1e-6)
    q = nin(sd, p + '.NIN_0', h).reshape(b, c, hh * ww)
    k = nin(sd, p + '.NIN_1', h).reshape(b, c, hh * ww)
    v = nin(sd, p + '.NIN_2', h).reshape(b, c, hh * ww)
    w = torch.einsum('bcq,bck->bqk', q, k) * (int(c) ** (-0.5))
    w = F.softmax(w, dim=-1)
    h = torch.einsum('bqk,bck->bcq', w, v).reshape(b, c, hh, ww)
    h = nin(sd, p + '.NIN_3', h)
    return (x + h) / SQRT2 if skip_rescale else x + h


def conv_feat_block(sd, p, x):
    """ConvFeatBlock.forward, backbones/layerspp.py:410-423."""
    h = _conv(sd, p + '.conv1', x)
    h = F.silu(F.group_norm(h, _groups(h.shape[1]), eps=1e-6))
    return _conv(sd, p + '.conv2', h)


def conv_block(sd, p, x, style):
    """ConvBlock.forward (AdaGN with the pseudo-target style), backbones/layerspp.py:442-455."""
    h = _conv(sd, p + '.conv1', x)
    h = F.silu(adagn(sd, p + '.group_norm', h, style))
    return _conv(sd, p + '.conv2', h)


def conv_block_gap(sd, p, x):
    """ConvBlock_GAP.forward, backbones/layerspp.py:478-501."""
    h = conv_feat_block(sd, p, x)
    return _lin(sd, p + '.fc', h.mean(dim=(2, 3)))


def pyramid_downsample(sd, p, x, fir=True):
    """Downsample(fir, with_conv=True): FIR + strided conv (up_or_down_sampling.Conv2d(down=True), layerspp.py:196-210,
    up_or_down_sampling.py:50-61) or, with fir=False, zero-pad right/bottom by one and a stride-2 pad-0 conv (:199-202)."""
    if fir:
        return conv_downsample_2d(x, sd[p + '.Conv2d_0.weight']) + sd[p + '.Conv2d_0.bias'].reshape(1, -1, 1, 1)
    return F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[p + '.Conv_0.weight'], sd[p + '.Conv_0.bias'], stride=2, padding=0)


def plain_downsample(x, fir=True):
    """Downsample(with_conv=False) (layerspp.py:203-207): FIR /2 or 2x2 average pooling."""
    return downsample_2d(x) if fir else F.avg_pool2d(x, 2, stride=2)


def plain_upsample(x, fir=True):
    """Upsample(with_conv=False, fir=True) (layerspp.py:167-169).  The fir=False branch of the reference (:164) passes
    'nearest' as F.interpolate's scale_factor and raises, so progressive='output_skip' exists with fir=True only."""
    assert fir, "Upsample(fir=False) raises in the reference (layerspp.py:164)"
    return upsample_2d(x)


def combine(sd, p, x, y, method):
    """Combine.forward (layerspp.py:80-95): conv1x1 of the image pyramid, then sum or channel concat."""
    h = _conv(sd, p + '.Conv_0', x, padding=0)
    return torch.cat([h, y], dim=1) if method == 'cat' else h + y


def fourier_embedding(sd, p, x):
    """GaussianFourierProjection.forward (layerspp.py:75-77) of log(time_cond) (ncsnpp_generator_adagn_feat.py:288-289)."""
    x_proj = torch.log(x)[:, None] * sd[p + '.W'][None, :] * 2 * np.pi
    return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


# --------------------------------------------------------------------------------------
# L2: generator layout + forward
# --------------------------------------------------------------------------------------
def build_plan(cfg, which, n_cond=3):
    """Walks the reference constructors (backbones/ncsnpp_generator_adagn_feat.py:86-269 for G1, :486-684 for G2;
    n_cond=2: the two-condition twins in ncsnpp_generator_adagn_feat_healthy.py) and returns the ordered module list as
    dicts.  Covers every configuration the reference itself can construct and run (tests/golden/make_golden.py probes
    them): embedding_type positional|fourier, conditional, progressive none|output_skip, progressive_input
    residual|input_skip(sum|cat)|none, fir True|False, skip_rescale, any nf / ch_mult / num_res_blocks / attn_resolutions.
    resblock_type 'ddpm' / 'biggan_oneadagn' and progressive='residual' raise inside the reference's own constructor /
    forward (UnboundLocalError at :180 / ValueError in its Conv2d(up=True)), so there is nothing to restate."""
    _check_supported(cfg)
    nf, ch_mult, nrb = cfg.num_channels_dae, list(cfg.ch_mult), cfg.num_res_blocks
    nres = len(ch_mult)
    res = [cfg.image_size // (2 ** i) for i in range(nres)]
    attn_res = tuple(int(a) for a in cfg.attn_resolutions)
    prog, pin, comb = cfg.progressive.lower(), cfg.progressive_input.lower(), cfg.progressive_combine.lower()
    mods = []
    embed_dim = nf
    if cfg.embedding_type.lower() == 'fourier':
        mods.append(dict(kind='fourier', n=nf))
        embed_dim = 2 * nf
    if cfg.conditional:
        mods += [dict(kind='linear', cin=embed_dim, cout=nf * 4), dict(kind='linear', cin=nf * 4, cout=nf * 4)]
    ch = cfg.num_channels
    if which == 'g1':
        mods += [dict(kind='feat', cin=ch, cout=nf) for _ in range(1 + n_cond)]
        head_c = nf * (1 + n_cond)
    else:
        mods += [dict(kind='gap', cin=ch, cout=nf), dict(kind='feat', cin=ch, cout=nf)]
        mods += [dict(kind='ada', cin=ch, cout=nf) for _ in range(n_cond)]
        head_c = nf * (4 if n_cond == 3 else 2)
    hs_c = [head_c]
    in_ch = head_c
    pyr_ch = ch
    for lvl in range(nres):
        for _ in range(nrb):
            out_ch = nf * ch_mult[lvl]
            mods.append(dict(kind='res', cin=in_ch, cout=out_ch, up=False, down=False, stage='down', level=lvl))
            in_ch = out_ch
            if res[lvl] in attn_res:
                mods.append(dict(kind='attn', c=in_ch, stage='down'))
            hs_c.append(in_ch)
        if lvl != nres - 1:
            mods.append(dict(kind='res', cin=in_ch, cout=in_ch, up=False, down=True, stage='downsample', level=lvl))
            if pin == 'input_skip':
                mods.append(dict(kind='combine', cin=pyr_ch, cout=in_ch, method=comb))
                if comb == 'cat':
                    in_ch *= 2
            elif pin == 'residual':
                mods.append(dict(kind='pyr', cin=pyr_ch, cout=in_ch))
                pyr_ch = in_ch
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(dict(kind='res', cin=in_ch, cout=in_ch, up=False, down=False, stage='mid'))
    mods.append(dict(kind='attn', c=in_ch, stage='mid'))
    mods.append(dict(kind='res', cin=in_ch, cout=in_ch, up=False, down=False, stage='mid'))
    for lvl in reversed(range(nres)):
        for _ in range(nrb + 1):
            out_ch = nf * ch_mult[lvl]
            mods.append(dict(kind='res', cin=in_ch + hs_c.pop(), cout=out_ch, up=False, down=False, stage='up', level=lvl))
            in_ch = out_ch
        if res[lvl] in attn_res:
            mods.append(dict(kind='attn', c=in_ch, stage='up'))
        if prog == 'output_skip':
            mods.append(dict(kind='gn', c=in_ch, stage='pyramid', first=(lvl == nres - 1)))
            mods.append(dict(kind='conv', cin=in_ch, cout=ch, stage='pyramid'))
        if lvl != 0:
            mods.append(dict(kind='res', cin=in_ch, cout=in_ch, up=True, down=False, stage='upsample', level=lvl))
    assert not hs_c
    if prog != 'output_skip':
        mods.append(dict(kind='gn', c=in_ch, stage='tail'))
        mods.append(dict(kind='conv', cin=in_ch, cout=ch, stage='tail'))
    for i, m in enumerate(mods):
        m['idx'] = i
    return mods


def param_spec(cfg, which, n_cond=3):
    """name -> shape, in the reference's state_dict() order (checked against the reference's own modules by
    tests/golden/make_golden.py for every configuration it records)."""
    spec = OrderedDict()
    zd = cfg.z_emb_dim
    nf = cfg.num_channels_dae
    if which == 'g2':
        pairs = ('c12', 'c23', 'c31') if n_cond == 3 else ('c12',)
        for n in ('feat_weight_c1', 'feat_weight_c2', 'feat_weight_c3')[:len(pairs)]:
            spec[n + '.weight'] = (nf, nf, 3, 3)
            spec[n + '.bias'] = (nf,)
        for pair in pairs:
            for a in ('feat_att1_', 'feat_att2_'):
                spec[a + pair + '.weight'] = (nf, n_cond * nf, 3, 3)
                spec[a + pair + '.bias'] = (nf,)

    def conv(p, cin, cout, k=3):
        spec[p + '.weight'] = (cout, cin, k, k)
        spec[p + '.bias'] = (cout,)

    def lin(p, cin, cout):
        spec[p + '.weight'] = (cout, cin)
        spec[p + '.bias'] = (cout,)

    for m in build_plan(cfg, which, n_cond):
        p = f"all_modules.{m['idx']}"
        k = m['kind']
        if k == 'fourier':
            spec[p + '.W'] = (m['n'],)
        elif k == 'linear':
            lin(p, m['cin'], m['cout'])
        elif k in ('feat', 'gap', 'ada'):
            conv(p + '.conv1', m['cin'], m['cout'])
            if k == 'ada':   # ConvBlock/ConvBlock_GAP keep their own default zemb_dim=256 (layerspp.py:427,459)
                lin(p + '.group_norm.style', 256, 2 * m['cout'])
            conv(p + '.conv2', m['cout'], m['cout'])
            if k == 'gap':
                lin(p + '.fc', m['cout'], 256)
        elif k == 'res':
            lin(p + '.GroupNorm_0.style', zd, 2 * m['cin'])
            conv(p + '.Conv_0', m['cin'], m['cout'])
            lin(p + '.Dense_0', nf * 4, m['cout'])
            lin(p + '.GroupNorm_1.style', zd, 2 * m['cout'])
            conv(p + '.Conv_1', m['cout'], m['cout'])
            if m['cin'] != m['cout'] or m['up'] or m['down']:
                conv(p + '.Conv_2', m['cin'], m['cout'], k=1)
        elif k == 'attn':
            spec[p + '.GroupNorm_0.weight'] = (m['c'],)
            spec[p + '.GroupNorm_0.bias'] = (m['c'],)
            for i in range(4):
                spec[p + f'.NIN_{i}.W'] = (m['c'], m['c'])
                spec[p + f'.NIN_{i}.b'] = (m['c'],)
        elif k == 'pyr':     # Downsample(with_conv=True): FIR form holds Conv2d_0, the naive form Conv_0 (layerspp.py:181-191)
            conv(p + ('.Conv2d_0' if cfg.fir else '.Conv_0'), m['cin'], m['cout'])
        elif k == 'combine':
            conv(p + '.Conv_0', m['cin'], m['cout'], k=1)
        elif k == 'gn':
            spec[p + '.weight'] = (m['c'],)
            spec[p + '.bias'] = (m['c'],)
        elif k == 'conv':
            conv(p, m['cin'], m['cout'])
    lin('z_transform.1', cfg.nz, zd)
    for i in range(cfg.n_mlp):
        lin(f'z_transform.{3 + 2 * i}', zd, zd)
    return spec


def make_state_dict(cfg, which, seed=1234, n_cond=3):
    """Weights-from-seed scheme owned by the build: every tensor is
    drawn from its own CPU generator keyed by (seed, which, name), at fan-avg scale 1 - including
    the tensors the reference initialises with init_scale=0 (Conv_1, NIN_3, final conv), which
    would otherwise make every parity check vacuous - biases are perturbed, norm gains ~1."""
    sd = OrderedDict()
    for name, shape in param_spec(cfg, which, n_cond).items():
        g = torch.Generator().manual_seed((zlib.crc32(f'{which}:{name}'.encode()) + 7919 * seed) % (2 ** 31))
        if len(shape) >= 2:
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            fan_avg = (shape[0] + shape[1]) * rf / 2.0
            bound = math.sqrt(3.0 / fan_avg)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        else:
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
            if name.endswith('.W'):                # GaussianFourierProjection: randn * scale (layerspp.py:73)
                t = t * (10.0 * cfg.fourier_scale)
            elif name.endswith('style.bias'):
                t[: shape[0] // 2] += 1.0          # gamma half (layerspp.py:44)
            elif name.endswith('.weight'):
                t += 1.0                           # GroupNorm affine gain
        sd[name] = t
    return sd


def _trunk(sd, cfg, plan, start, hs0, x_in, temb, zemb):
    """Shared down/mid/up trunk, backbones/ncsnpp_generator_adagn_feat.py:335-447."""
    sr, fir = cfg.skip_rescale, cfg.fir
    pin = cfg.progressive_input.lower()
    hs = [hs0]
    pyr = x_in if pin != 'none' else None
    out_pyr = None                      # progressive='output_skip' image pyramid (:389-417)
    i = start
    n = len(plan)
    h = None
    while i < n:
        m = plan[i]
        p = f"all_modules.{m['idx']}"
        k = m['kind']
        if k == 'res' and m['stage'] == 'down':
            h = resblock(sd, p, hs[-1], temb, zemb, skip_rescale=sr, fir=fir)
            if plan[i + 1]['kind'] == 'attn' and plan[i + 1]['stage'] == 'down':
                i += 1
                h = attn_block(sd, f"all_modules.{plan[i]['idx']}", h, sr)
            hs.append(h)
        elif k == 'res' and m['stage'] == 'downsample':
            h = resblock(sd, p, hs[-1], temb, zemb, down=True, skip_rescale=sr, fir=fir)
            if pin == 'input_skip':
                i += 1
                pyr = plain_downsample(pyr, fir)
                h = combine(sd, f"all_modules.{plan[i]['idx']}", pyr, h, plan[i]['method'])
            elif pin == 'residual':
                i += 1
                pyr = pyramid_downsample(sd, f"all_modules.{plan[i]['idx']}", pyr, fir)
                pyr = (pyr + h) / SQRT2 if sr else pyr + h
                h = pyr
            hs.append(h)
        elif k == 'res' and m['stage'] == 'mid':
            h = resblock(sd, p, h, temb, zemb, skip_rescale=sr, fir=fir)   # first one: h is hs[-1] (:370)
        elif k == 'attn':
            h = attn_block(sd, p, h, sr)
        elif k == 'res' and m['stage'] == 'up':
            h = resblock(sd, p, torch.cat([h, hs.pop()], dim=1), temb, zemb, skip_rescale=sr, fir=fir)
        elif k == 'res' and m['stage'] == 'upsample':
            h = resblock(sd, p, h, temb, zemb, up=True, skip_rescale=sr, fir=fir)
        elif k == 'gn' and m['stage'] == 'pyramid':
            c = h.shape[1]
            ph = F.silu(F.group_norm(h, _groups(c), sd[p + '.weight'], sd[p + '.bias'], eps=1e-6))
            i += 1
            ph = _conv(sd, f"all_modules.{plan[i]['idx']}", ph)
            out_pyr = ph if m['first'] else plain_upsample(out_pyr, fir) + ph
        elif k == 'gn':
            assert not hs
            c = h.shape[1]
            h = F.silu(F.group_norm(h, _groups(c), sd[p + '.weight'], sd[p + '.bias'], eps=1e-6))
        elif k == 'conv':
            h = _conv(sd, p, h)
        else:
            raise AssertionError(m)
        i += 1
    if cfg.progressive.lower() == 'output_skip':
        h = out_pyr
    return h if cfg.not_use_tanh else torch.tanh(h)


def _embeddings(sd, cfg, t, z):
    zemb = z_transform(sd, z, cfg.n_mlp)
    i = 0
    if cfg.embedding_type.lower() == 'fourier':
        temb = fourier_embedding(sd, 'all_modules.0', t)
        i = 1
    else:
        temb = timestep_embedding(t, cfg.num_channels_dae)
    if not cfg.conditional:
        return None, zemb, i
    temb = _lin(sd, f'all_modules.{i}', temb)
    temb = _lin(sd, f'all_modules.{i + 1}', F.silu(temb))
    return temb, zemb, i + 2


def g1_forward(sd, cfg, x, c1, c2, c3, t, z):
    """NCSNpp.forward, backbones/ncsnpp_generator_adagn_feat.py:279-447 (c3=None: the two-condition variant,
    ncsnpp_generator_adagn_feat_healthy.py:279-445)."""
    conds = (c1, c2) if c3 is None else (c1, c2, c3)
    plan = build_plan(cfg, 'g1', len(conds))
    temb, zemb, m0 = _embeddings(sd, cfg, t, z)
    if not cfg.centered:
        x = 2 * x - 1.0
    feats = [conv_feat_block(sd, f'all_modules.{m0 + j}', v) for j, v in enumerate((x,) + conds)]
    return _trunk(sd, cfg, plan, m0 + len(feats), torch.cat(feats, dim=1), x, temb, zemb)


def g2_forward(sd, cfg, x, c1, c2, c3, t, z, pseudo_target):
    """NCSNpp_adaptive.forward, backbones/ncsnpp_generator_adagn_feat.py:694-905 (c3=None: the two-condition variant,
    ncsnpp_generator_adagn_feat_healthy.py:693-873, one fused pair)."""
    conds = (c1, c2) if c3 is None else (c1, c2, c3)
    plan = build_plan(cfg, 'g2', len(conds))
    temb, zemb, m0 = _embeddings(sd, cfg, t, z)
    if not cfg.centered:
        x = 2 * x - 1.0
    style = conv_block_gap(sd, f'all_modules.{m0}', pseudo_target)
    xf = conv_feat_block(sd, f'all_modules.{m0 + 1}', x)
    f = [conv_block(sd, f'all_modules.{m0 + 2 + j}', c, style) for j, c in enumerate(conds)]
    cat = torch.cat(f, dim=1)

    def gate(name):
        return torch.sigmoid(_conv(sd, name, cat))

    def fuse(pair, wname, fa, fb):          # :769-788
        g1, g2 = gate('feat_att1_' + pair), gate('feat_att2_' + pair)
        att = _conv(sd, wname, g1 * fa)
        return g2 * att + (1 - g2) * fb

    fused = [fuse('c12', 'feat_weight_c1', f[0], f[1])]
    if len(conds) == 3:
        fused += [fuse('c23', 'feat_weight_c2', f[1], f[2]), fuse('c31', 'feat_weight_c3', f[2], f[0])]
    return _trunk(sd, cfg, plan, m0 + 2 + len(conds), torch.cat([xf] + fused, dim=1), x, temb, zemb)


def sample_from_model(coef, sd1, sd2, cfg, c1, c2, c3, x_init, zs, noises, return_steps=False):
    """engine/test.py:180-199 with the per-step latent z and posterior noise injected
    (zs[k], noises[k] for the k-th executed step, i.e. i = n_time-1-k)."""
    x = x_init
    steps = []
    n_time = cfg.num_timesteps
    with torch.no_grad():
        for k, i in enumerate(reversed(range(n_time))):
            t = torch.full((x.size(0),), i, dtype=torch.int64)
            x01 = g1_forward(sd1, cfg, x, c1, c2, c3, t, zs[k])
            x02 = g2_forward(sd2, cfg, x, c1, c2, c3, t, zs[k], x01[:, [0], :])
            x = sample_posterior_combine(coef, x01[:, [0], :], x02[:, [0], :], x, t, noises[k])
            steps.append((x01, x02, x))
    return (x, steps) if return_steps else x


# --------------------------------------------------------------------------------------
# the time-conditioned critic, inference forward
# --------------------------------------------------------------------------------------
def _lrelu(x):
    return F.leaky_relu(x, 0.2)


def down_conv_block(sd, p, x, t_emb, downsample):
    """DownConvBlock.forward, backbones/discriminator.py:76-99."""
    out = _lrelu(x)
    out = F.conv2d(out, sd[p + '.conv1.0.weight'], sd[p + '.conv1.0.bias'], padding=1)
    out = out + _lin(sd, p + '.dense_t1', t_emb)[..., None, None]
    out = _lrelu(out)
    if downsample:
        out, x = downsample_2d(out), downsample_2d(x)
    out = F.conv2d(out, sd[p + '.conv2.0.weight'], sd[p + '.conv2.0.bias'], padding=1)
    skip = F.conv2d(x, sd[p + '.skip.0.weight'])
    return (out + skip) / SQRT2


def discriminator_large_forward(sd, x, t, x_t, t_emb_dim, stddev_group=4):
    """Discriminator_large.forward, backbones/discriminator.py:215-263 -> (logit [B], mid_feat)."""
    temb = timestep_embedding(t, t_emb_dim)
    temb = _lin(sd, 't_embed.main.2', _lrelu(_lin(sd, 't_embed.main.0', temb)))
    temb = _lrelu(temb)
    h = F.conv2d(torch.cat((x, x_t), dim=1), sd['start_conv.weight'], sd['start_conv.bias'])
    h = down_conv_block(sd, 'conv1', h, temb, True)
    h = down_conv_block(sd, 'conv2', h, temb, True)
    mid = down_conv_block(sd, 'conv3', h, temb, True)
    h = down_conv_block(sd, 'conv4', mid, temb, True)
    h = down_conv_block(sd, 'conv5', h, temb, True)
    out = down_conv_block(sd, 'conv6', h, temb, True)
    b, c, hh, ww = out.shape
    group = min(b, stddev_group)
    sdv = out.view(group, -1, 1, c, hh, ww)
    sdv = torch.sqrt(sdv.var(0, unbiased=False) + 1e-8)
    sdv = sdv.mean([2, 3, 4], keepdims=True).squeeze(2).repeat(group, 1, hh, ww)
    out = torch.cat([out, sdv], 1)
    out = _lrelu(F.conv2d(out, sd['final_conv.weight'], sd['final_conv.bias'], padding=1))
    out = out.view(b, out.shape[1], -1).sum(2)
    return _lin(sd, 'end_linear', out).view(-1), mid


def discriminator_param_spec(nc, ngf, t_emb_dim):
    spec = OrderedDict()
    spec['t_embed.main.0.weight'], spec['t_embed.main.0.bias'] = (t_emb_dim, t_emb_dim), (t_emb_dim,)
    spec['t_embed.main.2.weight'], spec['t_embed.main.2.bias'] = (t_emb_dim, t_emb_dim), (t_emb_dim,)
    spec['start_conv.weight'], spec['start_conv.bias'] = (ngf * 2, nc, 1, 1), (ngf * 2,)
    for i, (ci, co) in enumerate(((2, 4), (4, 8), (8, 8), (8, 8), (8, 8), (8, 8)), start=1):
        pfx = f'conv{i}'
        spec[pfx + '.conv1.0.weight'], spec[pfx + '.conv1.0.bias'] = (ngf * co, ngf * ci, 3, 3), (ngf * co,)
        spec[pfx + '.conv2.0.weight'], spec[pfx + '.conv2.0.bias'] = (ngf * co, ngf * co, 3, 3), (ngf * co,)
        spec[pfx + '.dense_t1.weight'], spec[pfx + '.dense_t1.bias'] = (ngf * co, t_emb_dim), (ngf * co,)
        spec[pfx + '.skip.0.weight'] = (ngf * co, ngf * ci, 1, 1)
    spec['final_conv.weight'], spec['final_conv.bias'] = (ngf * 8, ngf * 8 + 1, 3, 3), (ngf * 8,)
    spec['end_linear.weight'], spec['end_linear.bias'] = (1, ngf * 8), (1,)
    return spec


def make_discriminator_state_dict(nc, ngf, t_emb_dim, seed=1234):
    """Same per-name seeding rule as make_state_dict (conv2 is init_scale=0 in the reference: re-drawn at scale 1)."""
    sd = OrderedDict()
    for name, shape in discriminator_param_spec(nc, ngf, t_emb_dim).items():
        g = torch.Generator().manual_seed((zlib.crc32(f'd:{name}'.encode()) + 7919 * seed) % (2 ** 31))
        if len(shape) >= 2:
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            bound = math.sqrt(3.0 / ((shape[0] + shape[1]) * rf / 2.0))
            sd[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        else:
            sd[name] = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return sd


# --------------------------------------------------------------------------------------
# metrics (tools/metric_calc.py:40-47; skimage is not installed: restated from the published
# definitions, "parity unpinned" - the same code scores both the build and the oracle)
# --------------------------------------------------------------------------------------
def psnr(gt, pred, data_range=1.0):
    mse = np.mean((np.asarray(gt, np.float64) - np.asarray(pred, np.float64)) ** 2)
    return float(10.0 * np.log10(data_range ** 2 / mse))


def ssim(gt, pred, data_range=1.0, win=7, k1=0.01, k2=0.03):
    """skimage.metrics.structural_similarity defaults: 7x7 uniform window, sample covariance,
    3-pixel border cropped, mean over the map."""
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(gt, np.float64), np.asarray(pred, np.float64)
    n = win * win
    cov_norm = n / (n - 1.0)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    uxx, uyy, uxy = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


# --------------------------------------------------------------------------------------
# row f1: uncertainty map of the critic's mid feature (engine/train.py:466,957-962)
# --------------------------------------------------------------------------------------
def resize_bilinear(x, size):
    """`F.interpolate(x, size=size, mode='bilinear', align_corners=False)` - the call the reference makes at
    engine/train.py:959 and engine/test_volume.py:274.  Half-pixel centres: src = (dst + 0.5) * in/out - 0.5, clamped
    at 0; the two neighbours are floor(src) and min(floor(src)+1, in-1)."""
    return F.interpolate(x, size=tuple(size), mode='bilinear', align_corners=False)


def uncertainty_map(mid_feat, att_w, att_b, size):
    """sigmoid(conv1x1(mid_feat)) up-sampled bilinearly to the image size (engine/train.py:957-959; `att_conv` is
    `conv2d(64*8, 1, 1, padding=0)`, :466)."""
    return resize_bilinear(torch.sigmoid(F.conv2d(mid_feat, att_w, att_b)), size)


# --------------------------------------------------------------------------------------
# row f3: volume pipeline (engine/test_volume.py:135-191, 236-300)
# --------------------------------------------------------------------------------------
def robust_minmax_to_minus1_1(vol, mask=None, pmin=1.0, pmax=99.0):
    """engine/test_volume.py:135-157: percentiles [pmin, pmax] over the non-zero (or masked, non-NaN) voxels, linear map to
    [0,1] with clipping, then to [-1,1].  Degenerate inputs (no voxels / flat) give zeros."""
    data = np.asarray(vol).astype(np.float32, copy=False)
    sel = (data != 0) if mask is None else (np.asarray(mask).astype(bool) & ~np.isnan(data))
    if not sel.any():
        return np.zeros_like(data, dtype=np.float32)
    vals = data[sel]
    lo, hi = np.percentile(vals, pmin), np.percentile(vals, pmax)
    if not (np.isfinite(lo) and np.isfinite(hi)) or hi <= lo:
        lo, hi = float(vals.min()), float(vals.max())
        if hi <= lo:
            return np.zeros_like(data, dtype=np.float32)
    return np.clip((data - lo) / (hi - lo), 0.0, 1.0) * 2.0 - 1.0


def extract_center_slices(volume, half_range):
    """engine/test_volume.py:159-168: axial slices [c - half_range, c + half_range] around c = Z // 2, clipped to the volume."""
    z = volume.shape[2]
    c = z // 2
    s0, s1 = max(0, c - half_range), min(z - 1, c + half_range)
    return [volume[:, :, k] for k in range(s0, s1 + 1)], s0, s1


def reconstruct_volume_from_slices(slices, shape, s0, s1):
    """engine/test_volume.py:170-181: predicted slices go back to planes s0.., everything else stays zero."""
    vol = np.zeros(shape, dtype=np.float32)
    for i, sl in enumerate(slices):
        k = s0 + i
        if k <= s1 and k < shape[2]:
            vol[:, :, k] = np.asarray(sl, dtype=np.float32)
    return vol


def predict_volume(coef, sd1, sd2, cfg, volumes, half_range, x_inits, zs, noises):
    """engine/test_volume.py:262-291 restated per slice (B=1, like the reference) with the Gaussian draws injected:
    `volumes` are the three RAW condition volumes [X,Y,Z] in the order the target modality prescribes;
    x_inits[i] is slice i's starting noise [1,1,S,S], zs[i] / noises[i] its per-step draws.  Returns the [0,1] slices
    at the model's image size (the reference writes them straight back into the volume, so X,Y must equal S there)."""
    S = cfg.image_size
    per_mod = [extract_center_slices(robust_minmax_to_minus1_1(v), half_range) for v in volumes]
    n = len(per_mod[0][0])
    out = []
    for i in range(n):
        conds = []
        for sl, _, _ in per_mod:
            t = torch.from_numpy(np.asarray(sl[i], dtype=np.float32))[None, None]
            if tuple(t.shape[-2:]) != (S, S):
                t = resize_bilinear(t, (S, S))
            conds.append(t)
        fake = sample_from_model(coef, sd1, sd2, cfg, conds[0], conds[1], conds[2], x_inits[i], zs[i], noises[i])
        out.append(((fake + 1.0) / 2.0).clamp(0.0, 1.0).numpy().squeeze())
    return out, per_mod[0][1], per_mod[0][2]
