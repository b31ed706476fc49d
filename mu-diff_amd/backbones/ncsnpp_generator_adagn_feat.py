"""The two mutually-learned NCSN++ denoising generators of MU-Diff, MI355X-native.

`NCSNpp(config)` (G1, contrast-specific) and `NCSNpp_adaptive(config)` (G2, contrast-aware; consumes
G1's prediction as pseudo-target) keep the reference's constructor/forward signatures and state_dict
(reference backbones/ncsnpp_generator_adagn_feat.py:53-447 and :451-905; key layout in SURVEY.md
section 8b), so checkpoints and `engine/test.py` / `demo.ipynb` work unchanged.  `forward` never runs
torch arithmetic: it schedules hand-written gfx950 kernels (libmudiff_hip.so) on NHWC views -

  * embeddings: pixel-norm + z-MLP, sinusoidal t-embedding + MLP, then ALL per-block Dense_0 and AdaGN
    `style` Linears of the network as two batched weight-streaming GEMVs (they depend only on z and t);
  * trunk: fused ResBlocks / attention (backbones/layerspp.py), U-Net concatenations as channel-slice
    views of shared buffers (the producer writes straight into its slot; nothing is copied);
  * tail: GroupNorm+SiLU prologue, 3x3 conv to one channel, tanh epilogue in one direct kernel.

Supported: every configuration the reference itself can construct and run (probed and
recorded by tests/golden/make_golden.py::golden_variants) - embedding_type positional | fourier, conditional on/off,
progressive none | output_skip, progressive_input residual | input_skip (sum | cat) | none, fir on/off, skip_rescale
on/off, centered / not_use_tanh, any nf / ch_mult / num_res_blocks / attn_resolutions / image size, num_channels > 1
for G1, and the two-condition twins (ncsnpp_generator_adagn_feat_healthy.py).  resblock_type 'ddpm' / 'biggan_oneadagn',
progressive='residual' and fir=False with progressive='output_skip' raise inside the reference's own constructor /
forward; they raise NotImplementedError here with that explanation.  GPU tensors only; ambient autocast is ignored
(fp32 in, fp32 out).
"""
import functools

import numpy as np
import torch
import torch.nn as nn

from mudiff_hip import ops
from mudiff_hip.ops import ACT_NONE, ACT_SIGMOID, ACT_SILU, ACT_TANH, INV_SQRT2, PRO_AFFINE_SILU, View

from . import dense_layer, layers, layerspp, utils

ResnetBlockBigGAN = layerspp.ResnetBlockBigGANpp_Adagn
ResnetBlock_Feat = layerspp.ConvFeatBlock
ResnetBlock_Adapt_Feat = layerspp.ConvBlock
ResnetBlock_Feat_GAP = layerspp.ConvBlock_GAP
conv3x3 = layerspp.conv3x3
conv1x1 = layerspp.conv1x1
default_initializer = layers.default_init
dense = dense_layer.dense


class PixelNorm(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, input):
        return ops.pixel_norm(input)


def _check_config(config):
    problems = []
    if config.resblock_type.lower() != 'biggan':
        if config.resblock_type.lower() not in ('ddpm', 'biggan_oneadagn'):
            raise ValueError(f'resblock type {config.resblock_type.lower()} unrecognized.')
        problems.append(f"resblock_type={config.resblock_type!r} (UnboundLocalError on ConvBlock in the reference constructor, :177-180)")
    prog = config.progressive.lower()
    if prog not in ('none', 'output_skip', 'residual'):
        raise ValueError(f'{prog} is not a valid name.')
    if prog == 'residual':
        problems.append("progressive='residual' (the reference's Conv2d(up=True) raises in forward)")
    if prog == 'output_skip' and not config.fir:
        problems.append("progressive='output_skip' with fir=False (the reference's Upsample(fir=False) raises, layerspp.py:164)")
    if config.progressive_input.lower() not in ('none', 'input_skip', 'residual'):
        raise ValueError(f'progressive_input {config.progressive_input!r} unknown.')
    if config.embedding_type.lower() not in ('positional', 'fourier'):
        raise ValueError(f'embedding type {config.embedding_type.lower()} unknown.')
    if problems:
        raise NotImplementedError('this configuration cannot be built or run by the reference either: ' + '; '.join(problems))


class _NCSNppBase(nn.Module, layerspp._Prepared):
    ADAPTIVE = False
    N_COND = 3          # 2 in ncsnpp_generator_adagn_feat_healthy.py
    _loop_cache = None

    def __init__(self, config):
        super().__init__()
        _check_config(config)
        self.config = config
        self.not_use_tanh = config.not_use_tanh
        self.act = act = nn.SiLU()
        self.z_emb_dim = z_emb_dim = config.z_emb_dim
        self.nf = nf = config.num_channels_dae
        ch_mult = list(config.ch_mult)
        self.num_res_blocks = num_res_blocks = config.num_res_blocks
        self.attn_resolutions = attn_resolutions = tuple(int(a) for a in config.attn_resolutions)
        dropout = config.dropout
        self.num_resolutions = num_resolutions = len(ch_mult)
        self.all_resolutions = all_resolutions = [config.image_size // (2 ** i) for i in range(num_resolutions)]
        self.conditional = conditional = config.conditional
        fir, fir_kernel = config.fir, config.fir_kernel
        self.skip_rescale = skip_rescale = config.skip_rescale
        self.resblock_type = config.resblock_type.lower()
        self.progressive = progressive = config.progressive.lower()
        self.progressive_input = progressive_input = config.progressive_input.lower()
        self.embedding_type = embedding_type = config.embedding_type.lower()
        combine_method = config.progressive_combine.lower()
        init_scale = 0.
        channels = config.num_channels

        ResnetBlock = functools.partial(ResnetBlockBigGAN, act=act, dropout=dropout, fir=fir, fir_kernel=fir_kernel,
                                        init_scale=init_scale, skip_rescale=skip_rescale, temb_dim=nf * 4, zemb_dim=z_emb_dim)
        AttnBlock = functools.partial(layerspp.AttnBlockpp, init_scale=init_scale, skip_rescale=skip_rescale)
        if progressive == 'output_skip':
            self.pyramid_upsample = layerspp.Upsample(fir=fir, fir_kernel=fir_kernel, with_conv=False)
        if progressive_input == 'input_skip':
            self.pyramid_downsample = layerspp.Downsample(fir=fir, fir_kernel=fir_kernel, with_conv=False)
        pyramid_downsample = functools.partial(layerspp.Downsample, fir=fir, fir_kernel=fir_kernel, with_conv=True)

        modules = []
        self._plan = plan = []            # (kind, module index, ...) walked by forward

        def add(kind, mod, **kw):
            modules.append(mod)
            plan.append(dict(kind=kind, idx=len(modules) - 1, **kw))

        embed_dim = nf
        if embedding_type == 'fourier':
            add('fourier', layerspp.GaussianFourierProjection(embedding_size=nf, scale=config.fourier_scale))
            embed_dim = 2 * nf
        if conditional:
            for lin in (nn.Linear(embed_dim, nf * 4), nn.Linear(nf * 4, nf * 4)):
                lin.weight.data = default_initializer()(lin.weight.shape)
                nn.init.zeros_(lin.bias)
                add('temb', lin)

        if not self.ADAPTIVE:
            for _ in range(1 + self.N_COND):
                add('feat', ResnetBlock_Feat(act=act, in_ch=channels, out_ch=nf))
            head_c = nf * (1 + self.N_COND)
        else:
            add('gap', ResnetBlock_Feat_GAP(act=act, in_ch=channels, out_ch=nf))
            add('feat', ResnetBlock_Feat(act=act, in_ch=channels, out_ch=nf))
            for _ in range(self.N_COND):
                add('ada', ResnetBlock_Adapt_Feat(act=act, in_ch=channels, out_ch=nf))
            head_c = nf * (4 if self.N_COND == 3 else 2)      # x_feat + one fused map per pair (reference :790 / healthy :311)

        hs_c = [head_c]
        in_ch = head_c
        input_pyramid_ch = channels
        for i_level in range(num_resolutions):
            for _ in range(num_res_blocks):
                out_ch = nf * ch_mult[i_level]
                add('res', ResnetBlock(in_ch=in_ch, out_ch=out_ch), stage='down')
                in_ch = out_ch
                if all_resolutions[i_level] in attn_resolutions:
                    add('attn', AttnBlock(channels=in_ch), stage='down')
                hs_c.append(in_ch)
            if i_level != num_resolutions - 1:
                add('res', ResnetBlock(down=True, in_ch=in_ch), stage='downsample')
                if progressive_input == 'input_skip':
                    add('combine', layerspp.Combine(dim1=input_pyramid_ch, dim2=in_ch, method=combine_method))
                    if combine_method == 'cat':
                        in_ch *= 2
                elif progressive_input == 'residual':
                    add('pyr', pyramid_downsample(in_ch=input_pyramid_ch, out_ch=in_ch))
                    input_pyramid_ch = in_ch
                hs_c.append(in_ch)
        self._hs_channels = list(hs_c)
        in_ch = hs_c[-1]
        add('res', ResnetBlock(in_ch=in_ch), stage='mid')
        add('attn', AttnBlock(channels=in_ch), stage='mid')
        add('res', ResnetBlock(in_ch=in_ch), stage='mid')

        if self.ADAPTIVE:   # registered BEFORE all_modules, like the reference (state_dict order)
            nc = self.N_COND
            pairs = ('c12', 'c23', 'c31') if nc == 3 else ('c12',)
            for j in range(len(pairs)):
                setattr(self, f'feat_weight_c{j + 1}', conv3x3(nf, nf))
            for pair in pairs:
                setattr(self, f'feat_att1_{pair}', conv3x3(nc * nf, nf))
                setattr(self, f'feat_att2_{pair}', conv3x3(nc * nf, nf))
            self._pairs = pairs

        for i_level in reversed(range(num_resolutions)):
            for _ in range(num_res_blocks + 1):
                out_ch = nf * ch_mult[i_level]
                add('res', ResnetBlock(in_ch=in_ch + hs_c.pop(), out_ch=out_ch), stage='up')
                in_ch = out_ch
            if all_resolutions[i_level] in attn_resolutions:
                add('attn', AttnBlock(channels=in_ch), stage='up')
            if progressive == 'output_skip':
                first = i_level == num_resolutions - 1
                add('gn', nn.GroupNorm(num_groups=min(in_ch // 4, 32), num_channels=in_ch, eps=1e-6), stage='pyramid')
                add('conv', conv3x3(in_ch, channels, init_scale=init_scale) if first else conv3x3(in_ch, channels, bias=True, init_scale=init_scale),
                    stage='pyramid', first=first, last=i_level == 0)
            if i_level != 0:
                add('res', ResnetBlock(in_ch=in_ch, up=True), stage='upsample')
        assert not hs_c
        if progressive != 'output_skip':
            add('gn', nn.GroupNorm(num_groups=min(in_ch // 4, 32), num_channels=in_ch, eps=1e-6), stage='tail')
            add('conv', conv3x3(in_ch, channels, init_scale=init_scale), stage='tail')

        self.all_modules = nn.ModuleList(modules)

        mapping_layers = [PixelNorm(), dense(config.nz, z_emb_dim), self.act]
        for _ in range(config.n_mlp):
            mapping_layers.append(dense(z_emb_dim, z_emb_dim))
            mapping_layers.append(self.act)
        self.z_transform = nn.Sequential(*mapping_layers)

    # ------------------------------------------------------------------------------------------
    def begin_loop_cache(self):
        """Opt-in for a sampling loop (mudiff_hip.sampling): between begin_loop_cache() and end_loop_cache() the CONDITION
        images are the caller's loop invariants, so everything that depends on them alone - G1's three condition feature
        blocks (reference :318-328 recomputes them at every reverse step) - is computed by the first forward and kept in
        the U-Net concatenation buffer for the following ones.  Outputs are unchanged.  Plain forward calls never cache."""
        self._loop_cache = {}

    def end_loop_cache(self):
        self._loop_cache = None

    def _prepare(self):
        """Batched small-dense weights: every GroupNorm_{0,1}.style of the ResBlocks in one matrix
        (input zemb), every Dense_0 in another (input silu(temb))."""
        mods = self.all_modules
        style_w, style_b, dense_w, dense_b = [], [], [], []
        offs = {}
        so = do = 0
        for e in self._plan:
            if e['kind'] != 'res':
                continue
            m = mods[e['idx']]
            o0, o1 = so, so + 2 * m.in_ch
            so = o1 + 2 * m.out_ch
            offs[e['idx']] = (o0, o1, so, do)
            do += m.out_ch
            style_w += [m.GroupNorm_0.style.weight, m.GroupNorm_1.style.weight]
            style_b += [m.GroupNorm_0.style.bias, m.GroupNorm_1.style.bias]
            dense_w.append(m.Dense_0.weight)
            dense_b.append(m.Dense_0.bias)
        p = dict(offs=offs,
                 style_w=torch.cat(style_w, 0).contiguous(), style_b=torch.cat(style_b, 0).contiguous(),
                 dense_w=torch.cat(dense_w, 0).contiguous(), dense_b=torch.cat(dense_b, 0).contiguous())
        p['convs'] = {e['idx']: layerspp.ConvParam(mods[e['idx']]) for e in self._plan if e['kind'] == 'conv'}
        if self.ADAPTIVE:
            # all sigmoid gate convs share their input (reference :769-776): ONE conv of 2*n_pairs*nf output channels - first the
            # att1 gates (multiplied by the feature they gate in the epilogue: emul on the first n_pairs*nf channels), then
            # the att2 gates - so the concatenated condition features are staged once
            gates = [getattr(self, 'feat_att1_' + pair) for pair in self._pairs] + [getattr(self, 'feat_att2_' + pair) for pair in self._pairs]
            wg = torch.cat([g.weight for g in gates], 0).contiguous()              # [2*n_pairs*nf, n_cond*nf, 3, 3]
            p['gates'] = layerspp.ConvParam(weight=wg, bias=torch.cat([g.bias for g in gates], 0).contiguous())
            p['fw'] = [layerspp.ConvParam(getattr(self, f'feat_weight_c{j + 1}')) for j in range(len(self._pairs))]
            ada = [mods[e['idx']] for e in self._plan if e['kind'] == 'ada']
            p['ada_w'] = torch.cat([m.group_norm.style.weight for m in ada], 0).contiguous()
            p['ada_b'] = torch.cat([m.group_norm.style.bias for m in ada], 0).contiguous()
        return p

    def _embeddings(self, time_cond, z):
        mods = self.all_modules
        # PixelNorm + the whole z-mapping MLP (SiLU after every layer, reference :271-277) and the timestep MLP (Linear, SiLU, Linear,
        # :301-305) do not depend on each other: both chains side by side in one launch
        zchain = dict(x=z, layers=[(m.weight, m.bias) for m in self.z_transform if isinstance(m, nn.Linear)], pixel_norm=True,
                      act=ACT_SILU, act_last=True)
        if self.embedding_type == 'fourier':      # Gaussian Fourier features of log(sigma) (reference :286-290)
            temb = mods[self._plan[0]['idx']].run_log(time_cond)
        else:
            temb = layers.get_timestep_embedding(time_cond, self.nf)
        if not self.conditional:
            return None, ops.mlp_chains([zchain])[0]
        l0, l1 = (mods[e['idx']] for e in self._plan if e['kind'] == 'temb')
        zemb, temb = ops.mlp_chains([zchain, dict(x=temb, layers=[(l0.weight, l0.bias), (l1.weight, l1.bias)], act=ACT_SILU, act_last=False)])
        return temb, zemb

    def _check_inputs(self, x, conds, pseudo=None):
        ops.require_gpu(x, *conds)
        if len(conds) != self.N_COND or any(c is None for c in conds):
            raise TypeError(f'{type(self).__name__} takes {self.N_COND} condition images')
        B, C, H, W = x.shape
        if C != self.config.num_channels:
            raise ValueError(f'x has {C} channels, the model was built for num_channels={self.config.num_channels}')
        for c in conds:
            if tuple(c.shape) != (B, C, H, W):
                raise ValueError(f'condition shape {tuple(c.shape)} does not match x {tuple(x.shape)}')
        if pseudo is not None:
            ops.require_gpu(pseudo)
            if tuple(pseudo.shape) != (B, C, H, W):     # the reference's conv raises on a channel mismatch too
                raise ValueError(f'pseudo_target shape {tuple(pseudo.shape)} does not match x {tuple(x.shape)}')
        if H % (2 ** (self.num_resolutions - 1)) or W % (2 ** (self.num_resolutions - 1)):
            raise ValueError(f'H, W must be divisible by {2 ** (self.num_resolutions - 1)}')
        return B, H, W

    def _prep_image(self, t):
        v = View.from_nchw(t.detach())
        if not self.config.centered:   # data in [0,1] (reference :309-311)
            v = View(ops.affine_clamp(v.base, 2.0, -1.0, -float('inf'), float('inf')), v.B, v.H, v.W, v.C)
        return v

    def _make_buffers(self, B, H, W, dev, arena=None):
        """U-Net concatenation buffers.  Skip k (the k-th tensor pushed on the down path) is popped by
        up-block j = n_skips-1-k, whose input is cat([h, skip_k]) (reference :383).  Each up-block gets
        ONE buffer [B,H,W,Ch+Cs]; the producer of h and the producer of skip_k write straight into
        their channel slots, so no concatenation is ever copied."""
        mods = self.all_modules
        trunk = [e for e in self._plan if e['kind'] in ('res', 'attn', 'pyr', 'combine') or e.get('stage') in ('pyramid', 'tail')]
        sizes = [(H, W)]
        hh, ww = H, W
        for e in trunk:
            if e['kind'] == 'res' and e['stage'] == 'down':
                sizes.append((hh, ww))
            elif e['kind'] == 'res' and e['stage'] == 'downsample':
                hh, ww = hh // 2, ww // 2
                sizes.append((hh, ww))
        n_skips = len(self._hs_channels)
        assert len(sizes) == n_skips
        up_blocks = [e for e in trunk if e['kind'] == 'res' and e['stage'] == 'up']
        assert len(up_blocks) == n_skips
        bufs = []
        for j, e in enumerate(up_blocks):
            m = mods[e['idx']]
            cs = self._hs_channels[n_skips - 1 - j]
            sh_, sw_ = sizes[n_skips - 1 - j]
            bufs.append((View.empty(B, sh_, sw_, m.in_ch, dev, arena), m.in_ch - cs, cs))
        return trunk, bufs

    def _trunk(self, p, trunk, bufs, x_img: View, temb, zemb, arena=None):
        """Down / mid / up path (reference :335-447, identical for both generators).  The head feature
        map (skip 0) has already been written into bufs[-1]'s skip slot by the caller."""
        mods = self.all_modules
        n_skips = len(bufs)
        styles = ops.dense(zemb, p['style_w'], p['style_b'])                       # [B, sum 2C]
        tb_all = ops.dense(temb, p['dense_w'], p['dense_b'], act_in=ACT_SILU) if temb is not None else None      # [B, sum Cout]
        rescale = INV_SQRT2 if self.skip_rescale else 1.0

        def res(e, x, out=None):
            o0, o1, o2, d0 = p['offs'][e['idx']]
            m = mods[e['idx']]
            if out is None:       # block output consumed by another GroupNorm (mid blocks, attention, upsample blocks, tail)
                out = View.empty(x.B, x.H * (2 if m.up else 1) // (2 if m.down else 1), x.W * (2 if m.up else 1) // (2 if m.down else 1),
                                 m.out_ch, x.device, arena)
            tb = tb_all[:, d0:d0 + m.out_ch] if tb_all is not None else None
            return m.run(x, styles[:, o0:o1], styles[:, o1:o2], tb, out=out, arena=arena)

        def skip_slot(k):
            buf, ch, cs = bufs[n_skips - 1 - k]
            return buf.slice(ch, cs)

        def h_slot(j):
            buf, ch, cs = bufs[j]
            return buf.slice(0, ch)

        def nxt_is(i, kind, stage):
            return i + 1 < len(trunk) and trunk[i + 1]['kind'] == kind and trunk[i + 1].get('stage') == stage

        skips = [skip_slot(0)]
        pyr = x_img if self.progressive_input != 'none' else None
        out_pyr = None          # progressive='output_skip' image pyramid [B,h,w,channels]
        h = skips[0]
        up_j = 0
        i = 0
        while i < len(trunk):
            e = trunk[i]
            kind, stage = e['kind'], e.get('stage')
            if kind == 'res' and stage == 'down':
                if nxt_is(i, 'attn', 'down'):
                    h = res(e, skips[-1])
                    i += 1
                    h = mods[trunk[i]['idx']].run(h, out=skip_slot(len(skips)))
                else:
                    h = res(e, skips[-1], out=skip_slot(len(skips)))
                skips.append(h)
            elif kind == 'res' and stage == 'downsample':
                slot = skip_slot(len(skips))
                if nxt_is(i, 'pyr', None):
                    hd = res(e, skips[-1])
                    i += 1
                    # input pyramid: FIR + strided conv, epilogue fuses (+bias, + hd) / sqrt2 (reference :359-366)
                    h = mods[trunk[i]['idx']].run(pyr, res=hd, out_scale=rescale, out=slot)
                    pyr = h
                elif nxt_is(i, 'combine', None):
                    comb = mods[trunk[i + 1]['idx']]
                    pyr = self.pyramid_downsample.run(pyr)                     # parameter-free /2 of the image (reference :349)
                    if comb.method == 'cat':                                    # cat([conv1x1(pyr), hd]): hd written in place
                        d = slot.C // 2
                        hd = res(e, skips[-1], out=slot.slice(d, d))
                        h = comb.run(pyr, hd, out=slot)
                    else:
                        hd = res(e, skips[-1])
                        h = comb.run(pyr, hd, out=slot)
                    i += 1
                else:                                                           # progressive_input='none'
                    h = res(e, skips[-1], out=slot)
                skips.append(h)
            elif kind == 'res' and stage == 'mid':
                h = res(e, h, out=None if nxt_is(i, 'attn', 'mid') else h_slot(0))
            elif kind == 'attn':
                h = mods[e['idx']].run(h, out=View.empty(h.B, h.H, h.W, h.C, h.device, arena))
            elif kind == 'res' and stage == 'up':
                buf, ch, cs = bufs[up_j]
                assert h.base is buf.base and skips[-1].base is buf.base, 'concat slot bookkeeping broke'
                skips.pop()
                up_j += 1
                h = res(e, buf, out=h_slot(up_j) if nxt_is(i, 'res', 'up') else None)
            elif kind == 'res' and stage == 'upsample':
                h = res(e, h, out=h_slot(up_j))
            elif kind == 'gn' and stage == 'pyramid':
                # output pyramid (reference :389-411): pyramid = up(pyramid) + conv3x3(silu(GN(h))); the sum (and the final
                # tanh) run in the conv epilogue
                gn = mods[e['idx']]
                sc, sh = ops.gn_scale_shift(h, gn.num_groups, gn.weight.detach(), gn.bias.detach())
                i += 1
                ce = trunk[i]
                up = None if ce['first'] else self.pyramid_upsample.run(out_pyr)
                last_act = ACT_TANH if (ce['last'] and not self.not_use_tanh) else ACT_NONE
                out_pyr = p['convs'][ce['idx']](h, pro=(sc, sh, PRO_AFFINE_SILU), res=up, act=last_act)
            elif kind == 'gn' and stage == 'tail':
                # ---- tail: GroupNorm(affine) + SiLU prologue, conv3x3 -> image channels, tanh epilogue
                assert not skips and up_j == n_skips
                gn = mods[e['idx']]
                sc, sh = ops.gn_scale_shift(h, gn.num_groups, gn.weight.detach(), gn.bias.detach())
                i += 1
                out_pyr = p['convs'][trunk[i]['idx']](h, pro=(sc, sh, PRO_AFFINE_SILU), act=ACT_NONE if self.not_use_tanh else ACT_TANH)
            else:
                raise AssertionError(e)
            i += 1
        assert not skips and up_j == n_skips
        return out_pyr.to_nchw()


class _G1(_NCSNppBase):
    ADAPTIVE = False

    def _forward(self, x, conds, time_cond, z):
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            B, H, W = self._check_inputs(x, conds)
            p = self.prepared()
            mods = self.all_modules
            temb, zemb = self._embeddings(time_cond, z)
            xv = self._prep_image(x)
            imgs = [xv] + [View.from_nchw(c.detach()) for c in conds]
            nf = self.nf
            arena = ops.StatsArena(xv.device)
            trunk, bufs = self._make_buffers(B, H, W, xv.device, arena)
            feats = [e for e in self._plan if e['kind'] == 'feat']
            lc = self._loop_cache
            key = (B, H, W) + tuple((c.data_ptr(), c._version) for c in conds)
            reuse = lc is not None and lc.get('key') == key
            if reuse:
                # the last concatenation buffer [h | x_feat | c1f | c2f | c3f] is the cached one: the condition slots (and
                # their GroupNorm sums) are already there; x_feat and h are rewritten below / by the up path
                buf = lc['buf']
                buf.stats = arena.take(B, buf.C)
                c0 = bufs[-1][1] + nf
                if buf.stats is not None:
                    buf.stats[:, c0:].copy_(lc['stats'])
                bufs[-1] = (buf, bufs[-1][1], bufs[-1][2])
            hs0 = bufs[-1][0].slice(bufs[-1][1], bufs[-1][2])
            for j, (e, img) in enumerate(zip(feats, imgs)):
                if reuse and j > 0:
                    continue
                mods[e['idx']].run(img, out=hs0.slice(j * nf, nf), arena=arena)
            if lc is not None and not reuse:
                st_ = bufs[-1][0].stats
                lc.update(key=key, buf=bufs[-1][0], stats=None if st_ is None else st_[:, bufs[-1][1] + nf:].clone())
            return self._trunk(p, trunk, bufs, xv, temb, zemb, arena)


class _G2(_NCSNppBase):
    ADAPTIVE = True

    def _forward(self, x, conds, time_cond, z, pseudo_target):
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            B, H, W = self._check_inputs(x, conds, pseudo_target)
            p = self.prepared()
            mods = self.all_modules
            temb, zemb = self._embeddings(time_cond, z)
            xv = self._prep_image(x)
            nf, dev, nc = self.nf, xv.device, self.N_COND
            e_gap = next(e for e in self._plan if e['kind'] == 'gap')
            e_feat = next(e for e in self._plan if e['kind'] == 'feat')
            e_ada = [e for e in self._plan if e['kind'] == 'ada']
            arena = ops.StatsArena(dev)
            pseudo_weight = mods[e_gap['idx']].run(View.from_nchw(pseudo_target.detach().contiguous()), arena=arena)   # [B,256]
            trunk, bufs = self._make_buffers(B, H, W, dev, arena)
            hs0 = bufs[-1][0].slice(bufs[-1][1], bufs[-1][2])
            mods[e_feat['idx']].run(xv, out=hs0.slice(0, nf), arena=arena)
            ada_styles = ops.dense(pseudo_weight, p['ada_w'], p['ada_b'])                                  # [B, nc*2nf]
            cat = View.empty(B, H, W, nc * nf, dev)
            lc = self._loop_cache
            if lc is not None:      # inside a sampling loop: conv1 of the condition blocks is a loop invariant (begin_loop_cache)
                key = (B, H, W) + tuple((c.data_ptr(), c._version) for c in conds)
                if lc.get('key') != key:
                    lc.clear()
                    lc.update(key=key, ada=[{} for _ in e_ada])
            for j, (e, c) in enumerate(zip(e_ada, conds)):
                mods[e['idx']].run(View.from_nchw(c.detach()), ada_styles[:, j * 2 * nf:(j + 1) * 2 * nf], out=cat.slice(j * nf, nf), arena=arena,
                                   cache=lc['ada'][j] if lc is not None else None)
            # att1 gates already multiplied by the feature they gate: sigmoid(conv(cat)) * c_i (pair k gates c_k: c12 -> c1,
            # c23 -> c2, c31 -> c3, reference :778,783,787); att2 gates: plain sigmoid
            npair = len(self._pairs)
            gall = p['gates'](cat, act=ACT_SIGMOID, emul=cat.slice(0, npair * nf), emul_cout=npair * nf)
            gated, g2all = gall.slice(0, npair * nf), gall.slice(npair * nf, npair * nf)
            for j in range(npair):      # fused_ij = g2 * conv(g1 * c_i) + (1 - g2) * c_j   (reference :779-788)
                other = (j + 1) % nc
                p['fw'][j](gated.slice(j * nf, nf), gate=(g2all.slice(j * nf, nf), cat.slice(other * nf, nf)),
                           out=hs0.slice((j + 1) * nf, nf))
            return self._trunk(p, trunk, bufs, xv, temb, zemb, arena)


@utils.register_model(name='ncsnpp')
class NCSNpp(_G1):
    """G1 - contrast-specific NCSN++ generator (reference :53-447)."""

    def forward(self, x, cond1, cond2, cond3, time_cond, z):
        return self._forward(x, (cond1, cond2, cond3), time_cond, z)


@utils.register_model(name='ncsnpp_adaptive')
class NCSNpp_adaptive(_G2):
    """G2 - contrast-aware NCSN++ generator (reference :451-905): the three condition feature maps are
    AdaGN-modulated by a style vector pooled from the pseudo-target (G1's prediction) and fused pairwise
    through sigmoid gates before entering the shared trunk."""

    def forward(self, x, cond1, cond2, cond3, time_cond, z, pseudo_target):
        return self._forward(x, (cond1, cond2, cond3), time_cond, z, pseudo_target)
