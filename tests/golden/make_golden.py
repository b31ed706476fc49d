"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself (CPU) and,
on the way, check the oracle restatement (oracle/mudiff_oracle.py) against it.

Runs only in the build container (needs /root/reference).  Nothing from the reference is
copied: the fixtures hold inputs and the reference's outputs only.  Safe-import recipe
(SURVEY.md section 8c): no bytecode, the reference's import-time JIT build of its CUDA ops is
made to fail (it then takes its own CPU path, utils/op/upfirdn2d.py:32-35,171-174), scratch cwd,
`engine.train` is never imported; `engine/test.py` lines 47-199 (pure torch/numpy diffusion
math) are exec'd from text because the module itself needs torchvision.

    python tests/golden/make_golden.py            # validate + (re)write fixtures
    python tests/golden/make_golden.py --check    # validate only
"""
import argparse
import contextlib
import io
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('MUDIFF_REFERENCE', '/root/reference')

import numpy as np
import torch
import torch.utils.cpp_extension as _cpp


def _no_jit(*a, **k):
    raise RuntimeError('JIT build of the reference CUDA ops disabled (oracle run)')


_cpp.load = _no_jit
os.chdir(tempfile.mkdtemp(prefix='mudiff_ref_'))
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    from backbones import layerspp as R_layerspp                     # noqa: E402
    from backbones import layers as R_layers                         # noqa: E402
    from backbones import up_or_down_sampling as R_ud                # noqa: E402
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp as R_G1, NCSNpp_adaptive as R_G2  # noqa: E402
from oracle import mudiff_oracle as O                                # noqa: E402

torch.set_grad_enabled(False)


def ref_engine_namespace():
    """exec engine/test.py:47-199 (diffusion coefficients, posterior, sample_from_model) and
    engine/train.py:246-281 (Diffusion_Coefficients, q_sample, q_sample_pairs)."""
    ns = {'torch': torch, 'np': np, 'autocast': contextlib.nullcontext}
    lines = open(os.path.join(REF, 'engine/test.py')).read().split('\n')
    exec('\n'.join(lines[46:199]), ns)
    lines = open(os.path.join(REF, 'engine/train.py')).read().split('\n')
    exec('\n'.join(lines[245:281]), ns)
    return ns


E = ref_engine_namespace()
REPORT = []


def check(name, a, b, tol):
    assert torch.equal(torch.isnan(a), torch.isnan(b)), name
    d = float(torch.nan_to_num(a - b).abs().max()) if a.numel() else 0.0
    REPORT.append((name, d, tol))
    status = 'ok ' if d <= tol else 'FAIL'
    print(f'  [{status}] {name:48s} max|oracle-ref| = {d:.3e} (tol {tol:g})')
    assert d <= tol, name


def load_sd(module, sd):
    missing, unexpected = module.load_state_dict(sd, strict=True), None
    return module


def t2n(d):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


# ----------------------------------------------------------------------------------------
def golden_schedules():
    print('schedules / posterior tables')
    out = {}
    for tag, kw in (('T4', dict(num_timesteps=4)), ('T8', dict(num_timesteps=8)),
                    ('T4geo', dict(num_timesteps=4, use_geometric=True, beta_min=0.01, beta_max=0.9))):
        cfg = O.default_config(**kw)
        rp = E['Posterior_Coefficients'](cfg, 'cpu')
        rd = E['Diffusion_Coefficients'](cfg, 'cpu')
        rT = E['get_time_schedule'](cfg, 'cpu')
        op, od, oT = O.PosteriorCoefficients(cfg), O.DiffusionCoefficients(cfg), O.get_time_schedule(cfg)
        for f in ('betas', 'alphas_cumprod', 'posterior_variance', 'posterior_mean_coef1', 'posterior_mean_coef2',
                  'posterior_log_variance_clipped'):
            check(f'{tag}.{f}', getattr(op, f), getattr(rp, f), 0.0)
            out[f'{tag}.{f}'] = getattr(rp, f)
        for f in ('sigmas', 'a_s', 'a_s_cum', 'sigmas_cum', 'a_s_prev'):
            check(f'{tag}.{f}', getattr(od, f), getattr(rd, f), 0.0)
            out[f'{tag}.{f}'] = getattr(rd, f)
        check(f'{tag}.T', oT, rT, 0.0)
        out[f'{tag}.T'] = rT
    return out


def golden_elementwise():
    print('posterior / q_sample')
    cfg = O.default_config()
    rp, rd = E['Posterior_Coefficients'](cfg, 'cpu'), E['Diffusion_Coefficients'](cfg, 'cpu')
    op, od = O.PosteriorCoefficients(cfg), O.DiffusionCoefficients(cfg)
    g = torch.Generator().manual_seed(11)
    B, H, W = 5, 12, 20
    x01, x02, xt = (torch.randn(B, 1, H, W, generator=g) for _ in range(3))
    t = torch.tensor([0, 1, 2, 3, 0])
    out = dict(x01=x01, x02=x02, xt=xt, t=t)
    torch.manual_seed(3); r = E['sample_posterior_combine'](rp, x01, x02, xt, t)
    torch.manual_seed(3); nz = torch.randn_like(xt)
    check('sample_posterior_combine', O.sample_posterior_combine(op, x01, x02, xt, t, nz), r, 0.0)
    out.update(noise=nz, posterior_combine=r)
    torch.manual_seed(3); r = E['sample_posterior'](rp, x01, xt, t)
    check('sample_posterior', O.sample_posterior(op, x01, xt, t, nz), r, 0.0)
    out.update(posterior=r)
    torch.manual_seed(3); r = E['q_sample'](rd, x01, t)
    check('q_sample', O.q_sample(od, x01, t, nz), r, 0.0)
    out.update(q_sample=r)
    torch.manual_seed(4); r0, r1 = E['q_sample_pairs'](rd, x01, t)
    torch.manual_seed(4); n_outer = torch.randn_like(x01); n_inner = torch.randn_like(x01)
    o0, o1 = O.q_sample_pairs(od, x01, t, n_inner, n_outer)
    check('q_sample_pairs[0]', o0, r0, 0.0); check('q_sample_pairs[1]', o1, r1, 0.0)
    out.update(noise_inner=n_inner, noise_outer=n_outer, q_pair0=r0, q_pair1=r1)
    return out


def golden_fir():
    print('FIR resamplers (upfirdn2d CPU path of the reference)')
    g = torch.Generator().manual_seed(21)
    out = {}
    for tag, shape in (('a', (2, 4, 8, 8)), ('b', (1, 3, 6, 10)), ('c', (1, 2, 16, 16))):
        x = torch.randn(*shape, generator=g)
        up, dn = R_ud.upsample_2d(x, (1, 3, 3, 1), factor=2), R_ud.downsample_2d(x, (1, 3, 3, 1), factor=2)
        check(f'upsample_2d.{tag}', O.upsample_2d(x), up, 1e-6)
        check(f'downsample_2d.{tag}', O.downsample_2d(x), dn, 1e-6)
        w = torch.randn(5, shape[1], 3, 3, generator=g) * 0.2
        cd = R_ud.conv_downsample_2d(x, w, k=(1, 3, 3, 1))
        check(f'conv_downsample_2d.{tag}', O.conv_downsample_2d(x, w), cd, 2e-6)
        out.update({f'{tag}.x': x, f'{tag}.up': up, f'{tag}.down': dn, f'{tag}.w': w, f'{tag}.convdown': cd})
    # generic upfirdn2d call with asymmetric pads and a non-separable kernel
    from utils.op import upfirdn2d as R_upfirdn2d
    x = torch.randn(2, 3, 7, 9, generator=g)
    k = torch.randn(4, 4, generator=g)
    for tag, (u, d, pad) in (('g1', (1, 1, (2, 1))), ('g2', (2, 1, (2, 1))), ('g3', (1, 2, (1, 1))), ('g4', (2, 2, (3, 0)))):
        r = R_upfirdn2d(x, k, up=u, down=d, pad=pad)
        check(f'upfirdn2d.{tag}', O.upfirdn2d(x, k, up=u, down=d, pad=pad), r, 2e-6)
        out[f'{tag}.out'] = r
    out.update({'g.x': x, 'g.k': k})
    return out


def _seeded(module, tag, seed=77):
    """Fill a reference module's parameters from a per-name generator (same rule as
    oracle.make_state_dict so degenerate zero-init tensors become non-trivial)."""
    import zlib, math
    sd = {}
    for name, p in module.state_dict().items():
        g = torch.Generator().manual_seed((zlib.crc32(f'{tag}:{name}'.encode()) + seed) % (2 ** 31))
        if p.dim() >= 2:
            rf = int(np.prod(p.shape[2:])) if p.dim() > 2 else 1
            bound = math.sqrt(3.0 / ((p.shape[0] + p.shape[1]) * rf / 2.0))
            t = (torch.rand(p.shape, generator=g) * 2 - 1) * bound
        else:
            t = 0.1 * torch.randn(p.shape, generator=g)
            if name.endswith('style.bias'):
                t[: p.shape[0] // 2] += 1.0
            elif name.endswith('.weight'):
                t += 1.0
        sd[name] = t
    module.load_state_dict(sd)
    return sd


def golden_blocks():
    print('L1 blocks (reference nn.Modules with seeded weights)')
    import torch.nn as nn
    act = nn.SiLU()
    g = torch.Generator().manual_seed(31)
    out = {}

    def pref(sd, p):
        return {p + '.' + k: v for k, v in sd.items()}

    B, zd, td = 2, 24, 32
    zemb, temb = torch.randn(B, zd, generator=g), torch.randn(B, td, generator=g)
    out.update(zemb=zemb, temb=temb)
    for tag, cin, cout, up, down, hw in (('plain', 8, 8, False, False, 8), ('skip', 8, 16, False, False, 8),
                                         ('up', 12, 12, True, False, 6), ('down', 8, 8, False, True, 8),
                                         ('cat', 24, 16, False, False, 10)):
        m = R_layerspp.ResnetBlockBigGANpp_Adagn(act, cin, cout, temb_dim=td, zemb_dim=zd, up=up, down=down,
                                                dropout=0.0, fir=True, fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.)
        sd = _seeded(m, 'res_' + tag)
        x = torch.randn(B, cin, hw, hw + 2, generator=g)
        r = m(x, temb, zemb)
        check(f'resblock.{tag}', O.resblock(pref(sd, 'm'), 'm', x, temb, zemb, up=up, down=down), r, 5e-6)
        out.update({f'res_{tag}.x': x, f'res_{tag}.y': r}); out.update(pref(t2n_t(sd), f'res_{tag}.sd'))
    m = R_layerspp.AdaptiveGroupNorm(4, 16, zd)
    sd = _seeded(m, 'adagn')
    x = torch.randn(B, 16, 5, 7, generator=g) * 2 + 0.5
    r = m(x, zemb)
    check('adagn', O.adagn(pref(sd, 'm'), 'm', x, zemb), r, 2e-6)
    out.update({'adagn.x': x, 'adagn.y': r}); out.update(pref(t2n_t(sd), 'adagn.sd'))
    for tag, c, hw in (('c16', 16, 8), ('c32', 32, 6)):
        m = R_layerspp.AttnBlockpp(c, skip_rescale=True, init_scale=0.)
        sd = _seeded(m, 'attn_' + tag)
        x = torch.randn(B, c, hw, hw, generator=g)
        r = m(x)
        check(f'attn.{tag}', O.attn_block(pref(sd, 'm'), 'm', x), r, 5e-6)
        out.update({f'attn_{tag}.x': x, f'attn_{tag}.y': r}); out.update(pref(t2n_t(sd), f'attn_{tag}.sd'))
    x1 = torch.randn(B, 1, 12, 12, generator=g)
    m = R_layerspp.ConvFeatBlock(act, in_ch=1, out_ch=16)
    sd = _seeded(m, 'feat'); r = m(x1)
    check('conv_feat_block', O.conv_feat_block(pref(sd, 'm'), 'm', x1), r, 5e-6)
    out.update({'feat.x': x1, 'feat.y': r}); out.update(pref(t2n_t(sd), 'feat.sd'))
    m = R_layerspp.ConvBlock(act, in_ch=1, out_ch=16, zemb_dim=zd)
    sd = _seeded(m, 'ada'); r = m(x1, zemb)
    check('conv_block', O.conv_block(pref(sd, 'm'), 'm', x1, zemb), r, 5e-6)
    out.update({'ada.y': r}); out.update(pref(t2n_t(sd), 'ada.sd'))
    m = R_layerspp.ConvBlock_GAP(act, in_ch=1, out_ch=16, zemb_dim=zd)
    sd = _seeded(m, 'gap')
    with contextlib.redirect_stdout(io.StringIO()):
        r = m(x1)
    check('conv_block_gap', O.conv_block_gap(pref(sd, 'm'), 'm', x1), r, 5e-6)
    out.update({'gap.y': r}); out.update(pref(t2n_t(sd), 'gap.sd'))
    for tag, cin, cout in (('p1', 1, 8), ('p8', 8, 16)):
        m = R_layerspp.Downsample(in_ch=cin, out_ch=cout, with_conv=True, fir=True, fir_kernel=(1, 3, 3, 1))
        sd = _seeded(m, 'pyr_' + tag)
        x = torch.randn(B, cin, 12, 16, generator=g)
        r = m(x)
        check(f'pyramid_downsample.{tag}', O.pyramid_downsample(pref(sd, 'm'), 'm', x), r, 5e-6)
        out.update({f'pyr_{tag}.x': x, f'pyr_{tag}.y': r}); out.update(pref(t2n_t(sd), f'pyr_{tag}.sd'))
    t = torch.tensor([0, 1, 2, 3, 7, 999])
    r = R_layers.get_timestep_embedding(t, 64)
    check('timestep_embedding', O.timestep_embedding(t, 64), r, 0.0)
    out.update({'temb.t': t, 'temb.y': r})
    return out


def t2n_t(sd):
    return dict(sd)


def _ref_models(cfg, seed):
    with contextlib.redirect_stdout(io.StringIO()):
        g1, g2 = R_G1(cfg), R_G2(cfg)
    sd1, sd2 = O.make_state_dict(cfg, 'g1', seed), O.make_state_dict(cfg, 'g2', seed)
    # names, order and shapes must be the reference's own
    for m, sd, w in ((g1, sd1, 'g1'), (g2, sd2, 'g2')):
        rsd = m.state_dict()
        assert list(rsd.keys()) == list(sd.keys()), f'{w}: state_dict key order differs'
        assert all(tuple(rsd[k].shape) == tuple(sd[k].shape) for k in sd), f'{w}: shapes differ'
        m.load_state_dict(sd, strict=True)
    return g1, g2, sd1, sd2


def run_sampler(cfg, seed_w, seed_x, conds, B):
    g1, g2, sd1, sd2 = _ref_models(cfg, seed_w)
    n = cfg.num_timesteps
    H = cfg.image_size
    g = torch.Generator().manual_seed(seed_x)
    x_init = torch.randn(B, 1, H, H, generator=g)
    rp = E['Posterior_Coefficients'](cfg, 'cpu')
    # replay the reference's RNG consumption: per step randn(B,nz) then randn_like(x)  (engine/test.py:188,169)
    torch.manual_seed(seed_x + 1)
    zs, noises = [], []
    for _ in range(n):
        zs.append(torch.randn(B, cfg.nz)); noises.append(torch.randn(B, 1, H, H))
    torch.manual_seed(seed_x + 1)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_final = E['sample_from_model'](rp, g1, conds[0], g2, conds[1], conds[2], n, x_init, None, cfg)
    # per-step reference intermediates, with the same injected draws
    steps = []
    x = x_init
    for k, i in enumerate(reversed(range(n))):
        t = torch.full((B,), i, dtype=torch.int64)
        x01 = g1(x, conds[0], conds[1], conds[2], t, zs[k])
        x02 = g2(x, conds[0], conds[1], conds[2], t, zs[k], x01[:, [0], :])
        torch.manual_seed(0)
        mean_noise = noises[k]
        xn = O.sample_posterior_combine(O.PosteriorCoefficients(cfg), x01, x02, x, t, mean_noise)  # checked bit-exact above
        steps.append((x01, x02, xn)); x = xn
    assert torch.equal(x, ref_final), 'replayed RNG stream does not reproduce the reference sample_from_model'
    o_final, o_steps = O.sample_from_model(O.PosteriorCoefficients(cfg), sd1, sd2, cfg, conds[0], conds[1], conds[2],
                                           x_init, zs, noises, return_steps=True)
    return x_init, zs, noises, steps, o_steps


def golden_small_models():
    print('small full models (both generators, every step)')
    out = {}
    for tag, kw in (('s32', dict(image_size=32, num_channels_dae=32, ch_mult=[1, 2, 4], attn_resolutions=(16,))),
                    ('s32na', dict(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(), num_res_blocks=1)),
                    ('s16t8', dict(image_size=16, num_channels_dae=16, ch_mult=[1, 1, 2], attn_resolutions=(4,),
                                   num_timesteps=8, nz=50, z_emb_dim=64, n_mlp=2))):
        cfg = O.default_config(**kw)
        B = 2
        g = torch.Generator().manual_seed(101)
        conds = [torch.tanh(torch.randn(B, 1, cfg.image_size, cfg.image_size, generator=g)) for _ in range(3)]
        x_init, zs, noises, steps, o_steps = run_sampler(cfg, 1234, 42, conds, B)
        for k, (r, o) in enumerate(zip(steps, o_steps)):
            for nm, a, b in zip(('x01', 'x02', 'xnew'), o, r):
                check(f'{tag}.step{k}.{nm}', a, b, 2e-5)
                out[f'{tag}.step{k}.{nm}'] = b
        out.update({f'{tag}.x_init': x_init, f'{tag}.c1': conds[0], f'{tag}.c2': conds[1], f'{tag}.c3': conds[2]})
        for k in range(cfg.num_timesteps):
            out[f'{tag}.z{k}'] = zs[k]; out[f'{tag}.noise{k}'] = noises[k]
        print(f'    {tag}: |x01| std {float(steps[0][0].std()):.3f}')
    return out


def golden_discriminator():
    print('critic (Discriminator_large, reference nn.Module with seeded weights)')
    from backbones.discriminator import Discriminator_large as R_D
    import torch.nn as nn
    out = {}
    for tag, nc, ngf, td, B, H in (('d8', 2, 8, 32, 4, 64), ('d16', 2, 16, 64, 2, 128), ('d8b8', 2, 8, 32, 8, 64)):
        with contextlib.redirect_stdout(io.StringIO()):
            d = R_D(nc=nc, ngf=ngf, t_emb_dim=td, act=nn.LeakyReLU(0.2))
        sd = O.make_discriminator_state_dict(nc, ngf, td, 1234)
        assert list(d.state_dict().keys()) == list(sd.keys()), 'critic state_dict key order differs'
        d.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(ngf + B)
        x, xt = torch.randn(B, 1, H, H, generator=g), torch.randn(B, 1, H, H, generator=g)
        t = torch.randint(0, 4, (B,), generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            logit, mid = d(x, t, xt)
        ol, om = O.discriminator_large_forward(sd, x, t, xt, td)
        check(f'critic.{tag}.logit', ol, logit, 2e-5)
        check(f'critic.{tag}.mid', om, mid, 2e-5)
        out.update({f'{tag}.x': x, f'{tag}.xt': xt, f'{tag}.t': t, f'{tag}.logit': logit, f'{tag}.mid': mid})
    return out


def demo_inputs_u8():
    """The reference's own demo data (demo/sample_data/*.jpg), decoded to 8-bit grayscale."""
    from PIL import Image
    out = {}
    for n in ('flair', 't2', 't1', 't1ce'):
        out[n] = np.array(Image.open(os.path.join(REF, 'demo/sample_data', n + '.jpg')).convert('L'))
        assert out[n].shape == (256, 256)
    return out


def preprocess_demo(u8):
    """demo.ipynb cell 4: percentile clip over non-zero pixels, min-max, (x-0.5)/0.5, rot90(k=-1)."""
    img = u8.astype(np.float64) if False else u8
    nz = img > 0
    low, high = np.percentile(img[nz], [1, 99])
    img = np.clip(img, low, high)
    img = (img - img.min()) / (img.max() - img.min())
    img = (img - 0.5) / 0.5
    t = torch.tensor(img, dtype=torch.float32)[None, None]
    return torch.rot90(t, k=-1, dims=(2, 3)).contiguous()


def golden_full():
    print('config 1/2: 256x256, nf=64, ch_mult 1-2-4, demo JPEG inputs, 4 steps (takes ~1 min)')
    cfg = O.default_config()
    u8 = demo_inputs_u8()
    conds = [preprocess_demo(u8[n]) for n in ('flair', 't2', 't1')]
    x_init, zs, noises, steps, o_steps = run_sampler(cfg, 1234, 42, conds, 1)
    out = {}
    for k, (r, o) in enumerate(zip(steps, o_steps)):
        for nm, a, b in zip(('x01', 'x02', 'xnew'), o, r):
            check(f'cfg2.step{k}.{nm}', a, b, 5e-5)
            out[f'step{k}.{nm}'] = b.numpy().astype(np.float32)
    print(f'    cfg2: |x01| std {float(steps[0][0].std()):.3f}, final range [{float(steps[-1][2].min()):.3f},{float(steps[-1][2].max()):.3f}]')
    return u8, out


def brats_like_slices_u8(n, seed=2019, size=256):
    """Synthetic BraTS-shaped slices: smooth Gaussian fields inside a head-sized disc, black
    background, 4 contrasts per slice, quantised to uint8 so that the fixture is exact on every host."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing='ij')
    out = np.zeros((n, 4, size, size), np.uint8)
    k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]) / 64.0
    for i in range(n):
        cx, cy, r = (torch.rand(3, generator=g) * torch.tensor([0.2, 0.2, 0.15]) + torch.tensor([-0.1, -0.1, 0.7])).tolist()
        mask = ((xx - cx) ** 2 + ((yy - cy) / 1.15) ** 2) < r * r
        base = torch.randn(1, 1, size // 8, size // 8, generator=g)
        base = torch.nn.functional.interpolate(base, size=(size, size), mode='bilinear', align_corners=False)
        for c in range(4):
            f = 0.7 * base + 0.5 * torch.randn(1, 1, size, size, generator=g)
            for _ in range(3):
                f = torch.nn.functional.conv2d(torch.nn.functional.pad(f, (2, 1, 2, 1)), k[None, None])
            f = (f - f[0, 0][mask].mean()) / f[0, 0][mask].std()
            v = torch.clamp(f[0, 0], -3, 3) / 3
            v = torch.where(mask, v, torch.full_like(v, -1.0))
            out[i, c] = torch.round((v + 1) / 2 * 255).to(torch.uint8).numpy()
    return out


def golden_cfg3():
    print('config 3: BraTS-shaped batch, 256x256, 4 slices x (3 conditions + target), 4 steps (takes a few minutes)')
    cfg = O.default_config()
    u8 = brats_like_slices_u8(4)
    sl = torch.from_numpy(u8.astype(np.float32)) / 255.0 * 2.0 - 1.0
    conds = [sl[:, c:c + 1].contiguous() for c in range(3)]
    x_init, zs, noises, steps, o_steps = run_sampler(cfg, 1234, 314, conds, 4)
    for k, (r, o) in enumerate(zip(steps, o_steps)):
        for nm, a, b in zip(('x01', 'x02', 'xnew'), o, r):
            check(f'cfg3.step{k}.{nm}', a, b, 5e-5)
    out = {'slices_u8': u8, 'final': steps[-1][2].numpy().astype(np.float32)}
    print(f'    cfg3: final range [{float(steps[-1][2].min()):.3f},{float(steps[-1][2].max()):.3f}]')
    return out


CFG3_MODS = ['FLAIR', 'T2', 'T1', 'T1CE']        # channel order of brats_like_slices_u8's 4 synthetic contrasts
CFG3_ORDERS = {      # reference dataset/dataset_brats.py:29-34: condition order per target contrast (target last)
    'T1CE': ['FLAIR', 'T2', 'T1', 'T1CE'],
    'FLAIR': ['T1CE', 'T1', 'T2', 'FLAIR'],
    'T2': ['T1CE', 'T1', 'FLAIR', 'T2'],
    'T1': ['FLAIR', 'T1CE', 'T2', 'T1'],
}


def golden_cfg3_wide():
    """BASELINE config 3 as SURVEY.md section 8(d) item 3 words it: a BraTS-shaped test split over ALL FOUR target contrasts.
    16 distinct synthetic slices, 4 per target ordering; the reference samples each group as one B=4 batch (its own
    sample_from_model, bit-checked against the replay), every step's x_new is stored."""
    print('config 3 (wide): 16 distinct BraTS-shaped slices, 4 per target ordering, 4 steps each (takes ~10 min)')
    cfg = O.default_config()
    u8 = brats_like_slices_u8(16, seed=2020)
    sl = torch.from_numpy(u8.astype(np.float32)) / 255.0 * 2.0 - 1.0
    out = {'slices_u8': u8}
    for gi, (tgt, order) in enumerate(CFG3_ORDERS.items()):
        idx = slice(4 * gi, 4 * gi + 4)
        conds = [sl[idx, CFG3_MODS.index(m)][:, None].contiguous() for m in order[:3]]
        x_init, zs, noises, steps, o_steps = run_sampler(cfg, 1234, 314 + gi, conds, 4)
        for k, (r, o) in enumerate(zip(steps, o_steps)):
            for nm, a, b in zip(('x01', 'x02', 'xnew'), o, r):
                check(f'cfg3w.{tgt}.step{k}.{nm}', a, b, 5e-5)
            out[f'{tgt}.step{k}.xnew'] = r[2].numpy().astype(np.float32)
        print(f'    cfg3w {tgt}: final range [{float(steps[-1][2].min()):.3f},{float(steps[-1][2].max()):.3f}]', flush=True)
    return out


CFG5 = dict(ch_mult=[1, 1, 2, 2, 4], num_timesteps=8, attn_resolutions=(16,))   # BASELINE config 5


def golden_cfg5():
    print('config 5: 256x256, ch_mult 1-1-2-2-4, attention at 16x16 in the down/up paths, 8 steps (takes a few minutes)')
    cfg = O.default_config(**CFG5)
    u8 = np.load(os.path.join(HERE, 'demo_inputs_u8.npz')) if os.path.exists(os.path.join(HERE, 'demo_inputs_u8.npz')) else demo_inputs_u8()
    conds = [preprocess_demo(u8[n]) for n in ('flair', 't2', 't1')]
    x_init, zs, noises, steps, o_steps = run_sampler(cfg, 1234, 42, conds, 1)
    out = {}
    for k, (r, o) in enumerate(zip(steps, o_steps)):
        for nm, a, b in zip(('x01', 'x02', 'xnew'), o, r):
            check(f'cfg5.step{k}.{nm}', a, b, 5e-5)
            if nm == 'xnew' or k in (0, len(steps) - 1):       # every x_new, both predictions at the first and last step
                out[f'step{k}.{nm}'] = b.numpy().astype(np.float32)
    print(f'    cfg5: |x01| std {float(steps[0][0].std()):.3f}, final range [{float(steps[-1][2].min()):.3f},{float(steps[-1][2].max()):.3f}]')
    return out


def ref_volume_namespace():
    """exec engine/test_volume.py:135-181 (robust_minmax_to_minus1_1, extract_center_slices, reconstruct_volume_from_slices):
    the module itself cannot be imported (nibabel is not installed), these three functions are pure numpy."""
    import typing
    ns = {'np': np, 'torch': torch, 'Optional': typing.Optional, 'List': typing.List, 'Tuple': typing.Tuple, 'Dict': typing.Dict}
    lines = open(os.path.join(REF, 'engine/test_volume.py')).read().split('\n')
    exec('\n'.join(lines[134:181]), ns)
    return ns


def golden_volume():
    print('volume pipeline pieces (engine/test_volume.py) + uncertainty map (engine/train.py:957-959)')
    import torch.nn.functional as F
    V = ref_volume_namespace()
    rng = np.random.default_rng(2024)
    out = {}
    # three "modalities" of one synthetic head: smooth blobs inside an ellipsoid, exact zeros outside, a few hot voxels
    X, Y, Z = 20, 24, 11
    xx, yy, zz = np.meshgrid(np.linspace(-1, 1, X), np.linspace(-1, 1, Y), np.linspace(-1, 1, Z), indexing='ij')
    inside = (xx ** 2 + yy ** 2 / 0.8 + zz ** 2 / 1.2) < 0.8
    for m in range(3):
        v = (200 + 80 * np.sin(3 * xx + m) * np.cos(2 * yy - m) + 30 * rng.standard_normal((X, Y, Z))) * inside
        v[rng.integers(0, X, 5), rng.integers(0, Y, 5), rng.integers(0, Z, 5)] = 4000.0      # outliers the percentiles must clip
        v = v.astype(np.float64)                                                            # nibabel's get_fdata() dtype
        out[f'vol{m}'] = v
        r = V['robust_minmax_to_minus1_1'](v)
        check(f'volume.norm{m}', torch.from_numpy(O.robust_minmax_to_minus1_1(v)), torch.from_numpy(r), 0)
        out[f'norm{m}'] = r
    mask = inside & (yy > -0.2)
    vn = out['vol0'].copy(); vn[3, 3, 3] = np.nan
    r = V['robust_minmax_to_minus1_1'](vn, mask=mask, pmin=5.0, pmax=90.0)
    o = O.robust_minmax_to_minus1_1(vn, mask=mask, pmin=5.0, pmax=90.0)
    assert np.array_equal(r, o, equal_nan=True)
    out['mask'] = mask; out['vol_nan'] = vn; out['norm_masked'] = r
    for nm, v in (('zeros', np.zeros((4, 4, 3))), ('flat', np.full((4, 4, 3), 7.0))):
        r = V['robust_minmax_to_minus1_1'](v)
        assert np.array_equal(r, O.robust_minmax_to_minus1_1(v)) and r.dtype == np.float32
        out[f'norm_{nm}'] = r
    bounds = []
    for z, hr in ((11, 3), (11, 80), (155, 80), (1, 0), (6, 2)):
        vol = rng.standard_normal((3, 2, z)).astype(np.float32)
        sl, s0, s1 = V['extract_center_slices'](vol, hr)
        osl, o0, o1 = O.extract_center_slices(vol, hr)
        assert (s0, s1) == (o0, o1) and len(sl) == len(osl) and all(np.array_equal(a, b) for a, b in zip(sl, osl))
        rec = V['reconstruct_volume_from_slices'](sl, vol.shape, s0, s1)
        assert np.array_equal(rec, O.reconstruct_volume_from_slices(osl, vol.shape, o0, o1))
        bounds.append((z, hr, s0, s1, len(sl)))
    out['slice_bounds'] = np.asarray(bounds, dtype=np.int64)
    REPORT.append(('volume.slices', 0.0))
    # bilinear resize (F.interpolate, align_corners=False) as the reference calls it: down, up, non-square, 240 -> 256
    g = torch.Generator().manual_seed(5)
    for tag, shp, size in (('down', (3, 1, 20, 24), (16, 16)), ('up', (2, 1, 12, 10), (32, 32)), ('x8', (2, 1, 4, 4), (32, 32)),
                           ('brats', (1, 1, 240, 240), (256, 256)), ('same', (1, 1, 16, 16), (16, 16))):
        x = torch.randn(*shp, generator=g)
        y = F.interpolate(x, size=size, mode='bilinear', align_corners=False)
        check(f'resize.{tag}', O.resize_bilinear(x, size), y, 0)
        out[f'resize.{tag}.in'] = x; out[f'resize.{tag}.out'] = y
    # uncertainty map: the reference's own conv2d factory (backbones/dense_layer.py) for att_conv, then sigmoid + resize
    from backbones.dense_layer import conv2d as R_conv2d
    att = R_conv2d(64 * 8, 1, 1, padding=0)
    att.weight.data = 0.05 * torch.randn(att.weight.shape, generator=g); att.bias.data = 0.1 * torch.randn(1, generator=g)
    feat = torch.randn(2, 512, 8, 8, generator=g)
    y = F.interpolate(torch.sigmoid(att(feat)), size=(64, 64), mode='bilinear', align_corners=False)
    check('uncertainty_map', O.uncertainty_map(feat, att.weight, att.bias, (64, 64)), y, 0)
    out.update({'att.feat': feat, 'att.w': att.weight.detach(), 'att.b': att.bias.detach(), 'att.out': y})
    return out


VARIANT_BASE = dict(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(16,), num_res_blocks=1)
VARIANTS = {     # every alternate configuration the reference itself can construct and run
    'output_skip': dict(progressive='output_skip'),
    'input_skip_sum': dict(progressive_input='input_skip', progressive_combine='sum'),
    'input_skip_cat': dict(progressive_input='input_skip', progressive_combine='cat'),
    'input_none': dict(progressive_input='none'),
    'fir_false': dict(fir=False),
    'fir_false_input_skip': dict(fir=False, progressive_input='input_skip'),
    'fourier': dict(embedding_type='fourier', fourier_scale=16.0),
    'unconditional': dict(conditional=False),
    'no_rescale': dict(skip_rescale=False),
    'uncentered_notanh': dict(centered=False, not_use_tanh=True),
    'three_levels_all': dict(ch_mult=[1, 1, 2], attn_resolutions=(8,), progressive='output_skip', progressive_input='input_skip',
                             progressive_combine='cat', skip_rescale=False),
    'two_channels': dict(num_channels=2),            # G1 only: the reference's G2 cannot take a 1-channel pseudo-target then
    'healthy': dict(),                               # ncsnpp_generator_adagn_feat_healthy.py: two conditions
}
UNBUILDABLE = {   # recorded for the record: these raise inside the reference, so there is nothing to be compatible with
    'resblock_ddpm': dict(resblock_type='ddpm'), 'resblock_oneadagn': dict(resblock_type='biggan_oneadagn'),
    'progressive_residual': dict(progressive='residual'),
    'fir_false_output_skip': dict(fir=False, progressive='output_skip'),     # Upsample(fir=False): bad F.interpolate call (layerspp.py:164)
}


def golden_variants():
    print('alternate configurations (row f4): one G1 + G2 forward each, B=2, 32x32')
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        # the two-condition twins register the same model names as the main file, so the reference cannot import both in
        # one process: empty its registry first (nothing here looks models up by name)
        from backbones import utils as R_utils
        R_utils._MODELS.clear()
        from backbones import ncsnpp_generator_adagn_feat_healthy as R_H
    out = {}
    g = torch.Generator().manual_seed(31)
    for name, kw in UNBUILDABLE.items():
        cfg = O.default_config(**{**VARIANT_BASE, **kw})
        for cls in (R_G1, R_G2):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    m = cls(cfg)
                    x = torch.randn(2, 1, 32, 32)
                    m(*([x, x, x, x, torch.tensor([1, 3]), torch.randn(2, cfg.nz)] + ([x] if cls is R_G2 else [])))
                raise AssertionError(f'{name}: the reference now builds - restate it')
            except (UnboundLocalError, ValueError) as e:
                print(f'    {name:22s} {cls.__name__:16s} reference raises {type(e).__name__}: {str(e)[:60]}')
    for name, kw in VARIANTS.items():
        cfg = O.default_config(**{**VARIANT_BASE, **kw})
        nc = 2 if name == 'healthy' else 3
        C = cfg.num_channels
        x, c1, c2, c3 = (torch.tanh(torch.randn(2, C, 32, 32, generator=g)) for _ in range(4))
        if not cfg.centered:
            x = (x + 1) / 2
        z = torch.randn(2, cfg.nz, generator=g)
        t = torch.tensor([0.5, 2.0]) if cfg.embedding_type == 'fourier' else torch.tensor([1, 3])
        out.update({f'{name}.x': x, f'{name}.c1': c1, f'{name}.c2': c2, f'{name}.c3': c3, f'{name}.z': z, f'{name}.t': t})
        classes = ((R_H.NCSNpp, R_H.NCSNpp_adaptive) if name == 'healthy' else (R_G1, R_G2))
        y1 = None
        for which, cls in zip(('g1', 'g2'), classes):
            if which == 'g2' and C != 1:
                continue
            with contextlib.redirect_stdout(io.StringIO()):
                m = cls(cfg).eval()
            sd = O.make_state_dict(cfg, which, 77, n_cond=nc)
            rsd = m.state_dict()
            assert list(rsd.keys()) == list(sd.keys()), f'{name}/{which}: state_dict key order differs'
            assert all(tuple(rsd[k].shape) == tuple(sd[k].shape) for k in sd), f'{name}/{which}: shapes differ'
            m.load_state_dict(sd, strict=True)
            conds = [c1, c2] if nc == 2 else [c1, c2, c3]
            with contextlib.redirect_stdout(io.StringIO()):
                if which == 'g1':
                    y = m(x, *conds, t, z); y1 = y
                    o = O.g1_forward(sd, cfg, x, c1, c2, c3 if nc == 3 else None, t, z)
                else:
                    y = m(x, *conds, t, z, y1[:, [0], :])
                    o = O.g2_forward(sd, cfg, x, c1, c2, c3 if nc == 3 else None, t, z, y1[:, [0], :])
            check(f'variant.{name}.{which}', o, y, 2e-5)
            assert float(y.std()) > 0.05
            out[f'{name}.{which}'] = y
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--check', action='store_true', help='validate the oracle only, write nothing')
    ap.add_argument('--skip-full', action='store_true')
    ap.add_argument('--only-cfg5', action='store_true', help='(re)generate full_cfg5.npz alone')
    ap.add_argument('--only-cfg3', action='store_true', help='(re)generate batch_cfg3.npz alone')
    ap.add_argument('--only-cfg3-wide', action='store_true', help='(re)generate wide_cfg3.npz alone')
    ap.add_argument('--only-volume', action='store_true', help='(re)generate volume.npz alone')
    ap.add_argument('--only-variants', action='store_true', help='(re)generate variants.npz alone')
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.only_cfg3:
        d = golden_cfg3()
        if not a.check:
            np.savez_compressed(os.path.join(HERE, 'batch_cfg3.npz'), **d)
            print('wrote batch_cfg3.npz', f'{os.path.getsize(os.path.join(HERE, "batch_cfg3.npz")) / 1e6:.2f} MB')
        return
    if a.only_cfg3_wide:
        d = golden_cfg3_wide()
        if not a.check:
            np.savez_compressed(os.path.join(HERE, 'wide_cfg3.npz'), **d)
            print('wrote wide_cfg3.npz', f'{os.path.getsize(os.path.join(HERE, "wide_cfg3.npz")) / 1e6:.2f} MB')
        return
    if a.only_variants:
        d = t2n(golden_variants())
        if not a.check:
            np.savez_compressed(os.path.join(HERE, 'variants.npz'), **d)
            print('wrote variants.npz', f'{os.path.getsize(os.path.join(HERE, "variants.npz")) / 1e6:.2f} MB')
        return
    if a.only_volume:
        d = t2n(golden_volume())
        if not a.check:
            np.savez_compressed(os.path.join(HERE, 'volume.npz'), **d)
            print('wrote volume.npz', f'{os.path.getsize(os.path.join(HERE, "volume.npz")) / 1e6:.2f} MB')
        return
    if a.only_cfg5:
        d = golden_cfg5()
        if not a.check:
            np.savez_compressed(os.path.join(HERE, 'full_cfg5.npz'), **d)
            print('wrote full_cfg5.npz', f'{os.path.getsize(os.path.join(HERE, "full_cfg5.npz")) / 1e6:.2f} MB')
        return
    files = {
        'kat_schedules.npz': t2n(golden_schedules()),
        'elementwise.npz': t2n(golden_elementwise()),
        'fir.npz': t2n(golden_fir()),
        'blocks.npz': t2n(golden_blocks()),
        'small_models.npz': t2n(golden_small_models()),
        'critic.npz': t2n(golden_discriminator()),
        'volume.npz': t2n(golden_volume()),
        'variants.npz': t2n(golden_variants()),
    }
    if not a.skip_full:
        u8, full = golden_full()
        files['demo_inputs_u8.npz'] = u8
        files['full_cfg2.npz'] = full
        files['full_cfg5.npz'] = golden_cfg5()
        files['batch_cfg3.npz'] = golden_cfg3()
        files['wide_cfg3.npz'] = golden_cfg3_wide()
    worst = max(REPORT, key=lambda r: r[1])
    print(f'{len(REPORT)} comparisons, worst: {worst[0]} {worst[1]:.3e}')
    if not a.check:
        for fn, d in files.items():
            np.savez_compressed(os.path.join(HERE, fn), **d)
            print('wrote', fn, f'{os.path.getsize(os.path.join(HERE, fn)) / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
