"""CPU: the oracle restatement against the golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py).  The reference is never imported here."""
import numpy as np
import pytest
import torch

from oracle import mudiff_oracle as O
from helpers import SMALL_CFGS, demo_conds, load_golden, sampler_inputs, small_conds

torch.set_grad_enabled(False)


def test_kat_tables_survey_section4():
    """Closed-form values quoted in SURVEY.md section 4 (T=4, beta in [0.1, 20])."""
    c = O.PosteriorCoefficients(O.default_config())
    np.testing.assert_allclose(c.betas.numpy(), [0.47825530, 0.84920603, 0.95641768, 0.98740393], rtol=2e-7)
    np.testing.assert_allclose(c.posterior_mean_coef1.numpy(), [1.0000001, 0.66577846, 0.26919085, 0.057821561], rtol=3e-7)
    np.testing.assert_allclose(c.posterior_mean_coef2.numpy(), [0, 0.20157626, 0.19300087, 0.11185222], rtol=3e-7)
    np.testing.assert_allclose(c.posterior_log_variance_clipped.numpy(), [-46.051701, -0.81912059, -0.12306926, -0.016067632], rtol=3e-7)
    np.testing.assert_allclose(O.get_time_schedule(O.default_config()).numpy(), [0.001, 0.25075, 0.5005, 0.75025, 1.0])


@pytest.mark.parametrize('tag,kw', [('T4', dict(num_timesteps=4)), ('T8', dict(num_timesteps=8)),
                                    ('T4geo', dict(num_timesteps=4, use_geometric=True, beta_min=0.01, beta_max=0.9))])
def test_schedules_bit_exact(tag, kw):
    g = load_golden('kat_schedules.npz')
    cfg = O.default_config(**kw)
    p, d = O.PosteriorCoefficients(cfg), O.DiffusionCoefficients(cfg)
    # bit-exact in the build container; 1 ulp allowed because the float64 exp/log behind the tables
    # depend on the host CPU's SIMD dispatch
    def close(a, b, name):
        torch.testing.assert_close(a, b, rtol=3e-7, atol=1e-30, msg=name)
    for f in ('betas', 'alphas_cumprod', 'posterior_variance', 'posterior_mean_coef1', 'posterior_mean_coef2',
              'posterior_log_variance_clipped'):
        close(getattr(p, f), g[f'{tag}.{f}'], f)
    for f in ('sigmas', 'a_s', 'a_s_cum', 'sigmas_cum', 'a_s_prev'):
        close(getattr(d, f), g[f'{tag}.{f}'], f)
    close(O.get_time_schedule(cfg), g[f'{tag}.T'], 'T')


def test_posterior_and_q_sample():
    g = load_golden('elementwise.npz')
    cfg = O.default_config()
    p, d = O.PosteriorCoefficients(cfg), O.DiffusionCoefficients(cfg)
    def close(a, b):
        assert (a - b).abs().max() <= 4.8e-7      # 0 in the build container; tables may move 1 ulp on another CPU
    close(O.sample_posterior_combine(p, g['x01'], g['x02'], g['xt'], g['t'], g['noise']), g['posterior_combine'])
    close(O.sample_posterior(p, g['x01'], g['xt'], g['t'], g['noise']), g['posterior'])
    close(O.q_sample(d, g['x01'], g['t'], g['noise']), g['q_sample'])
    a, b = O.q_sample_pairs(d, g['x01'], g['t'], g['noise_inner'], g['noise_outer'])
    close(a, g['q_pair0']); close(b, g['q_pair1'])
    # t == 0 rows carry no noise (engine/test.py:171-173)
    z = O.sample_posterior_combine(p, g['x01'], g['x02'], g['xt'], g['t'], torch.zeros_like(g['noise']))
    assert torch.equal(z[0], g['posterior_combine'][0]) and torch.equal(z[4], g['posterior_combine'][4])


def test_fir_closed_forms_and_golden():
    g = load_golden('fir.npz')
    for tag in 'abc':
        x = g[f'{tag}.x']
        assert (O.upsample_2d(x) - g[f'{tag}.up']).abs().max() <= 1e-6
        assert (O.downsample_2d(x) - g[f'{tag}.down']).abs().max() <= 1e-6
        assert (O.conv_downsample_2d(x, g[f'{tag}.w']) - g[f'{tag}.convdown']).abs().max() <= 2e-6
    for tag, (u, d, pad) in (('g1', (1, 1, (2, 1))), ('g2', (2, 1, (2, 1))), ('g3', (1, 2, (1, 1))), ('g4', (2, 2, (3, 0)))):
        assert (O.upfirdn2d(g['g.x'], g['g.k'], up=u, down=d, pad=pad) - g[f'{tag}.out']).abs().max() <= 2e-6
    # SURVEY.md section 4 closed forms (separable polyphase, zero padded)
    x = g['c.x'][0, 0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))

    def up1d(v):           # along last axis of a padded array
        c = v[..., 1:-1]
        even = (v[..., :-2] + 3 * c) / 4
        odd = (3 * c + v[..., 2:]) / 4
        return torch.stack([even, odd], -1).reshape(*c.shape[:-1], -1)
    rows = up1d(xp[1:-1])
    cols = up1d(torch.nn.functional.pad(rows.t(), (1, 1)))
    assert (cols.t() - O.upsample_2d(g['c.x'])[0, 0]).abs().max() < 1e-6


def test_blocks_golden():
    g = load_golden('blocks.npz')

    def sd(prefix):
        return {'m.' + k[len(prefix) + 1:]: v for k, v in g.items() if k.startswith(prefix + '.')}
    for tag, up, down in (('plain', 0, 0), ('skip', 0, 0), ('up', 1, 0), ('down', 0, 1), ('cat', 0, 0)):
        y = O.resblock(sd(f'res_{tag}.sd'), 'm', g[f'res_{tag}.x'], g['temb'], g['zemb'], up=bool(up), down=bool(down))
        assert (y - g[f'res_{tag}.y']).abs().max() <= 5e-6, tag
    assert (O.adagn(sd('adagn.sd'), 'm', g['adagn.x'], g['zemb']) - g['adagn.y']).abs().max() <= 2e-6
    for tag in ('c16', 'c32'):
        assert (O.attn_block(sd(f'attn_{tag}.sd'), 'm', g[f'attn_{tag}.x']) - g[f'attn_{tag}.y']).abs().max() <= 5e-6
    assert (O.conv_feat_block(sd('feat.sd'), 'm', g['feat.x']) - g['feat.y']).abs().max() <= 5e-6
    assert (O.conv_block(sd('ada.sd'), 'm', g['feat.x'], g['zemb']) - g['ada.y']).abs().max() <= 5e-6
    assert (O.conv_block_gap(sd('gap.sd'), 'm', g['feat.x']) - g['gap.y']).abs().max() <= 5e-6
    for tag in ('p1', 'p8'):
        assert (O.pyramid_downsample(sd(f'pyr_{tag}.sd'), 'm', g[f'pyr_{tag}.x']) - g[f'pyr_{tag}.y']).abs().max() <= 5e-6
    assert torch.equal(O.timestep_embedding(g['temb.t'], 64), g['temb.y'])


@pytest.mark.parametrize('tag', list(SMALL_CFGS))
def test_small_models_every_step(tag):
    g = load_golden('small_models.npz')
    cfg = O.default_config(**SMALL_CFGS[tag])
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    conds = small_conds(cfg)
    x_init, zs, noises = sampler_inputs(cfg, 2)
    assert torch.equal(x_init, g[f'{tag}.x_init']) and torch.equal(conds[1], g[f'{tag}.c2'])
    assert torch.equal(zs[1], g[f'{tag}.z1']) and torch.equal(noises[0], g[f'{tag}.noise0'])
    _, steps = O.sample_from_model(O.PosteriorCoefficients(cfg), sd1, sd2, cfg, *conds, x_init, zs, noises, return_steps=True)
    for k, st in enumerate(steps):
        for nm, v in zip(('x01', 'x02', 'xnew'), st):
            assert (v - g[f'{tag}.step{k}.{nm}']).abs().max() <= 2e-5, (k, nm)


CRITIC_CASES = (('d8', 2, 8, 32), ('d16', 2, 16, 64), ('d8b8', 2, 8, 32))


@pytest.mark.parametrize('tag,nc,ngf,td', CRITIC_CASES)
def test_critic_forward_golden(tag, nc, ngf, td):
    """Discriminator_large inference forward (logit and mid_feat) vs the reference's outputs."""
    g = load_golden('critic.npz')
    sd = O.make_discriminator_state_dict(nc, ngf, td, 1234)
    logit, mid = O.discriminator_large_forward(sd, g[f'{tag}.x'], g[f'{tag}.t'], g[f'{tag}.xt'], td)
    assert (logit - g[f'{tag}.logit']).abs().max() <= 2e-5 and (mid - g[f'{tag}.mid']).abs().max() <= 2e-5


def test_critic_param_count_matches_reference_log():
    """D = 27,736,705 parameters at nc=2, ngf=64, t_emb_dim=256 (reference error_logs/log_mudiff_T1.13967221.out:116)."""
    assert sum(int(np.prod(s)) for s in O.discriminator_param_spec(2, 64, 256).values()) == 27736705


def test_param_counts_match_reference_log():
    """error_logs/log_mudiff_T1.13967221.out:116 of the reference."""
    cfg = O.default_config()
    n1 = sum(int(np.prod(s)) for s in O.param_spec(cfg, 'g1').values())
    n2 = sum(int(np.prod(s)) for s in O.param_spec(cfg, 'g2').values())
    assert (n1, n2) == (20472065, 21399681)
    assert len(O.param_spec(cfg, 'g1')) == 288 and len(O.param_spec(cfg, 'g2')) == 318


def test_config1_single_g1_forward_full_size():
    """BASELINE config 1 (plumbing): one G1 forward at t=3 on the demo JPEG inputs, 256x256, CPU."""
    g = load_golden('full_cfg2.npz')
    cfg = O.default_config()
    sd1 = O.make_state_dict(cfg, 'g1', 1234)
    conds = demo_conds()
    x_init, zs, _ = sampler_inputs(cfg, 1)
    y = O.g1_forward(sd1, cfg, x_init, *conds, torch.full((1,), 3, dtype=torch.int64), zs[0])
    assert (y - g['step0.x01']).abs().max() <= 5e-5


def test_config5_first_step_full_size():
    """BASELINE config 5 (8 steps, ch_mult 1-1-2-2-4, attention at 16x16 in the down/up paths): G1 and G2 at the first
    reverse step (t=7) against the reference's own outputs."""
    g = load_golden('full_cfg5.npz')
    cfg = O.default_config(ch_mult=[1, 1, 2, 2, 4], num_timesteps=8, attn_resolutions=(16,))
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    conds = demo_conds()
    x_init, zs, _ = sampler_inputs(cfg, 1)
    t = torch.full((1,), 7, dtype=torch.int64)
    y1 = O.g1_forward(sd1, cfg, x_init, *conds, t, zs[0])
    y2 = O.g2_forward(sd2, cfg, x_init, *conds, t, zs[0], y1[:, [0], :])
    assert (y1 - g['step0.x01']).abs().max() <= 5e-5
    assert (y2 - g['step0.x02']).abs().max() <= 5e-5


def test_metrics_sanity():
    rng = np.random.default_rng(0)
    a = rng.random((64, 64))
    assert O.ssim(a, a) == pytest.approx(1.0)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1)
    assert 0 < O.ssim(a, b) < 1
    assert O.psnr(a, b) == pytest.approx(10 * np.log10(1 / np.mean((a - b) ** 2)))
