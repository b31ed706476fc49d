"""checkpoint-loader semantics, dataset normalisation, 8-bit export + metrics (CPU), and the
batched sharded driver end to end on synthetic .npy volumes (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import mudiff_oracle as O
from helpers import SMALL_CFGS


def test_checkpoint_loader_strips_seven_chars_and_is_non_strict(tmp_path):
    from mudiff_hip import driver
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp
    cfg = O.default_config(**SMALL_CFGS['s32na'])
    sd = O.make_state_dict(cfg, 'g1', 9)
    ckpt = {'module.' + k: v for k, v in sd.items()}
    ckpt['module.not_a_parameter'] = torch.zeros(3)                  # silently ignored (strict=False), like the reference
    os.makedirs(tmp_path / 'exp1')
    torch.save(ckpt, tmp_path / 'exp1' / 'gen_diffusive_1.pth')
    m = NCSNpp(cfg)
    m.train()
    driver.load_checkpoint_with_fallback(str(tmp_path), 'exp1', m, 'gen_diffusive_1', device='cpu')   # secondary location
    assert not m.training
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    with pytest.raises(FileNotFoundError):
        driver.load_checkpoint_with_fallback(str(tmp_path), 'nope', m, 'gen_diffusive_2', device='cpu')


def _write_volumes(root, n=5, hw=32, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'test'), exist_ok=True)
    vols = {}
    for mod in ('T1', 'T2', 'FLAIR', 'T1CE'):
        vols[mod] = (rng.standard_normal((n, hw, hw)) * 2).astype(np.float32)
        np.save(os.path.join(root, 'test', mod + '.npy'), vols[mod])
    return vols


def test_slice_source_order_and_normalisation(tmp_path):
    from mudiff_hip.driver import ORDERS, SliceSource
    vols = _write_volumes(str(tmp_path))
    for target, order in ORDERS.items():
        src = SliceSource('test', str(tmp_path), target)
        assert len(src) == 5
        c1, c2, c3, y = src.batch(1, 4)
        for t, mod in zip((c1, c2, c3, y), order):
            ref = torch.clamp(torch.from_numpy(vols[mod][1:4]), -3.0, 3.0) / 3.0
            assert t.shape == (3, 1, 32, 32) and torch.equal(t[:, 0], ref)
    with pytest.raises(ValueError):
        SliceSource('test', str(tmp_path), 'PD')


def test_export_and_metrics_match_oracle_metrics(tmp_path):
    from mudiff_hip import driver
    rng = np.random.default_rng(1)
    gts = [rng.uniform(-1, 1, (40, 40)).astype(np.float32) for _ in range(3)]
    preds = [np.clip(g + 0.05 * rng.standard_normal(g.shape), -1.2, 1.1).astype(np.float32) for g in gts]
    res = driver.export_and_score(preds, gts, str(tmp_path / 'out'))
    assert res['count'] == 3 and sorted(os.listdir(tmp_path / 'out' / 'pred')) == ['pred_00000.png', 'pred_00001.png', 'pred_00002.png']
    from PIL import Image
    gmin, gmax = res['global_min'], res['global_max']
    ps = ss = 0.0
    for i, (p, g) in enumerate(zip(preds, gts)):
        p8 = np.array(Image.open(tmp_path / 'out' / 'pred' / f'pred_{i:05d}.png'))
        assert np.array_equal(p8, np.clip((p - gmin) / (gmax - gmin) * 255.0, 0, 255).astype(np.uint8))
        g8 = np.array(Image.open(tmp_path / 'out' / 'gt' / f'gt_{i:05d}.png'))
        ps += O.psnr(g8 / 255.0, p8 / 255.0)
        ss += O.ssim(g8 / 255.0, p8 / 255.0)
    assert res['psnr'] == pytest.approx(ps / 3, abs=1e-4) and res['ssim'] == pytest.approx(ss / 3, abs=1e-6)


def _draws_for(cfg, n_total, hw, seed=11):
    """Per-slice draws indexed by GLOBAL slice number (parity runs must not depend on batching / sharding)."""
    g = torch.Generator().manual_seed(seed)
    x_init = torch.randn(n_total, 1, hw, hw, generator=g)
    zs = [torch.randn(n_total, cfg.nz, generator=g) for _ in range(cfg.num_timesteps)]
    noises = [torch.randn(n_total, 1, hw, hw, generator=g) for _ in range(cfg.num_timesteps)]
    return x_init, zs, noises, (lambda lo, n: (x_init[lo:lo + n], [z[lo:lo + n] for z in zs], [e[lo:lo + n] for e in noises]))


@pytest.mark.gpu
def test_batched_driver_vs_oracle_with_injected_draws(tmp_path):
    """Row f2 numerically: driver.sample_slices (batches of 4 through the captured graph, 7 slices = one full batch + a
    padded one) with injected per-slice draws against the oracle's sample_from_model on the same slices, <= 1e-3; and the
    union of two rank shards equals the single-rank run (reference engine/test.py:325-336 samples one slice at a time)."""
    from mudiff_hip import driver
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    _write_volumes(str(tmp_path), n=7, hw=32, seed=3)
    cfg = O.default_config(**SMALL_CFGS['s32'])
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    g1.load_state_dict(sd1); g2.load_state_dict(sd2)
    g1, g2 = g1.cuda().eval(), g2.cuda().eval()
    dev = torch.device('cuda:0')
    for target in ('T1CE', 'FLAIR', 'T2', 'T1'):        # all four contrast orderings (reference dataset/dataset_brats.py:29-34)
        src = driver.SliceSource('test', str(tmp_path), target)
        x_init, zs, noises, draws = _draws_for(cfg, 7, 32)
        lo, preds, gts = driver.sample_slices(cfg, g1, g2, src, 4, dev, draws=draws)
        assert lo == 0 and preds.shape == (7, 32, 32) and gts.shape == (7, 32, 32)
        c1, c2, c3, y = src.batch(0, 7)
        ref = O.sample_from_model(O.PosteriorCoefficients(cfg), sd1, sd2, cfg, c1, c2, c3, x_init, zs, noises)
        err = float(np.abs(preds - ref[:, 0].numpy()).max())
        print(f'driver vs oracle ({target}, 7 slices in batches of 4): max-abs {err:.2e}')
        assert err <= 1e-3
        assert np.array_equal(gts, y[:, 0].numpy())
        # two "ranks" cover the same slices exactly once, and sharding changes nothing
        lo0, p0, _ = driver.sample_slices(cfg, g1, g2, src, 4, dev, rank=0, world=2, draws=draws)
        lo1, p1, _ = driver.sample_slices(cfg, g1, g2, src, 4, dev, rank=1, world=2, draws=draws)
        assert (lo0, p0.shape[0], lo1, p1.shape[0]) == (0, 4, 4, 3)
        assert float(np.abs(np.concatenate([p0, p1], 0) - preds).max()) <= 1e-4      # run-to-run: fp64 atomics order only (measured 2.4e-5)


@pytest.mark.gpu
def test_batched_driver_end_to_end(tmp_path):
    from mudiff_hip import driver
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    _write_volumes(str(tmp_path), n=7, hw=32, seed=3)
    cfg = O.default_config(**SMALL_CFGS['s32'])
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    g1.load_state_dict(O.make_state_dict(cfg, 'g1', 1234)); g2.load_state_dict(O.make_state_dict(cfg, 'g2', 1234))
    g1, g2 = g1.cuda().eval(), g2.cuda().eval()
    src = driver.SliceSource('test', str(tmp_path), 'T1CE')
    dev = torch.device('cuda:0')
    lo, preds, gts = driver.sample_slices(cfg, g1, g2, src, 4, dev)      # device-drawn noise, seed 42
    assert lo == 0 and preds.shape == (7, 32, 32) and gts.shape == (7, 32, 32) and np.isfinite(preds).all()
    assert np.abs(preds).max() <= 1.0 + 1e-5                             # tanh output range survives the posterior at t = 0
    _, again, _ = driver.sample_slices(cfg, g1, g2, src, 4, dev)        # reproducible from `seed`
    assert float(np.abs(again - preds).max()) <= 1e-4
    _, other, _ = driver.sample_slices(cfg, g1, g2, src, 4, dev, seed=43)
    assert float(np.abs(other - preds).max()) > 1e-3
    _, r1, _ = driver.sample_slices(cfg, g1, g2, src, 4, dev, rank=1, world=2)   # another rank draws another stream
    assert float(np.abs(r1 - preds[4:]).max()) > 1e-3
    res = driver.export_and_score(list(preds), list(gts), str(tmp_path / 'png'))
    assert res['count'] == 7 and np.isfinite(res['psnr']) and 0 < res['ssim'] < 1


@pytest.mark.gpu
def test_driver_cli_end_to_end(tmp_path):
    """`python -m mudiff_hip.driver` as a user runs it (the counterpart of `python engine/test.py ...`, reference engine/test.py:400-491):
    DDP-prefixed checkpoints found through the fallback directory, .npy volumes, batched sampling, PNG export, metrics in the log."""
    import subprocess
    import sys
    from conftest import PKG, REPO
    data, out = tmp_path / 'data', tmp_path / 'out'
    _write_volumes(str(data), n=6, hw=32, seed=5)
    cfg = O.default_config(**SMALL_CFGS['s32'])
    os.makedirs(out / 'exp7')
    for which, name in (('g1', 'gen_diffusive_1'), ('g2', 'gen_diffusive_2')):
        torch.save({'module.' + k: v for k, v in O.make_state_dict(cfg, which, 1234).items()}, out / 'exp7' / f'{name}.pth')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, PKG, os.environ.get('PYTHONPATH', '')]))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'mudiff_hip.driver', '--input_path', str(data), '--output_path', str(out), '--exp', 'exp7', '--target_modality', 'T2',
           '--image_size', '32', '--num_channels_dae', '32', '--ch_mult', '1', '2', '4', '--attn_resolutions', '16', '--batch_size', '4']
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert 'Average PSNR' in p.stderr and 'over 6 slices' in p.stderr
    pngs = sorted(os.listdir(out / 'generated_samples' / 'pred'))
    assert pngs == [f'pred_{i:05d}.png' for i in range(6)] and len(os.listdir(out / 'generated_samples' / 'gt')) == 6
