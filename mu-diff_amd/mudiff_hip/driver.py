"""Batched test driver for the MI355X sampling path.

Reproduces the observable behaviour of the reference's `engine/test.py::sample_and_test` (:265-396) and
`tools/metric_calc.py` (:28-53) - checkpoint loading semantics, dataset normalisation, global min/max 8-bit PNG
export, PSNR / SSIM / MAE on the quantised images - but samples slices in batches through one captured hipGraph
per GPU and shards the slice list over ranks (the reference runs batch_size = 1 on one GPU).

    python -m mudiff_hip.driver --input_path data/BRATS --output_path results --target_modality T1CE \\
           --image_size 256 --num_channels 1 --num_channels_dae 64 --ch_mult 1 2 4 --batch_size 16
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mudiff_hip.driver ...   # 8 GPUs

LPIPS (needs downloaded AlexNet weights) is out of scope.  skimage is not installed here, so PSNR / SSIM are
restated from the published definitions with skimage's defaults ("parity unpinned")."""
from __future__ import annotations

import argparse
import logging
import os
from types import SimpleNamespace

import numpy as np
import torch

ORDERS = {      # condition order per target contrast (reference dataset/dataset_brats.py:29-34)
    'T1CE': ['FLAIR', 'T2', 'T1', 'T1CE'],
    'FLAIR': ['T1CE', 'T1', 'T2', 'FLAIR'],
    'T2': ['T1CE', 'T1', 'FLAIR', 'T2'],
    'T1': ['FLAIR', 'T1CE', 'T2', 'T1'],
}


# ---------------------------------------------------------------------------------------------------
def load_checkpoint(checkpoint_dir, netG, name_of_network, device='cuda:0'):
    """Reference engine/test.py:202-212: `checkpoint_dir` is a '{}.pth' pattern; every key loses its first 7
    characters (the DDP 'module.' prefix) unconditionally; strict=False; eval()."""
    ckpt = torch.load(checkpoint_dir.format(name_of_network), map_location=device, weights_only=True)
    for key in list(ckpt.keys()):
        ckpt[key[7:]] = ckpt.pop(key)
    netG.load_state_dict(ckpt, strict=False)
    netG.eval()


def load_checkpoint_with_fallback(output_dir, exp, netG, name_of_network, device='cuda:0'):
    """Reference engine/test.py:215-232: <output_dir>/<name>.pth, else <output_dir>/<exp>/<name>.pth."""
    for pattern in (os.path.join(output_dir, '{}.pth'), os.path.join(output_dir, exp, '{}.pth')):
        if os.path.isfile(pattern.format(name_of_network)):
            logging.info('Loading checkpoint %s', pattern.format(name_of_network))
            return load_checkpoint(pattern, netG, name_of_network, device=device)
    raise FileNotFoundError(f"Checkpoint not found for {name_of_network} in '{output_dir}' or '{os.path.join(output_dir, exp)}'")


class SliceSource:
    """<base_path>/<split>/<MOD>.npy volumes of z-scored slices [N,H,W] -> clamp(+-3)/3 in [-1,1]; three condition
    contrasts + target in the ORDERS order (reference dataset/dataset_brats.py:36-92).  Memory-mapped."""

    def __init__(self, split='test', base_path='data/BRATS', target_modality='T1CE'):
        if target_modality not in ORDERS:
            raise ValueError(f'Invalid target_modality {target_modality}.')
        self.order = ORDERS[target_modality]
        self.arrays = []
        for mod in self.order:
            fp = os.path.join(base_path, split, f'{mod}.npy')
            if not os.path.isfile(fp):
                raise FileNotFoundError(fp)
            self.arrays.append(np.load(fp, mmap_mode='r', allow_pickle=False))
        self.length = self.arrays[0].shape[0]

    def __len__(self):
        return self.length

    def batch(self, lo, hi):
        """-> (cond1, cond2, cond3, target), each float32 [hi-lo, 1, H, W] on the host."""
        out = []
        for arr in self.arrays:
            t = torch.from_numpy(np.ascontiguousarray(arr[lo:hi]).astype(np.float32))
            out.append((torch.clamp(t, -3.0, 3.0) / 3.0).unsqueeze(1))
        return out


# ---------------------------------------------------------------------------------------------------
def psnr(gt, pred, data_range=1.0):
    mse = np.mean((np.asarray(gt, np.float64) - np.asarray(pred, np.float64)) ** 2)
    return float('inf') if mse == 0 else float(10.0 * np.log10(data_range ** 2 / mse))


def ssim(gt, pred, data_range=1.0, win=7, k1=0.01, k2=0.03):
    """skimage.metrics.structural_similarity defaults: 7x7 uniform window, sample covariance, border cropped."""
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(gt, np.float64), np.asarray(pred, np.float64)
    cov_norm = win * win / (win * win - 1.0)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    vx = cov_norm * (uniform_filter(a * a, win) - ux * ux)
    vy = cov_norm * (uniform_filter(b * b, win) - uy * uy)
    vxy = cov_norm * (uniform_filter(a * b, win) - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def to_uint8(slices, global_min, global_max):
    """Reference engine/test.py:386-387: clip((x - min)/(max - min) * 255, 0, 255).astype(uint8) with GLOBAL min/max."""
    return [np.clip((s - global_min) / (global_max - global_min) * 255.0, 0, 255).astype(np.uint8) for s in slices]


def export_and_score(pred_slices, gt_slices, save_dir=None):
    """Global-range 8-bit quantisation (+ optional PNG export, reference :370-390) and the metric_calc.py scores on
    the quantised images.  -> dict(psnr, ssim, mae, count, global_min, global_max)."""
    gmin = float(min(min(p.min() for p in pred_slices), min(g.min() for g in gt_slices)))
    gmax = float(max(max(p.max() for p in pred_slices), max(g.max() for g in gt_slices)))
    if gmax <= gmin:
        gmin, gmax = 0.0, 1.0
    pred8, gt8 = to_uint8(pred_slices, gmin, gmax), to_uint8(gt_slices, gmin, gmax)
    if save_dir is not None:
        from PIL import Image
        os.makedirs(os.path.join(save_dir, 'pred'), exist_ok=True)
        os.makedirs(os.path.join(save_dir, 'gt'), exist_ok=True)
        for i, (p, g) in enumerate(zip(pred8, gt8)):
            Image.fromarray(p).save(os.path.join(save_dir, 'pred', f'pred_{i:05d}.png'))
            Image.fromarray(g).save(os.path.join(save_dir, 'gt', f'gt_{i:05d}.png'))
    ps = ss = ma = 0.0
    for p, g in zip(pred8, gt8):
        pn, gn = p.astype(np.float32) / 255.0, g.astype(np.float32) / 255.0
        ps += psnr(gn, pn)
        ss += ssim(gn, pn)
        ma += float(np.mean(np.abs(gn - pn)))
    n = max(len(pred8), 1)
    return dict(psnr=ps / n, ssim=ss / n, mae=ma / n, count=len(pred8), global_min=gmin, global_max=gmax)


# ---------------------------------------------------------------------------------------------------
def sample_slices(args, gen1, gen2, source, batch_size, device, rank=0, world=1, seed=42, progress=None, draws=None):
    """Sample this rank's contiguous shard of `source` in batches of `batch_size` through one captured reverse step.
    -> (lo, predictions [n,H,W] float32 numpy, targets [n,H,W]).
    Draws: x_init, z and the posterior noise come from ONE device generator seeded with `seed + rank` (so a run is
    reproducible from `seed`, and ranks do not repeat each other's streams).  `draws(lo, n) -> (x_init [n,1,H,W],
    zs [T][n,nz], noises [T][n,1,H,W])` (host tensors, indexed by GLOBAL slice number) injects them instead - parity runs
    must not depend on how the slices are batched or sharded."""
    from . import sampling as S
    from .distributed import shard_range
    lo, hi = shard_range(len(source), rank, world)
    device = torch.device(device)
    coef = S.Posterior_Coefficients(args, device)
    preds, gts = [], []
    sampler = None
    gen = torch.Generator(device=device).manual_seed(seed + rank)

    def pad(t, n):      # last partial batch: repeat the last slice up to the fixed graph shape, trimmed afterwards
        return t if t.shape[0] == batch_size else torch.cat([t, t[-1:].expand(batch_size - n, *t.shape[1:])], 0)

    for b0 in range(lo, hi, batch_size):
        c1, c2, c3, y = source.batch(b0, min(b0 + batch_size, hi))
        n = c1.shape[0]
        c1, c2, c3 = (pad(c, n) for c in (c1, c2, c3))
        if sampler is None:
            sampler = S.GraphSampler(coef, gen1, gen2, args, batch_size, c1.shape[2], c1.shape[3], device)
        if draws is not None:
            x_init, zs, noises = draws(b0, n)
            out = sampler.sample(c1.to(device), c2.to(device), c3.to(device), pad(x_init, n).to(device), args.num_timesteps,
                                 zs=[pad(z, n).to(device) for z in zs], noises=[pad(e, n).to(device) for e in noises])
        else:
            x_init = torch.randn(batch_size, 1, c1.shape[2], c1.shape[3], device=device, generator=gen)
            out = sampler.sample(c1.to(device), c2.to(device), c3.to(device), x_init, args.num_timesteps, generator=gen)
        preds.append(out[:n, 0].cpu().numpy())
        gts.append(y[:, 0].numpy())
        if progress:
            progress(min(b0 + batch_size, hi) - lo, hi - lo)
    cat = (lambda xs: np.concatenate(xs, 0) if xs else np.zeros((0, 1, 1), np.float32))
    return lo, cat(preds), cat(gts)


def build_parser():
    p = argparse.ArgumentParser('mudiff MI355X batched test driver (flags as in the reference engine/test.py:400-484)')
    p.add_argument('--centered', action='store_false', default=True)
    p.add_argument('--use_geometric', action='store_true', default=False)
    p.add_argument('--beta_min', type=float, default=0.1)
    p.add_argument('--beta_max', type=float, default=20.)
    p.add_argument('--num_channels', type=int, default=1)
    p.add_argument('--num_channels_dae', type=int, default=64)
    p.add_argument('--n_mlp', type=int, default=3)
    p.add_argument('--ch_mult', nargs='+', type=int, default=[1, 2, 4])
    p.add_argument('--num_res_blocks', type=int, default=2)
    p.add_argument('--attn_resolutions', nargs='+', type=int, default=[16])
    p.add_argument('--dropout', type=float, default=0.)
    p.add_argument('--resamp_with_conv', action='store_false', default=True)
    p.add_argument('--conditional', action='store_false', default=True)
    p.add_argument('--fir', action='store_false', default=True)
    p.add_argument('--fir_kernel', nargs='+', type=int, default=[1, 3, 3, 1])
    p.add_argument('--skip_rescale', action='store_false', default=True)
    p.add_argument('--resblock_type', default='biggan')
    p.add_argument('--progressive', default='none')
    p.add_argument('--progressive_input', default='residual')
    p.add_argument('--progressive_combine', default='sum')
    p.add_argument('--embedding_type', default='positional')
    p.add_argument('--fourier_scale', type=float, default=16.)
    p.add_argument('--not_use_tanh', action='store_true', default=False)
    p.add_argument('--exp', default='ixi_synth')
    p.add_argument('--input_path', default='data/BRATS')
    p.add_argument('--output_path', default='./results')
    p.add_argument('--image_size', type=int, default=256)
    p.add_argument('--nz', type=int, default=100)
    p.add_argument('--num_timesteps', type=int, default=4)
    p.add_argument('--z_emb_dim', type=int, default=256)
    p.add_argument('--t_emb_dim', type=int, default=256)
    p.add_argument('--batch_size', type=int, default=16, help='slices per GPU per captured reverse step')
    p.add_argument('--target_modality', default='T1CE')
    p.add_argument('--no_png', action='store_true')
    return p


def main(argv=None):
    import torch.distributed as dist
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    from .distributed import broadcast_parameters
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s | %(levelname)s | %(message)s')
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1), ('LOCAL_RANK', 0)))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', init_method='env://', device_id=device)
    torch.manual_seed(42)                                        # like the reference (engine/test.py:266)
    g1, g2 = NCSNpp(args).to(device), NCSNpp_adaptive(args).to(device)
    if rank == 0:                                                # one reader, one flattened RCCL broadcast per model
        load_checkpoint_with_fallback(args.output_path, args.exp, g1, 'gen_diffusive_1', device=device)
        load_checkpoint_with_fallback(args.output_path, args.exp, g2, 'gen_diffusive_2', device=device)
    broadcast_parameters(g1)
    broadcast_parameters(g2)
    g1.eval(); g2.eval()
    source = SliceSource('test', args.input_path, args.target_modality)
    lo, preds, gts = sample_slices(args, g1, g2, source, args.batch_size, device, rank, world,
                                   progress=lambda d, n: logging.info('rank %d: %d/%d slices', rank, d, n) if d % (args.batch_size * 8) == 0 else None)
    if world > 1:                                                # gather the shards on rank 0 (256 KB per slice)
        parts = [None] * world
        dist.gather_object((lo, preds, gts), parts if rank == 0 else None, dst=0)
        if rank == 0:
            parts.sort(key=lambda t: t[0])
            preds, gts = np.concatenate([p[1] for p in parts], 0), np.concatenate([p[2] for p in parts], 0)
    if rank == 0:
        res = export_and_score(list(preds), list(gts), None if args.no_png else os.path.join(args.output_path, 'generated_samples'))
        logging.info('Average PSNR: %.4f dB  SSIM: %.4f  MAE: %.6f over %d slices (global range [%.4f, %.4f])', res['psnr'], res['ssim'],
                     res['mae'], res['count'], res['global_min'], res['global_max'])
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
