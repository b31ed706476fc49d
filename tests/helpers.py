"""Shared test helpers (inputs for the BASELINE configs, golden loading)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def preprocess_demo(u8):
    """demo.ipynb cell 4 (reference): percentile-1/99 clip over non-zero pixels, min-max,
    (x-0.5)/0.5, rot90(k=-1)."""
    img = np.asarray(u8)
    low, high = np.percentile(img[img > 0], [1, 99])
    img = np.clip(img, low, high)
    img = (img - img.min()) / (img.max() - img.min())
    img = (img - 0.5) / 0.5
    t = torch.tensor(img, dtype=torch.float32)[None, None]
    return torch.rot90(t, k=-1, dims=(2, 3)).contiguous()


def demo_conds():
    g = load_golden('demo_inputs_u8.npz')
    return [preprocess_demo(g[n].numpy()) for n in ('flair', 't2', 't1')]


def sampler_inputs(cfg, B, seed_x=42):
    """x_init and the per-step (z, noise) stream exactly as tests/golden/make_golden.py drew them."""
    H = cfg.image_size
    g = torch.Generator().manual_seed(seed_x)
    x_init = torch.randn(B, 1, H, H, generator=g)
    st = torch.get_rng_state()
    torch.manual_seed(seed_x + 1)
    zs, noises = [], []
    for _ in range(cfg.num_timesteps):
        zs.append(torch.randn(B, cfg.nz))
        noises.append(torch.randn(B, 1, H, H))
    torch.set_rng_state(st)
    return x_init, zs, noises


CFG3_MODS = ['FLAIR', 'T2', 'T1', 'T1CE']        # channel order of wide_cfg3.npz's slices_u8 (tests/golden/make_golden.py)
CFG3_ORDERS = {      # reference dataset/dataset_brats.py:29-34: condition order per target contrast (target last)
    'T1CE': ['FLAIR', 'T2', 'T1', 'T1CE'],
    'FLAIR': ['T1CE', 'T1', 'T2', 'FLAIR'],
    'T2': ['T1CE', 'T1', 'FLAIR', 'T2'],
    'T1': ['FLAIR', 'T1CE', 'T2', 'T1'],
}


def wide_cfg3_case(cfg, copies=2):
    """BASELINE config 3 as SURVEY.md section 8(d) item 3 words it (tests/golden/wide_cfg3.npz): 16 distinct BraTS-shaped
    slices, 4 per target ordering, the reference's own B=4 run of each group with every step's x_new recorded.
    -> conds [3 x (16*copies,1,H,W)], x_init, zs, noises (the reference's draws, group by group), per-step reference x_new
    [16,1,H,W], targets [16,H,W] in [-1,1], group names; `copies` replicas of the 16 slices make one batch (2 -> 32)."""
    gd = load_golden('wide_cfg3.npz')
    sl = gd['slices_u8'].float() / 255.0 * 2.0 - 1.0                     # [16, 4, H, W]
    conds = [[], [], []]
    xs, zs, ns, tgt = [], [[] for _ in range(cfg.num_timesteps)], [[] for _ in range(cfg.num_timesteps)], []
    for gi, (name, order) in enumerate(CFG3_ORDERS.items()):
        idx = slice(4 * gi, 4 * gi + 4)
        for c in range(3):
            conds[c].append(sl[idx, CFG3_MODS.index(order[c])][:, None])
        tgt.append(sl[idx, CFG3_MODS.index(order[3])])
        x_init, z, n = sampler_inputs(cfg, 4, seed_x=314 + gi)
        xs.append(x_init)
        for k in range(cfg.num_timesteps):
            zs[k].append(z[k]); ns[k].append(n[k])
    rep = lambda parts: torch.cat(parts, 0).repeat(copies, *([1] * (parts[0].dim() - 1))).contiguous()   # noqa: E731
    refs = [torch.cat([gd[f'{name}.step{k}.xnew'] for name in CFG3_ORDERS], 0) for k in range(cfg.num_timesteps)]
    return dict(conds=[rep(c) for c in conds], x_init=rep(xs), zs=[rep(z) for z in zs], noises=[rep(n) for n in ns],
                refs=refs, targets=torch.cat(tgt, 0), groups=list(CFG3_ORDERS))


SMALL_CFGS = {
    's32': dict(image_size=32, num_channels_dae=32, ch_mult=[1, 2, 4], attn_resolutions=(16,)),
    's32na': dict(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(), num_res_blocks=1),
    's16t8': dict(image_size=16, num_channels_dae=16, ch_mult=[1, 1, 2], attn_resolutions=(4,),
                  num_timesteps=8, nz=50, z_emb_dim=64, n_mlp=2),
}


def small_conds(cfg, B=2):
    g = torch.Generator().manual_seed(101)
    return [torch.tanh(torch.randn(B, 1, cfg.image_size, cfg.image_size, generator=g)) for _ in range(3)]


# the alternate configurations recorded in tests/golden/variants.npz (same table as
# tests/golden/make_golden.py::VARIANTS - every configuration the reference itself can construct and run)
VARIANT_BASE = dict(image_size=32, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(16,), num_res_blocks=1)
VARIANTS = {
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
    'two_channels': dict(num_channels=2),
    'healthy': dict(),
}
UNBUILDABLE = {     # these raise inside the reference's own constructor / forward
    'resblock_ddpm': dict(resblock_type='ddpm'), 'resblock_oneadagn': dict(resblock_type='biggan_oneadagn'),
    'progressive_residual': dict(progressive='residual'), 'fir_false_output_skip': dict(fir=False, progressive='output_skip'),
}
