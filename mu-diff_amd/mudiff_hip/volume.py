"""Whole-volume prediction on MI355X.

Mirrors the reference's clinical entry point `engine/test_volume.py` - same functions, flags and outputs:
robust percentile normalisation of each input volume (:135-157), the centre +-half_range axial slices (:159-168),
bilinear resize to the model's image size (:273-274), 4-step dual-generator sampling, mapping to [0,1] (:281),
re-assembly into a volume of the original shape (:170-181) and a NIfTI written next to the inputs' geometry (:296-299).

What is different is the schedule, not the result: the reference pushes ONE slice at a time through the generators
(`for i in range(n)`, :266); here all <= 2*half_range+1 slices of the volume are uploaded once, resized by a HIP
kernel, and sampled in batches of `--batch_size` through one captured hipGraph per reverse step (sampling.GraphSampler).
Slices are independent, so the only observable difference is the order in which Gaussian draws are consumed; parity runs
inject the draws per slice (`predict_slices(..., x_inits, zs, noises)`).

NIfTI I/O: nibabel is used when importable (it is not in this image); otherwise a minimal built-in reader/writer
handles single-file NIfTI-1 (.nii / .nii.gz, little- or big-endian, scl_slope/inter applied like get_fdata()).
"""
from __future__ import annotations

import argparse
import gzip
import os
import struct

import numpy as np
import torch

MODALITY_ORDERS = {      # reference engine/test_volume.py:236-241 (same order as dataset/dataset_brats.py:29-34)
    'T1CE': ['FLAIR', 'T2', 'T1'],
    'FLAIR': ['T1CE', 'T1', 'T2'],
    'T2': ['T1CE', 'T1', 'FLAIR'],
    'T1': ['FLAIR', 'T1CE', 'T2'],
}


# ---------------------------------------------------------------------------------------------------
# host-side preprocessing (numpy in the reference too: one pass over a volume, not on the GPU hot path)
# ---------------------------------------------------------------------------------------------------
def robust_minmax_to_minus1_1(vol, mask=None, pmin=1.0, pmax=99.0):
    """Reference :135-157.  Intensities -> [-1,1] through the [pmin,pmax] percentiles of the non-zero voxels (or of
    `mask`, NaNs excluded); values outside are clipped.  No usable voxels or a flat volume -> zeros."""
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
    """Reference :159-168 -> (list of [X,Y] slices, first index, last index)."""
    z = volume.shape[2]
    c = z // 2
    s0, s1 = max(0, c - half_range), min(z - 1, c + half_range)
    return [volume[:, :, k] for k in range(s0, s1 + 1)], s0, s1


def reconstruct_volume_from_slices(predicted_slices, original_shape, start_slice, end_slice):
    """Reference :170-181: zeros everywhere except planes start_slice..end_slice."""
    vol = np.zeros(original_shape, dtype=np.float32)
    for i, sl in enumerate(predicted_slices):
        k = start_slice + i
        if k <= end_slice and k < original_shape[2]:
            vol[:, :, k] = np.asarray(sl, dtype=np.float32)
    return vol


# ---------------------------------------------------------------------------------------------------
# NIfTI-1 (single file) - used only when nibabel is absent
# ---------------------------------------------------------------------------------------------------
_NIFTI_DTYPES = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4', 1024: 'i8', 1280: 'u8'}


class NiftiHeader:
    """The 348 raw header bytes plus the fields this pipeline needs."""

    def __init__(self, raw, endian):
        self.raw, self.endian = bytes(raw), endian

    def _get(self, fmt, off):
        return struct.unpack_from(self.endian + fmt, self.raw, off)

    @property
    def shape(self):
        dim = self._get('8h', 40)
        return tuple(int(d) for d in dim[1:1 + dim[0]])

    @property
    def affine(self):
        """sform if set, else the pixdim scaling (qform rotations are not interpreted by this minimal reader)."""
        if self._get('h', 254)[0] > 0:
            rows = [self._get('4f', 280 + 16 * r) for r in range(3)]
            return np.array(rows + [(0., 0., 0., 1.)], dtype=np.float64)
        pix = self._get('8f', 76)
        return np.diag([pix[1], pix[2], pix[3], 1.0]).astype(np.float64)


def read_nifti(path):
    """-> (float64 array scaled like nibabel's get_fdata(), affine [4,4], header)."""
    try:
        import nibabel as nib                      # noqa: F401  (preferred when present)
        img = nib.load(path)
        return img.get_fdata(), img.affine, img.header
    except ImportError:
        pass
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 352:
        raise ValueError(f'{path}: too short for a NIfTI-1 file')
    endian = '<' if struct.unpack_from('<i', buf, 0)[0] == 348 else '>'
    if struct.unpack_from(endian + 'i', buf, 0)[0] != 348 or buf[344:347] != b'n+1':
        raise ValueError(f'{path}: not a single-file NIfTI-1 image (sizeof_hdr / magic mismatch)')
    hdr = NiftiHeader(buf[:348], endian)
    code = hdr._get('h', 70)[0]
    if code not in _NIFTI_DTYPES:
        raise ValueError(f'{path}: unsupported NIfTI datatype code {code}')
    offset = int(hdr._get('f', 108)[0])
    slope, inter = hdr._get('2f', 112)
    shape = hdr.shape
    n = int(np.prod(shape))
    data = np.frombuffer(buf, dtype=np.dtype(endian + _NIFTI_DTYPES[code]), count=n, offset=offset).reshape(shape, order='F')
    data = data.astype(np.float64)
    if slope not in (0.0,) and np.isfinite(slope) and (slope != 1.0 or inter != 0.0):
        data = data * slope + inter
    return data, hdr.affine, hdr


def write_nifti(path, vol, affine, header=None):
    """float32 volume + the inputs' geometry -> .nii / .nii.gz (reference :296-298 via nibabel)."""
    try:
        import nibabel as nib
        nib.save(nib.Nifti1Image(vol, affine, header), path)
        return
    except ImportError:
        pass
    vol = np.asarray(vol, dtype=np.float32)
    reuse = isinstance(header, NiftiHeader) and header.endian == '<'
    raw = bytearray(header.raw) if reuse else bytearray(348)
    struct.pack_into('<i', raw, 0, 348)
    dim = [vol.ndim] + list(vol.shape) + [1] * (7 - vol.ndim)
    struct.pack_into('<8h', raw, 40, *dim)
    struct.pack_into('<h', raw, 70, 16)           # datatype float32
    struct.pack_into('<h', raw, 72, 32)           # bitpix
    struct.pack_into('<f', raw, 108, 352.0)       # vox_offset
    struct.pack_into('<2f', raw, 112, 1.0, 0.0)   # scl_slope, scl_inter
    if not reuse:
        pix = [1.0] + [float(np.linalg.norm(np.asarray(affine)[:3, i])) for i in range(3)] + [1.0] * 4
        struct.pack_into('<8f', raw, 76, *pix)
    struct.pack_into('<h', raw, 254, 1)           # sform_code: scanner
    for r in range(3):
        struct.pack_into('<4f', raw, 280 + 16 * r, *[float(v) for v in np.asarray(affine)[r]])
    raw[344:348] = b'n+1\0'
    payload = bytes(raw) + b'\0\0\0\0' + vol.tobytes(order='F')
    opener = gzip.open if path.endswith('.gz') else open
    with opener(path, 'wb') as f:
        f.write(payload)


def load_and_preprocess_volume(file_path, slice_half_range):
    """Reference :183-191 -> (slices, shape, affine, header, first, last)."""
    vol, affine, header = read_nifti(file_path)
    slices, s0, s1 = extract_center_slices(robust_minmax_to_minus1_1(vol), slice_half_range)
    return slices, vol.shape, affine, header, s0, s1


def load_checkpoint(template, net, name, device):
    """Reference :193-203: strip 'module.' only where present, strict=False, eval()."""
    ckpt = torch.load(template.format(name), map_location=device, weights_only=True)
    if isinstance(ckpt, dict) and any(k.startswith('module.') for k in ckpt):
        ckpt = {(k[7:] if k.startswith('module.') else k): v for k, v in ckpt.items()}
    net.load_state_dict(ckpt, strict=False)
    net.eval()


# ---------------------------------------------------------------------------------------------------
# batched sampling of a stack of slices
# ---------------------------------------------------------------------------------------------------
def predict_slices(args, gen1, gen2, cond_stacks, device, batch_size=32, x_inits=None, zs=None, noises=None, seed=None,
                   use_graph=True, progress=None, sampler=None):
    """cond_stacks: three float arrays [n,X,Y] in [-1,1] (the condition contrasts, already normalised and sliced).
    -> [n,S,S] float32 numpy in [0,1], S = args.image_size.

    `sampler`: a sampling.GraphSampler built for these generators (any batch size, image_size x image_size) to reuse across
    volumes - warm-up and the two hipGraph captures are then paid once per process instead of once per volume.

    x_inits [n,1,S,S] / zs (per step [n,nz]) / noises (per step [n,1,S,S]) inject the Gaussian draws per slice for
    parity runs; otherwise they are drawn on the device (seeded by `seed` when given)."""
    from . import ops
    from . import sampling as S
    n = int(cond_stacks[0].shape[0])
    size = int(args.image_size)
    if n == 0:
        return np.zeros((0, size, size), np.float32)
    conds = []
    for st in cond_stacks:      # one upload + one resize launch per contrast (reference: per slice, on the CPU)
        t = torch.from_numpy(np.ascontiguousarray(st, dtype=np.float32)).to(device)[:, None]
        if tuple(t.shape[-2:]) != (size, size):
            t = ops.resize_bilinear(t, (size, size))
        conds.append(t.contiguous())
    coef = S.Posterior_Coefficients(args, device)
    gen = None
    if seed is not None:
        gen = torch.Generator(device=device).manual_seed(int(seed))
    T = int(args.num_timesteps)
    bs = min(int(batch_size), n)
    if sampler is not None:
        if not use_graph:
            raise ValueError('predict_slices: sampler= is a captured hipGraph; it cannot be combined with use_graph=False')
        if (sampler.H, sampler.W) != (size, size) or sampler.g1 is not gen1 or sampler.g2 is not gen2:
            raise ValueError('predict_slices: the sampler was built for other generators or another image size')
        bs = sampler.B                           # the captured batch size wins over batch_size (a short last batch is padded either way;
                                                 # seeded draws are made per slice below, so the result does not depend on it)
    elif use_graph:
        sampler = S.GraphSampler(coef, gen1, gen2, args, bs, size, size, device)
    out = torch.empty(n, size, size, device=device, dtype=torch.float32)

    def padded(t, lo, hi):      # the last batch is padded by repeating its last slice (fixed graph shape), trimmed afterwards
        t = t[lo:hi].to(device)
        return t if hi - lo == bs else torch.cat([t, t[-1:].expand(bs - (hi - lo), *t.shape[1:])], 0)

    if gen is not None and zs is None:
        # seeded run: every slice's draws come from its GLOBAL index (one pass over the generator for the whole volume), so the
        # same seed gives the same volume whatever the batch size or the reused sampler's captured shape (161 slices: 200 MB)
        if x_inits is None:
            x_inits = torch.randn(n, 1, size, size, device=device, generator=gen)
        zs = [torch.randn(n, args.nz, device=device, generator=gen) for _ in range(T)]
        noises = [torch.randn(n, 1, size, size, device=device, generator=gen) for _ in range(T)]
    for lo in range(0, n, bs):
        hi = min(lo + bs, n)
        c1, c2, c3 = (padded(c, lo, hi) for c in conds)
        x0 = padded(x_inits, lo, hi) if x_inits is not None else torch.randn(bs, 1, size, size, device=device)
        kw = {}
        if zs is not None:
            kw = dict(zs=[padded(z, lo, hi) for z in zs], noises=[padded(e, lo, hi) for e in noises])
        if sampler is not None:
            fake = sampler.sample(c1, c2, c3, x0, T, **kw)
        else:
            fake = S.sample_from_model(coef, gen1, c1, gen2, c2, c3, T, x0, None, args, **kw)
        out[lo:hi] = ops.to_range_0_1(fake)[:hi - lo, 0]
        if progress:
            progress(hi, n)
    return out.cpu().numpy()


def predict_volume(args):
    """Reference :209-300, same flags, same output file `predicted_<target>.nii.gz`."""
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    torch.manual_seed(args.seed)
    torch.cuda.set_device(args.gpu_chose)
    device = torch.device(f'cuda:{args.gpu_chose}')
    gen1, gen2 = NCSNpp(args).to(device), NCSNpp_adaptive(args).to(device)
    tmpl = os.path.join(args.output_path, args.exp, '{}.pth')
    load_checkpoint(tmpl, gen1, 'gen_diffusive_1', device)
    load_checkpoint(tmpl, gen2, 'gen_diffusive_2', device)

    if args.target_modality not in MODALITY_ORDERS:
        raise ValueError(f'Unsupported target modality: {args.target_modality}')
    needed = MODALITY_ORDERS[args.target_modality]
    provided = {'T1CE': args.input_t1ce, 'T1': args.input_t1, 'T2': args.input_t2, 'FLAIR': args.input_flair}
    for m in needed:
        if not provided.get(m):
            raise ValueError(f'Missing required input for {m}. Provide --input_{m.lower()}')
    stacks, ref = [], None
    for m in needed:
        slices, shp, aff, hdr, s0, s1 = load_and_preprocess_volume(provided[m], args.slice_half_range)
        if ref is None:
            ref = (shp, aff, hdr, s0, s1)
        elif shp != ref[0]:
            raise ValueError(f'All input volumes must share shape. Got {shp} vs {ref[0]} for {m}')
        stacks.append(np.stack(slices, 0))
    shp, aff, hdr, s0, s1 = ref
    if tuple(shp[:2]) != (args.image_size, args.image_size) and not args.resize_back:
        # the reference fails here too, later and less clearly (numpy broadcast error at :179 when the [S,S] prediction is
        # written into an [X,Y] plane); --resize_back is this build's opt-in extension
        raise ValueError(f'in-plane size {tuple(shp[:2])} differs from --image_size {args.image_size}: the prediction cannot be '
                         'written back into the volume (pass --resize_back to resample it bilinearly)')
    pred = predict_slices(args, gen1, gen2, stacks, device, batch_size=args.batch_size, seed=args.seed,
                          progress=lambda d, n: print(f'[infer] processed {d}/{n} slices'))
    if tuple(shp[:2]) != tuple(pred.shape[1:]):
        from . import ops
        pred = ops.resize_bilinear(torch.from_numpy(pred).to(device), shp[:2]).cpu().numpy()
    vol_pred = reconstruct_volume_from_slices(list(pred), shp, s0, s1)
    os.makedirs(args.output_dir, exist_ok=True)
    out_path = os.path.join(args.output_dir, f'predicted_{args.target_modality.lower()}.nii.gz')
    write_nifti(out_path, vol_pred, aff, hdr)
    print(f'[done] saved: {out_path} | shape={tuple(vol_pred.shape)} | slices={s0}..{s1}')
    return out_path


def build_argparser(argv=None):
    """Flags and defaults of the reference parser (:302-357; like it, returns the PARSED namespace), plus --centered (which
    the generators read and the reference parser forgot), --batch_size and --resize_back."""
    p = argparse.ArgumentParser('MU-Diff volume prediction (MI355X)')
    for m in ('t1ce', 't1', 't2', 'flair'):
        p.add_argument(f'--input_{m}', type=str, help=f'Path to {m.upper()} NIfTI')
    p.add_argument('--target_modality', type=str, required=True, choices=['T1CE', 'FLAIR', 'T2', 'T1'])
    p.add_argument('--output_dir', type=str, required=True)
    p.add_argument('--exp', type=str, required=True, help='Experiment directory name under --output_path')
    p.add_argument('--output_path', type=str, default='./results')
    p.add_argument('--slice_half_range', type=int, default=80)
    p.add_argument('--image_size', type=int, default=256)
    p.add_argument('--seed', type=int, default=1024)
    p.add_argument('--num_channels', type=int, default=1)
    p.add_argument('--num_channels_dae', type=int, default=128)
    p.add_argument('--n_mlp', type=int, default=3)
    p.add_argument('--ch_mult', nargs='+', type=int, default=[1, 2, 4])
    p.add_argument('--num_res_blocks', type=int, default=2)
    p.add_argument('--attn_resolutions', nargs='+', type=int, default=[16])
    p.add_argument('--dropout', type=float, default=0.0)
    p.add_argument('--resamp_with_conv', action='store_false', default=True)
    p.add_argument('--conditional', action='store_false', default=True)
    p.add_argument('--fir', action='store_false', default=True)
    p.add_argument('--fir_kernel', nargs='+', type=int, default=[1, 3, 3, 1])
    p.add_argument('--skip_rescale', action='store_false', default=True)
    p.add_argument('--resblock_type', type=str, default='biggan')
    p.add_argument('--progressive', type=str, default='none')
    p.add_argument('--progressive_input', type=str, default='residual')
    p.add_argument('--progressive_combine', type=str, default='sum')
    p.add_argument('--embedding_type', type=str, default='positional')
    p.add_argument('--fourier_scale', type=float, default=16.0)
    p.add_argument('--not_use_tanh', action='store_true', default=False)
    p.add_argument('--centered', action='store_false', default=True)
    p.add_argument('--nz', type=int, default=100)
    p.add_argument('--z_emb_dim', type=int, default=256)
    p.add_argument('--t_emb_dim', type=int, default=256)
    p.add_argument('--num_timesteps', type=int, default=4)
    p.add_argument('--use_geometric', action='store_true', default=False)
    p.add_argument('--beta_min', type=float, default=0.1)
    p.add_argument('--beta_max', type=float, default=20.0)
    p.add_argument('--use_bf16', action='store_true', default=False, help='accepted for compatibility; the MI355X path is fp32')
    p.add_argument('--gpu_chose', type=int, default=0)
    p.add_argument('--batch_size', type=int, default=32, help='slices per captured reverse step (MI355X build)')
    p.add_argument('--resize_back', action='store_true', help='resample the prediction to the in-plane size of the inputs')
    return p.parse_args(argv)


if __name__ == '__main__':
    predict_volume(build_argparser())
