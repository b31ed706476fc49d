"""Rows f1 (uncertainty map) and f3 (volume pipeline): CPU tests of the oracle restatement and of the build's host-side
logic against the fixtures tests/golden/volume.npz recorded from the reference's own functions
(engine/test_volume.py:135-181 executed by tests/golden/make_golden.py; F.interpolate / conv2d for the torch parts)."""
import gzip
import struct

import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import mudiff_oracle as O


def _np(gd, k):
    return gd[k].numpy() if isinstance(gd[k], torch.Tensor) else np.asarray(gd[k])


def _impls():
    from mudiff_hip import volume as V          # host logic only: importing it does not need the GPU
    return (('oracle', O), ('build', V))


@pytest.mark.parametrize('which', ['oracle', 'build'])
def test_robust_normalisation_matches_reference(which):
    gd = load_golden('volume.npz')
    M = dict(_impls())[which]
    for m in range(3):
        out = M.robust_minmax_to_minus1_1(_np(gd, f'vol{m}'))
        assert out.dtype == _np(gd, f'norm{m}').dtype and np.array_equal(out, _np(gd, f'norm{m}'))
        assert out.min() == -1.0 and out.max() == 1.0                      # outliers clipped at the percentiles
    out = M.robust_minmax_to_minus1_1(_np(gd, 'vol_nan'), mask=_np(gd, 'mask').astype(bool), pmin=5.0, pmax=90.0)
    assert np.array_equal(out, _np(gd, 'norm_masked'), equal_nan=True)
    for nm, v in (('zeros', np.zeros((4, 4, 3))), ('flat', np.full((4, 4, 3), 7.0))):      # degenerate volumes -> zeros
        out = M.robust_minmax_to_minus1_1(v)
        assert out.dtype == np.float32 and np.array_equal(out, _np(gd, f'norm_{nm}')) and not out.any()


@pytest.mark.parametrize('which', ['oracle', 'build'])
def test_slice_window_and_reassembly(which):
    gd = load_golden('volume.npz')
    M = dict(_impls())[which]
    rng = np.random.default_rng(1)
    for z, hr, s0, s1, n in _np(gd, 'slice_bounds').tolist():
        vol = rng.standard_normal((3, 2, z)).astype(np.float32)
        sl, a, b = M.extract_center_slices(vol, hr)
        assert (a, b, len(sl)) == (s0, s1, n)
        rec = M.reconstruct_volume_from_slices(sl, vol.shape, a, b)
        assert np.array_equal(rec[:, :, s0:s1 + 1], vol[:, :, s0:s1 + 1])
        assert not rec[:, :, :s0].any() and not rec[:, :, s1 + 1:].any()


def test_oracle_resize_and_uncertainty_map():
    gd = load_golden('volume.npz')
    for tag in ('down', 'up', 'x8', 'brats', 'same'):
        want = gd[f'resize.{tag}.out']
        assert torch.equal(O.resize_bilinear(gd[f'resize.{tag}.in'], want.shape[-2:]), want)
    assert torch.equal(O.uncertainty_map(gd['att.feat'], gd['att.w'], gd['att.b'], (64, 64)), gd['att.out'])


def test_nifti_round_trip_and_scaling(tmp_path):
    """The built-in NIfTI-1 reader/writer (nibabel is absent here): float32 round trip (.nii and .nii.gz) keeps data,
    shape and affine; an int16 big-endian file with scl_slope / scl_inter reads like nibabel's get_fdata()."""
    from mudiff_hip import volume as V
    rng = np.random.default_rng(3)
    vol = rng.standard_normal((6, 5, 4)).astype(np.float32)
    aff = np.array([[1.5, 0, 0, -10], [0, 2.0, 0, 5], [0, 0, 3.0, 7], [0, 0, 0, 1]], dtype=np.float64)
    for ext in ('nii', 'nii.gz'):
        p = str(tmp_path / f'v.{ext}')
        V.write_nifti(p, vol, aff)
        data, a2, hdr = V.read_nifti(p)
        assert data.dtype == np.float64 and data.shape == vol.shape and np.array_equal(data, vol.astype(np.float64))
        assert np.allclose(a2, aff)
        p2 = str(tmp_path / f'w.{ext}')                    # writing with the header we read keeps the geometry
        V.write_nifti(p2, vol * 2, a2, hdr)
        d2, a3, _ = V.read_nifti(p2)
        assert np.array_equal(d2, (vol * 2).astype(np.float64)) and np.allclose(a3, aff)
    raw = bytearray(348)
    struct.pack_into('>i', raw, 0, 348)
    struct.pack_into('>8h', raw, 40, 3, 3, 2, 2, 1, 1, 1, 1)
    struct.pack_into('>h', raw, 70, 4); struct.pack_into('>h', raw, 72, 16)
    struct.pack_into('>8f', raw, 76, 1, 1, 1, 1, 1, 1, 1, 1)
    struct.pack_into('>f', raw, 108, 352.0)
    struct.pack_into('>2f', raw, 112, 0.5, 10.0)
    raw[344:348] = b'n+1\0'
    ints = np.arange(12, dtype='>i2')
    p = str(tmp_path / 'be.nii.gz')
    with gzip.open(p, 'wb') as f:
        f.write(bytes(raw) + b'\0' * 4 + ints.tobytes())
    data, a, _ = V.read_nifti(p)
    assert np.array_equal(data, np.arange(12, dtype=np.float64).reshape((3, 2, 2), order='F') * 0.5 + 10.0)
    assert np.allclose(a, np.eye(4))
    with open(str(tmp_path / 'bad.nii'), 'wb') as f:
        f.write(b'\0' * 400)
    with pytest.raises(ValueError):
        V.read_nifti(str(tmp_path / 'bad.nii'))


def test_volume_parser_defaults_match_reference():
    """engine/test_volume.py:302-357 defaults (num_channels_dae=128!, slice_half_range=80, seed=1024 ...)."""
    from mudiff_hip import volume as V
    a = V.build_argparser(['--target_modality', 'T1CE', '--output_dir', 'o', '--exp', 'e'])
    assert (a.num_channels_dae, a.slice_half_range, a.image_size, a.seed, a.num_timesteps, a.nz) == (128, 80, 256, 1024, 4, 100)
    assert a.ch_mult == [1, 2, 4] and a.attn_resolutions == [16] and a.fir_kernel == [1, 3, 3, 1] and a.centered
    assert V.MODALITY_ORDERS['FLAIR'] == ['T1CE', 'T1', 'T2'] and V.MODALITY_ORDERS['T1'] == ['FLAIR', 'T1CE', 'T2']
    with pytest.raises(SystemExit):
        V.build_argparser(['--target_modality', 'PD', '--output_dir', 'o', '--exp', 'e'])
