"""CPU: libmudiff_hip.so loads and exports every symbol include/mudiff_hip.h declares (no compute)."""
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'mudiff_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mud_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_are_exported_and_bound():
    import mudiff_hip
    if not os.path.isfile(mudiff_hip.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = mudiff_hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/mudiff_hip.h but not exported'
    assert sorted(mudiff_hip.EXPORTED_SYMBOLS) == declared, 'ctypes binding and header disagree'
    assert lib.mud_version() >= 110
    assert lib.mud_build_flags() == b'', 'the in-tree library must be the clean build (no experiment flags)'
    assert lib.mud_packed_weight_bytes(3, 64, 64) == 1 * 4 * 9 * 4096 + 8192   # tiles * k16 chunks * taps * (hi+lo planes) + DMA slack
    assert lib.mud_packed_weight_bytes(2, 64, 64) == -1
    assert lib.mud_gn_ws_bytes(1, 65536, 256, 32) > 0


def test_missing_library_fails_loudly(monkeypatch):
    import mudiff_hip
    monkeypatch.setattr(mudiff_hip, '_lib', None)
    monkeypatch.setattr(mudiff_hip, '_LIB_PATH', '/nonexistent/libmudiff_hip.so')
    monkeypatch.setattr(mudiff_hip, '_SHIPPED', '/nonexistent/libmudiff_hip.so')
    with pytest.raises(mudiff_hip.MudiffHipError, match='no CPU'):
        mudiff_hip.load()


def test_only_the_shipped_library_is_loaded_unless_a_variant_is_allowed(monkeypatch, tmp_path):
    """a stray MUDIFF_HIP_LIB (an experiment build) must not serve the product path silently."""
    import shutil
    import mudiff_hip
    other = tmp_path / 'lib_variant.so'
    shutil.copy(mudiff_hip._SHIPPED, other)
    monkeypatch.setattr(mudiff_hip, '_lib', None)
    monkeypatch.setattr(mudiff_hip, '_LIB_PATH', str(other))
    monkeypatch.delenv('MUDIFF_ALLOW_VARIANT', raising=False)
    with pytest.raises(mudiff_hip.MudiffHipError, match='MUDIFF_ALLOW_VARIANT'):
        mudiff_hip.load()
    monkeypatch.setenv('MUDIFF_ALLOW_VARIANT', '1')
    assert mudiff_hip.load().mud_version() >= 110
    monkeypatch.setattr(mudiff_hip, '_lib', None)      # (do not leave the copy bound for the tests that follow)


def test_drop_in_modules_have_reference_state_dict():
    from oracle import mudiff_oracle as O
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    for kw in (dict(), dict(image_size=32, num_channels_dae=32, attn_resolutions=(16,)),
               dict(image_size=16, num_channels_dae=16, ch_mult=[1, 1, 2], attn_resolutions=(4,), nz=50, z_emb_dim=64, n_mlp=2)):
        cfg = O.default_config(**kw)
        for cls, which in ((NCSNpp, 'g1'), (NCSNpp_adaptive, 'g2')):
            sd = cls(cfg).state_dict()
            spec = O.param_spec(cfg, which)
            assert list(sd) == list(spec)
            assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)


def test_host_tables_match_oracle_bitwise():
    import torch
    from oracle import mudiff_oracle as O
    from mudiff_hip import sampling as S
    for kw in (dict(num_timesteps=4), dict(num_timesteps=8), dict(num_timesteps=4, use_geometric=True, beta_min=0.01, beta_max=0.9)):
        cfg = O.default_config(**kw)
        p, op = S.Posterior_Coefficients(cfg, 'cpu'), O.PosteriorCoefficients(cfg)
        for f in ('betas', 'alphas_cumprod', 'posterior_variance', 'posterior_mean_coef1', 'posterior_mean_coef2', 'posterior_log_variance_clipped'):
            assert torch.equal(getattr(p, f), getattr(op, f)), f
        d, od = S.Diffusion_Coefficients(cfg, 'cpu'), O.DiffusionCoefficients(cfg)
        for f in ('sigmas', 'a_s', 'a_s_cum', 'sigmas_cum', 'a_s_prev'):
            assert torch.equal(getattr(d, f), getattr(od, f)), f
        assert torch.equal(S.get_time_schedule(cfg, 'cpu'), O.get_time_schedule(cfg))


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The by-value argument structs of the C ABI are mirrored by hand in ctypes (mudiff_hip.ConvArgs / MlpArgs): compile the
    header with gcc and compare size and every field offset, so a field added on one side only cannot go unnoticed."""
    import ctypes as C
    import subprocess
    import mudiff_hip
    structs = {'mud_conv_args': mudiff_hip.ConvArgs, 'mud_mlp_args': mudiff_hip.MlpArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(REPO, "include", "mudiff_hip.h")}"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c11', '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout
    seen = 0
    for ln in out.strip().splitlines():
        cname, field, val = ln.split()
        cls = structs[cname]
        want = C.sizeof(cls) if field == 'size' else getattr(cls, field).offset
        assert int(val) == want, f'{cname}.{field}: header {val}, ctypes {want}'
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
