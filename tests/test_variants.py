"""Row f4 (alternate configurations), CPU side: the generalised oracle against the reference's own outputs
(tests/golden/variants.npz), and the build's module layout (state_dict names, order, shapes) against the oracle's spec,
which tests/golden/make_golden.py checked against the reference's modules key by key."""
import pytest
import torch

from helpers import UNBUILDABLE, VARIANT_BASE, VARIANTS, load_golden
from oracle import mudiff_oracle as O

torch.set_grad_enabled(False)


def _classes(name):
    if name == 'healthy':
        from backbones import ncsnpp_generator_adagn_feat_healthy as H
        return H.NCSNpp, H.NCSNpp_adaptive, 2
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    return NCSNpp, NCSNpp_adaptive, 3


@pytest.mark.parametrize('name', list(VARIANTS))
def test_oracle_variant_matches_reference(name):
    gd = load_golden('variants.npz')
    cfg = O.default_config(**{**VARIANT_BASE, **VARIANTS[name]})
    nc = 2 if name == 'healthy' else 3
    x, c1, c2, t, z = (gd[f'{name}.{k}'] for k in ('x', 'c1', 'c2', 't', 'z'))
    c3 = gd[f'{name}.c3'] if nc == 3 else None
    sd1 = O.make_state_dict(cfg, 'g1', 77, n_cond=nc)
    y1 = O.g1_forward(sd1, cfg, x, c1, c2, c3, t, z)
    assert (y1 - gd[f'{name}.g1']).abs().max() <= 2e-5
    if f'{name}.g2' in gd:
        sd2 = O.make_state_dict(cfg, 'g2', 77, n_cond=nc)
        y2 = O.g2_forward(sd2, cfg, x, c1, c2, c3, t, z, gd[f'{name}.g1'][:, [0], :])
        assert (y2 - gd[f'{name}.g2']).abs().max() <= 2e-5


@pytest.mark.parametrize('name', list(VARIANTS))
def test_build_layout_matches_reference(name):
    G1, G2, nc = _classes(name)
    cfg = O.default_config(**{**VARIANT_BASE, **VARIANTS[name]})
    for which, cls in (('g1', G1), ('g2', G2)):
        m = cls(cfg)
        spec = O.param_spec(cfg, which, nc)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
        m.load_state_dict(O.make_state_dict(cfg, which, 77, n_cond=nc), strict=True)


@pytest.mark.parametrize('name', list(UNBUILDABLE))
def test_configurations_the_reference_cannot_run_are_refused(name):
    """The reference raises for these inside its own constructor / forward (recorded by make_golden.py); the build
    refuses them up front with that explanation, and the oracle asserts."""
    G1, G2, _ = _classes(name)
    cfg = O.default_config(**{**VARIANT_BASE, **UNBUILDABLE[name]})
    for cls in (G1, G2):
        with pytest.raises(NotImplementedError, match='reference'):
            cls(cfg)
    with pytest.raises(AssertionError):
        O.build_plan(cfg, 'g1')
    with pytest.raises(ValueError):
        G1(O.default_config(**{**VARIANT_BASE, 'resblock_type': 'nonsense'}))
