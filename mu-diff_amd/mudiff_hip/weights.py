"""Reproducible random initialisation keyed by parameter NAME (no trained MU-Diff weights exist offline).

`seeded_state_dict(module, which, seed)` draws every tensor of `module.state_dict()` from its own CPU generator keyed by
(seed, which, name): filters / matrices U(+-sqrt(3 / fan_avg)) - including the tensors the reference starts at ~0 (Conv_1,
NIN_3, output conv: `init_scale=0.`, reference backbones/layers.py:58-89), which would make half the network multiply by
zero - biases 0.1*N(0,1), AdaGN gamma / GroupNorm gains 1 + 0.1*N(0,1).  Because the key is the state_dict name, the same
call on the reference's module, on the CPU oracle's parameter table and on the HIP modules yields identical weights: it is
how the committed fixtures under tests/golden/ were produced (tests/test_cabi_exports.py checks it against the table the
fixtures were made with).  Used by bench.py's parity leg and by callers that want deterministic stand-in weights.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import torch


def seeded_state_dict(module, which, seed=1234, fourier_scale=16.0):
    """which: 'g1' | 'g2' (part of the per-tensor seed).  -> OrderedDict name -> fp32 CPU tensor."""
    sd = OrderedDict()
    for name, ref in module.state_dict().items():
        shape = tuple(ref.shape)
        g = torch.Generator().manual_seed((zlib.crc32(f'{which}:{name}'.encode()) + 7919 * seed) % (2 ** 31))
        if len(shape) >= 2:
            rf = 1
            for s in shape[2:]:
                rf *= s
            bound = math.sqrt(3.0 / ((shape[0] + shape[1]) * rf / 2.0))
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        else:
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
            if name.endswith('.W'):                 # GaussianFourierProjection frequencies
                t = t * (10.0 * fourier_scale)
            elif name.endswith('style.bias'):
                t[: shape[0] // 2] += 1.0           # AdaGN gamma half
            elif name.endswith('.weight'):
                t += 1.0                            # GroupNorm gains
        sd[name] = t
    return sd
