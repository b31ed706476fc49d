"""`utils.op` of the reference (utils/op/__init__.py:1-2): the up-FIR-down resampler, backed by the
gfx950 kernel in libmudiff_hip.so instead of a JIT-built CUDA extension (nothing is compiled at
import time).  `FusedLeakyReLU` / `fused_leaky_relu` are exported by the reference but never called
(SURVEY.md section 2a K3); they are out of scope and raise."""
from .upfirdn2d import upfirdn2d  # noqa: F401


def fused_leaky_relu(*args, **kwargs):
    raise NotImplementedError('fused_leaky_relu is dead code in the reference (no caller) and is not part of the MI355X build')


class FusedLeakyReLU:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('FusedLeakyReLU is dead code in the reference (no caller) and is not part of the MI355X build')
