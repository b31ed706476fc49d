"""upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)) - the reference's Python signature
(utils/op/upfirdn2d.py:170-181) over mud_upfirdn2d (include/mudiff_hip.h).  Inference only: the
reference's autograd Function (backward / double backward) belongs to training, out of scope."""
from collections import abc

from mudiff_hip import ops


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return ops.upfirdn2d_planes(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


def upfirdn2d_ada(input, kernel, up=1, down=1, pad=(0, 0)):
    """Per-axis up/down/pad variant (utils/op/upfirdn2d.py:183-199 of the reference)."""
    up = up if isinstance(up, abc.Iterable) else (up, up)
    down = down if isinstance(down, abc.Iterable) else (down, down)
    pad = (pad[0], pad[1], pad[0], pad[1]) if len(pad) == 2 else pad
    return ops.upfirdn2d_planes(input, kernel, tuple(up), tuple(down), tuple(pad))
