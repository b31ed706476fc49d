"""Drop-in `backbones` package of the MI355X-native MU-Diff sampling path.

Same import paths, class names, constructor signatures, forward signatures and state_dict keys as
the reference's `backbones/` - the arithmetic underneath runs in
libmudiff_hip.so (hand-written gfx950 kernels).  GPU tensors only; no CPU fallback."""
