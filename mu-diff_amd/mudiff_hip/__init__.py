"""ctypes binding of libmudiff_hip.so (include/mudiff_hip.h) for PyTorch-ROCm tensors.

torch is used for device memory and streams only: every function here takes tensors that already
live on the GPU, checks shapes on the host and enqueues hand-written gfx950 kernels on torch's
current stream.  There is NO CPU or PyTorch fallback: if the shared library is missing or a tensor is
not on a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the library binds to torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIPPED = os.path.join(_HERE, 'libmudiff_hip.so')
_LIB_PATH = os.environ.get('MUDIFF_HIP_LIB', _SHIPPED)   # override: kernel experiments; refused unless MUDIFF_ALLOW_VARIANT=1 (see load())
_lib = None
PREC_16X3, PREC_FP8X = 0, 1

ACT_NONE, ACT_SIGMOID, ACT_TANH, ACT_SILU, ACT_LRELU = 0, 1, 2, 3, 4
PRO_NONE, PRO_AFFINE, PRO_AFFINE_SILU, PRO_LRELU = 0, 1, 2, 3


class MudiffHipError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    """struct mud_conv_args (include/mudiff_hip.h)."""
    _fields_ = [
        ('x', C.c_void_p), ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('ldx', C.c_int),
        ('w', C.c_void_p), ('w_bstride', C.c_int64),
        ('ks', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
        ('pro_scale', C.c_void_p), ('pro_shift', C.c_void_p), ('pro_ld', C.c_int), ('pro_mode', C.c_int),
        ('bias', C.c_void_p),
        ('bias2', C.c_void_p), ('bias2_ld', C.c_int),
        ('res', C.c_void_p), ('ldr', C.c_int),
        ('out_scale', C.c_float), ('act', C.c_int),
        ('emul', C.c_void_p), ('ld_emul', C.c_int), ('egate', C.c_void_p), ('ld_egate', C.c_int), ('eother', C.c_void_p), ('ld_eother', C.c_int),
        ('out', C.c_void_p), ('Cout', C.c_int), ('ldo', C.c_int),
        ('sub2', C.c_int),
        ('stats', C.c_void_p), ('stats_ld', C.c_int),
        ('emul_cout', C.c_int),
        ('gn_sums', C.c_void_p), ('gn_sums_ld', C.c_int), ('gn_G', C.c_int), ('gn_eps', C.c_float), ('gn_count', C.c_double),
        ('gn_gamma', C.c_void_p), ('gn_beta', C.c_void_p), ('gn_bstride', C.c_int64),
        ('splitk_ws', C.c_void_p), ('splitk_ws_bytes', C.c_int64),
        ('splitk_counters', C.c_void_p), ('splitk_ncounters', C.c_int),
        ('skip_w', C.c_void_p), ('skip_bias', C.c_void_p), ('skip_out', C.c_void_p), ('skip_ldo', C.c_int),
        ('prec', C.c_int), ('w_exp', C.c_int),
    ]


MLP_MAX_LAYERS = 6


class MlpArgs(C.Structure):
    """struct mud_mlp_args (include/mudiff_hip.h)."""
    _fields_ = [
        ('x', C.c_void_p), ('ldx', C.c_int), ('B', C.c_int),
        ('nlayers', C.c_int), ('dims', C.c_int * (MLP_MAX_LAYERS + 1)),
        ('W', C.c_void_p * MLP_MAX_LAYERS), ('b', C.c_void_p * MLP_MAX_LAYERS),
        ('pixel_norm', C.c_int), ('act', C.c_int), ('act_last', C.c_int),
        ('out', C.c_void_p), ('ldo', C.c_int), ('maxdim', C.c_int),
    ]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    'mud_version': (C.c_int, []),
    'mud_last_error': (C.c_char_p, []),
    'mud_build_flags': (C.c_char_p, []),
    'mud_posterior_sample': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _L, _P]),
    'mud_q_sample': (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _L, _P]),
    'mud_timestep_embedding': (_I, [_P, _P, _I, _I, _F, _P]),
    'mud_pixel_norm': (_I, [_P, _P, _I, _I, _P]),
    'mud_dense': (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'mud_mlp_chain': (_I, [C.POINTER(MlpArgs), _P]),
    'mud_mlp_chains': (_I, [C.POINTER(MlpArgs), _I, _P]),
    'mud_gn_ws_bytes': (_L, [_I, _L, _I, _I]),
    'mud_gn_scale_shift': (_I, [_P, _I, _L, _I, _I, _I, _F, _P, _P, _L, _P, _P, _I, _P, _P, _P]),
    'mud_gn_scale_shift_from_sums': (_I, [_P, _I, _I, _I, _I, C.c_double, _F, _P, _P, _L, _P, _P, _I, _P]),
    'mud_channel_mean': (_I, [_P, _I, _L, _I, _I, _P, _I, _P, _P]),
    'mud_conv2d_direct': (_I, [C.POINTER(ConvArgs), _P]),
    'mud_packed_weight_bytes': (_L, [_I, _I, _I]),
    'mud_pack_weights': (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P]),
    'mud_pack_weights_prec': (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mud_conv2d_mfma': (_I, [C.POINTER(ConvArgs), _P]),
    'mud_conv2d_mfma_prec_supported': (_I, [C.POINTER(ConvArgs), _I]),
    'mud_conv2d_mfma_splitk_bytes': (_L, [C.POINTER(ConvArgs)]),
    'mud_upfirdn2d': (_I, [_P, _L, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mud_fir_nhwc': (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _I, _P]),
    'mud_minibatch_stddev': (_I, [_P, _I, _L, _I, _I, _I, _P, _P]),
    'mud_attention_supported': (_I, [_I]),
    'mud_attention_ws_bytes': (_L, [_I, _I, _I]),
    'mud_attention': (_I, [_P, _I, _I, _I, _I, _F, _P, _I, _P, _P]),
    'mud_softmax_rows': (_I, [_P, _L, _I, _I, _P]),
    'mud_mul': (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _P]),
    'mud_gate_mix': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _L, _I, _P, _I, _P]),
    'mud_fourier_embedding': (_I, [_P, _P, _P, _I, _I, _P]),
    'mud_resize_bilinear': (_I, [_P, _L, _I, _I, _I, _I, _P, _P]),
    'mud_affine_clamp': (_I, [_P, _L, _F, _F, _F, _F, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path():
    return _LIB_PATH


def load():
    """Load libmudiff_hip.so (once).  Fails loudly when it has not been built.  Only the in-tree shipped build is accepted:
    another path (MUDIFF_HIP_LIB) or a library that reports experiment flags (mud_build_flags() != "") needs
    MUDIFF_ALLOW_VARIANT=1 - a stray variant must never serve the product path silently."""
    global _lib
    if _lib is None:
        allow = os.environ.get('MUDIFF_ALLOW_VARIANT', '0') == '1'
        if os.path.realpath(_LIB_PATH) != os.path.realpath(_SHIPPED) and not allow:
            raise MudiffHipError(f'MUDIFF_HIP_LIB={_LIB_PATH} is not the shipped library ({_SHIPPED}); '
                                 'set MUDIFF_ALLOW_VARIANT=1 to run an experiment build')
        if not os.path.isfile(_LIB_PATH):
            raise MudiffHipError(
                f'{_LIB_PATH} not found: build it with `make -C mu-diff_amd/csrc` (or __graft_entry__.build()). '
                'There is no CPU / PyTorch fallback for the MU-Diff sampling path.')
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        flags = lib.mud_build_flags().decode()
        if flags and not allow:
            raise MudiffHipError(f'{_LIB_PATH} is an experiment build ({flags}); set MUDIFF_ALLOW_VARIANT=1 to run it')
        _lib = lib
    return _lib


def check(code, what=''):
    if code != 0:
        raise MudiffHipError(f'{what or "libmudiff_hip"} failed ({code}): {load().mud_last_error().decode()}')


def stream_ptr(device=None):
    """torch's current HIP stream on `device` (default: the current device) as a void*."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(*tensors):
    """All operands of one launch must live on ONE GPU (mudiff_hip.ops launches on that device's current stream whatever
    the process's current device is)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MudiffHipError('the MU-Diff HIP path needs tensors on an MI355X (got a CPU tensor); there is no CPU fallback')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise MudiffHipError(f'operands of one launch live on different GPUs ({dev} and {t.device})')


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
