"""Tensor-level wrappers over the C ABI: NHWC views, shape checks on the host, launches on torch's
current stream.  Everything here is GPU-only (see mudiff_hip.__init__)."""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import (ACT_LRELU, ACT_NONE, ACT_SIGMOID, ACT_SILU, ACT_TANH, PREC_16X3, PREC_FP8X, PRO_AFFINE, PRO_AFFINE_SILU, PRO_LRELU, PRO_NONE,  # noqa: F401
               ConvArgs, MudiffHipError, check, load, ptr, require_gpu)


# MUD_DETERMINISTIC=1: bit-stable outputs run to run.  The only order-dependent arithmetic of the path is the fp64 atomic
# accumulation of the GroupNorm (sum, sumsq) in the producers' epilogues (~1e-6 jitter on the outputs); with this switch the
# producers accumulate nothing and every GroupNorm re-reads its input with per-workgroup partials + a fixed-order finalize
# (mud_gn_scale_shift).  The reference's CPU path is deterministic; this costs one extra pass over each normalised tensor.
DETERMINISTIC = os.environ.get('MUD_DETERMINISTIC', '0') == '1'


class StatsArena:
    """Zeroed fp64 scratch for the per-(sample, channel) (sum, sumsq) accumulators that producers fill in
    their epilogues for the next GroupNorm.  One memset per chunk instead of one per tensor."""

    def __init__(self, device, chunk_doubles=1 << 18):
        self.device, self.chunk = device, chunk_doubles
        self.buf, self.used = None, 0

    def take(self, B, C):
        if DETERMINISTIC:         # no producer-side statistics: every GroupNorm takes the fixed-order two-pass reduction
            return None
        n = B * C * 2
        if self.buf is None or self.used + n > self.buf.numel():
            self.buf = torch.zeros(max(self.chunk, n), device=self.device, dtype=torch.float64)
            self.used = 0
        t = self.buf[self.used:self.used + n].view(B, C, 2)
        self.used += n
        return t


class View:
    """NHWC fp32 view (ptr, B, H, W, C, ld) into a torch tensor that owns the memory.  `stats`, when
    present, is a [B, ld, 2] fp64 tensor aligned with the buffer's channel axis that producers of this
    view accumulate per-channel (sum, sum of squares) into (see mud_conv_args.stats)."""
    __slots__ = ('base', 'B', 'H', 'W', 'C', 'ld', 'c0', 'stats')

    def __init__(self, base, B, H, W, C, ld=None, c0=0, stats=None):
        self.base, self.B, self.H, self.W, self.C = base, B, H, W, C
        self.ld = C if ld is None else ld
        self.c0 = c0
        self.stats = stats

    @staticmethod
    def empty(B, H, W, C, device, arena=None):
        v = View(torch.empty(B, H, W, C, device=device, dtype=torch.float32), B, H, W, C)
        if arena is not None:
            v.stats = arena.take(B, C)
        return v

    @property
    def stats_ptr(self):
        return None if self.stats is None else C.c_void_p(self.stats.data_ptr() + 16 * self.c0)

    @staticmethod
    def from_nchw(x):
        """NCHW torch tensor -> NHWC view (zero-copy when C == 1)."""
        require_gpu(x)
        B, Cc, H, W = x.shape
        x = x.float()
        t = x.reshape(B, H, W, 1) if Cc == 1 else x.permute(0, 2, 3, 1)
        return View(t.contiguous(), B, H, W, Cc)

    def to_nchw(self):
        t = self.tensor()
        if self.C == 1:
            return t.reshape(self.B, 1, self.H, self.W)
        return t.permute(0, 3, 1, 2).contiguous()

    def tensor(self):
        """The viewed region as a (possibly non-contiguous) torch tensor [B,H,W,C]."""
        return self.base.reshape(self.B, self.H, self.W, self.ld)[..., self.c0:self.c0 + self.C]

    def slice(self, c0, C_):
        assert 0 <= c0 and c0 + C_ <= self.C
        return View(self.base, self.B, self.H, self.W, C_, self.ld, self.c0 + c0, self.stats)

    @property
    def ptr(self):
        return C.c_void_p(self.base.data_ptr() + 4 * self.c0)

    @property
    def npix(self):
        return self.B * self.H * self.W

    @property
    def device(self):
        return self.base.device


class _Profile:
    """Optional per-launch timing with HIP events on the launching stream (bench.py's roofline leg)."""

    def __init__(self):
        self.on = False
        self.records = []

    def enable(self):
        self.on, self.records = True, []

    def disable(self):
        self.on = False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, flops, nbytes in self.records:
            d = out.setdefault(name, dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
            d['n'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += flops
            d['bytes'] += nbytes
        return out


PROFILE = _Profile()


STREAM = object()      # placeholder argument: replaced by the launch device's current HIP stream


def _launch(name, dev, fn, *args, flops=0.0, nbytes=0.0):
    """Enqueue one C-ABI call on the current stream of `dev` - the device the operands live on, which need not be the
    process's current device (the kernels take raw pointers: a launch on another device would fault or silently run on
    the wrong GPU)."""
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    if idx != torch.cuda.current_device():
        with torch.cuda.device(idx):
            return _launch(name, dev, fn, *args, flops=flops, nbytes=nbytes)
    args = tuple(C.c_void_p(torch.cuda.current_stream().cuda_stream) if a_ is STREAM else a_ for a_ in args)
    if PROFILE.on:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        code = fn(*args)
        e1.record()
        PROFILE.records.append((name, e0, e1, flops, nbytes))
    else:
        code = fn(*args)
    check(code, name)


def _f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


# ---------------------------------------------------------------------------------------------------
def posterior_sample(x01, x02, xt, noise, t, coef1, coef2, std_tab, out=None):
    require_gpu(x01, xt, noise, t, coef1)
    B = xt.shape[0]
    per = xt[0].numel() if B else 0
    x01, xt, noise = _f32(x01.contiguous()), _f32(xt.contiguous()), _f32(noise.contiguous())
    if x02 is not None:
        x02 = _f32(x02.contiguous())
        assert x02.shape == xt.shape
    assert x01.shape == xt.shape == noise.shape and t.dtype == torch.int64 and t.numel() == B
    out = torch.empty_like(xt) if out is None else out
    _launch('posterior_sample', xt.device, load().mud_posterior_sample, ptr(x01), ptr(x02), ptr(xt), ptr(noise), ptr(t.contiguous()), ptr(coef1), ptr(coef2),
                                      ptr(std_tab), coef1.numel(), ptr(out), B, per, STREAM)
    return out


def q_sample(x, noise, t, toff, a_tab, s_tab):
    require_gpu(x, noise, t, a_tab, s_tab)
    x, noise = _f32(x.contiguous()), _f32(noise.contiguous())
    B = x.shape[0]
    out = torch.empty_like(x)
    _launch('q_sample', x.device, load().mud_q_sample, ptr(x), ptr(noise), ptr(t.contiguous()), toff, ptr(a_tab), ptr(s_tab), a_tab.numel(), ptr(out), B,
                              x[0].numel() if B else 0, STREAM)
    return out


def timestep_embedding(t, dim, max_positions=10000.0):
    require_gpu(t)
    assert t.dim() == 1 and t.dtype == torch.int64
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float32)
    _launch('timestep_embedding', t.device, load().mud_timestep_embedding, ptr(t.contiguous()), ptr(out), t.shape[0], dim, float(max_positions), STREAM)
    return out


def fourier_embedding(t, W):
    """[sin | cos](2*pi*W*log(t)) -> [B, 2*len(W)] (GaussianFourierProjection of log(time_cond))."""
    require_gpu(t, W)
    tf, Wf = _f32(t.float().contiguous()), _f32(W.detach().float().contiguous())
    out = torch.empty(tf.shape[0], 2 * Wf.shape[0], device=t.device, dtype=torch.float32)
    _launch('fourier_embedding', t.device, load().mud_fourier_embedding, ptr(tf), ptr(Wf), ptr(out), tf.shape[0], Wf.shape[0], STREAM)
    return out


def pixel_norm(z):
    require_gpu(z)
    z = _f32(z.contiguous())
    out = torch.empty_like(z)
    _launch('pixel_norm', z.device, load().mud_pixel_norm, ptr(z), ptr(out), z.shape[0], z.shape[1], STREAM)
    return out


def dense(x, W, bias, act_in=ACT_NONE, act_out=ACT_NONE):
    """x [B,K] (row stride allowed), W [N,K], bias [N] -> [B,N]."""
    require_gpu(x, W)
    assert x.dim() == 2 and x.stride(1) == 1 and W.is_contiguous() and x.shape[1] == W.shape[1]
    B, K = x.shape
    N = W.shape[0]
    out = torch.empty(B, N, device=x.device, dtype=torch.float32)
    _launch('dense', x.device, load().mud_dense, ptr(x), x.stride(0) if B > 1 else K, ptr(W), ptr(bias), ptr(out), N, B, K, N, act_in, act_out,
                           STREAM)
    return out


def _mlp_args(a, x, layers, pixel_norm, act, act_last):
    from . import MLP_MAX_LAYERS
    require_gpu(x, *[w for w, _ in layers])
    assert x.dim() == 2 and x.stride(1) == 1 and 1 <= len(layers) <= MLP_MAX_LAYERS
    xin = x if x.dtype == torch.float32 else x.float()
    a.x, a.ldx, a.B, a.nlayers = ptr(xin), (xin.stride(0) if xin.shape[0] > 1 else xin.shape[1]), xin.shape[0], len(layers)
    a.dims[0] = xin.shape[1]
    keep = [xin]
    for l, (w, b) in enumerate(layers):
        w = _f32(w.detach().contiguous())
        assert w.shape[1] == a.dims[l], (tuple(w.shape), a.dims[l])
        a.dims[l + 1] = w.shape[0]
        a.W[l] = w.data_ptr()
        if b is not None:
            b = _f32(b.detach().contiguous())
            a.b[l] = b.data_ptr()
        keep += [w, b]
    a.pixel_norm, a.act, a.act_last = int(pixel_norm), act, int(act_last)
    out = torch.empty(xin.shape[0], a.dims[len(layers)], device=x.device, dtype=torch.float32)
    a.out, a.ldo = ptr(out), out.shape[1]
    return out, keep


def mlp_chain(x, layers, *, pixel_norm=False, act=ACT_SILU, act_last=False):
    """x [B,K0] -> [B,N_last] through `layers` = [(W [N,K], bias [N] or None), ...] in ONE launch (activation between the
    layers, after the last one iff act_last; optional PixelNorm of x first)."""
    return mlp_chains([dict(x=x, layers=layers, pixel_norm=pixel_norm, act=act, act_last=act_last)])[0]


def mlp_chains(chains):
    """Up to four independent chains (dicts of mlp_chain's arguments) side by side in ONE launch -> list of outputs."""
    from . import MlpArgs
    assert 1 <= len(chains) <= 4
    arr = (MlpArgs * len(chains))()
    outs, keep = [], []
    for a, c in zip(arr, chains):
        o, k = _mlp_args(a, c['x'], c['layers'], c.get('pixel_norm', False), c.get('act', ACT_SILU), c.get('act_last', False))
        outs.append(o)
        keep.append(k)
    require_gpu(*[c['x'] for c in chains])
    _launch('mlp_chain', chains[0]['x'].device, load().mud_mlp_chains, arr, len(chains), STREAM)
    return outs


_WS = {}


def _workspace(device, nbytes):
    ws = _WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _WS[device] = ws
    return ws


def gn_scale_shift(x: View, G, gamma=None, beta=None, eps=1e-6):
    """-> (scale [B,C], shift [B,C]).  gamma/beta: None, [C] or [B,C] (row-strided views allowed)."""
    lib = load()
    HW = x.H * x.W
    ss = torch.empty(2, x.B, x.C, device=x.device, dtype=torch.float32)
    bstride = 0
    if gamma is not None:
        assert gamma.stride(-1) == 1 and beta.stride(-1) == 1 and gamma.shape[-1] == x.C
        if gamma.dim() == 2:
            assert gamma.shape[0] == x.B and gamma.stride(0) == beta.stride(0)
            bstride = gamma.stride(0)
    if x.stats is not None:      # the producers already accumulated (sum, sumsq): no pass over the tensor
        _launch('gn_from_sums', x.device, lib.mud_gn_scale_shift_from_sums, x.stats_ptr, x.stats.shape[1], x.B, x.C, G, float(HW), eps, ptr(gamma),
                ptr(beta), bstride, ptr(ss[0]), ptr(ss[1]), x.C, STREAM)
        return ss[0], ss[1]
    ws = _workspace(x.device, lib.mud_gn_ws_bytes(x.B, HW, x.C, G))
    _launch('gn_scale_shift', x.device, lib.mud_gn_scale_shift, x.ptr, x.B, HW, x.C, x.ld, G, eps, ptr(gamma), ptr(beta), bstride, ptr(ss[0]),
            ptr(ss[1]), x.C, None, ptr(ws), STREAM, nbytes=4.0 * x.npix * x.C)
    return ss[0], ss[1]


class LazyGN:
    """GroupNorm scale / shift of a view whose producers already accumulated the per-channel (sum, sumsq): nothing is
    launched for it.  A consumer that can finalise it in its own prologue (mud_conv2d_mfma: mud_conv_args.gn_*) takes it as
    `pro=(lazy, None, mode)`; any other consumer calls `.tensors()` (one small launch, cached)."""
    __slots__ = ('x', 'G', 'gamma', 'beta', 'bstride', 'eps', '_ss')

    def __init__(self, x: View, G, gamma, beta, bstride, eps):
        self.x, self.G, self.gamma, self.beta, self.bstride, self.eps, self._ss = x, G, gamma, beta, bstride, eps, None

    def tensors(self):
        if self._ss is None:
            x = self.x
            ss = torch.empty(2, x.B, x.C, device=x.device, dtype=torch.float32)
            _launch('gn_from_sums', x.device, load().mud_gn_scale_shift_from_sums, x.stats_ptr, x.stats.shape[1], x.B, x.C, self.G,
                    float(x.H * x.W), self.eps, ptr(self.gamma), ptr(self.beta), self.bstride, ptr(ss[0]), ptr(ss[1]), x.C, STREAM)
            self._ss = (ss[0], ss[1])
        return self._ss

    def __iter__(self):          # `sc, sh = ...` keeps working for callers that need the arrays
        return iter(self.tensors())


def gn_lazy(x: View, G, gamma=None, beta=None, eps=1e-6):
    """gn_scale_shift whose finalisation is deferred to the consumer when the producers left (sum, sumsq) behind
    (-> LazyGN); otherwise the two-pass statistics run now (-> (scale, shift))."""
    if x.stats is None or not FOLD_GN:
        return gn_scale_shift(x, G, gamma, beta, eps)
    bstride = 0
    if gamma is not None:
        assert gamma.stride(-1) == 1 and beta.stride(-1) == 1 and gamma.shape[-1] == x.C
        if gamma.dim() == 2:
            assert gamma.shape[0] == x.B and gamma.stride(0) == beta.stride(0)
            bstride = gamma.stride(0)
    return LazyGN(x, G, gamma, beta, bstride, eps)


FOLD_GN = os.environ.get('MUD_FOLD_GN', '1') != '0'      # A/B knob: 0 = one gn_from_sums launch per GroupNorm (round-1 behaviour)


FUSE_SKIP = os.environ.get('MUD_FUSE_SKIP', '1') != '0'     # A/B knob: 0 = the 1x1 skip conv stays its own launch (round-1 behaviour)


_SPLITK_COUNTERS = {}
_SPLITK_OWNED = []        # innermost `own_splitk_counters` buffer, if any


def new_splitk_counters(device):
    return torch.zeros(4096, device=device, dtype=torch.int32)


class own_splitk_counters:
    """Context: the split-K launches issued inside use THIS counter array.  A captured hipGraph can be replayed on any stream, and
    two graphs replayed concurrently (two samplers on two streams) must not share arrival counters - a tile would be reduced early
    or never - so every GraphSampler captures with an array of its own instead of the per-(device, stream) one below."""

    def __init__(self, buf):
        self.buf = buf

    def __enter__(self):
        _SPLITK_OWNED.append(self.buf)
        return self.buf

    def __exit__(self, *exc):
        _SPLITK_OWNED.pop()
        return False


def splitk_counters(device):
    """Arrival counters of the in-launch split-K reduction (mud_conv_args.splitk_counters): the array of the enclosing
    `own_splitk_counters` context (graph capture), else one zeroed array per (device, stream) - launches on one stream are ordered,
    and every launch leaves the counters at zero."""
    if _SPLITK_OWNED and _SPLITK_OWNED[-1].device == torch.device(device):
        return _SPLITK_OWNED[-1]
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _SPLITK_COUNTERS.get(key)
    if buf is None:
        buf = _SPLITK_COUNTERS[key] = new_splitk_counters(device)
    return buf


def conv3x3_would_split_k(x: View, cout):
    """Would mud_conv2d_mfma deal the K chunks of this 3x3 launch to several workgroups (small grid, long reduction)?"""
    a = ConvArgs()
    a.x, a.B, a.H, a.W, a.Cin, a.ldx, a.ks, a.stride, a.pad = x.ptr, x.B, x.H, x.W, x.C, x.ld, 3, 1, 1
    a.out, a.Cout, a.ldo = x.ptr, cout, (cout + 3) & ~3          # (only sizes and alignment are looked at)
    return load().mud_conv2d_mfma_splitk_bytes(C.byref(a)) > 0


FUSE_SKIP_SPLIT = os.environ.get('MUD_FUSE_SKIP_SPLIT', '1') != '0'       # A/B knob: keep the skip conv fused where the launch is split over K


def fused_skip_ok(x: View, cout, pro_mode):
    """Should mud_conv2d_mfma produce the block's 1x1 skip conv alongside its 3x3 conv (mud_conv_args.skip_*)?  Where the launch
    is split over K (one slice at a time, 64x64 maps) the fused kernel splits too - both accumulator sets go through the slabs and
    the last workgroup of a tile reduces them (an unsplit fused launch there cost 109 us against 59 + 23 us: 512->256)."""
    return (FUSE_SKIP and pro_mode == PRO_AFFINE_SILU and x.C % 4 == 0 and 8 <= x.C <= 512 and cout % 4 == 0
            and (FUSE_SKIP_SPLIT or not conv3x3_would_split_k(x, cout)))


def resolve_pro(pro):
    """(scale, shift, mode) with the arrays materialised (for consumers that cannot fold the GroupNorm finalisation)."""
    if pro is not None and isinstance(pro[0], LazyGN):
        sc, sh = pro[0].tensors()
        return (sc, sh, pro[2])
    return pro


def channel_mean(x: View):
    lib = load()
    HW = x.H * x.W
    ws = _workspace(x.device, lib.mud_gn_ws_bytes(x.B, HW, x.C, x.C))
    out = torch.empty(x.B, x.C, device=x.device, dtype=torch.float32)
    _launch('channel_mean', x.device, lib.mud_channel_mean, x.ptr, x.B, HW, x.C, x.ld, ptr(out), x.C, ptr(ws), STREAM, nbytes=4.0 * x.npix * x.C)
    return out


# ---------------------------------------------------------------------------------------------------
def pack_weights(src, s_tap, s_ci, s_co, ks, Cin, Cout, nbatch=1, src_bstride=0, src_offset=0, prec=PREC_16X3, w_exp=0):
    """-> uint8 tensor [nbatch, packed bytes] in the MFMA kernel's B-operand layout (for the arithmetic plan `prec`)."""
    lib = load()
    require_gpu(src)
    nbytes = lib.mud_packed_weight_bytes(ks, Cin, Cout)
    dst = torch.empty(nbatch, nbytes, device=src.device, dtype=torch.uint8)
    _launch('pack_weights', src.device, lib.mud_pack_weights_prec, C.c_void_p(src.data_ptr() + 4 * src_offset), s_tap, s_ci, s_co, src_bstride, ks, Cin, Cout,
            nbatch, prec, w_exp, ptr(dst), STREAM)
    return dst


def fp8x_weight_exponent(w):
    """The power-of-two pre-scale of a layer's e4m3 weight image (MUD_PREC_FP8X): the largest e with max|w| * 2^e <= 448 (e4m3's
    largest finite value), so that the 4 significant bits sit where this layer's weights are.  Host sync: pack time only."""
    m = float(w.detach().abs().max())
    if not (m > 0.0) or not math.isfinite(m):
        return 0
    return max(-100, min(100, int(math.floor(math.log2(448.0 / m)))))


def pack_conv_weight(w_oihw, prec=PREC_16X3, w_exp=0):
    """nn.Conv2d weight [O,I,k,k] -> packed MFMA operand."""
    O, I, k, _ = w_oihw.shape
    w = _f32(w_oihw.detach().contiguous())
    return pack_weights(w, 1, k * k, I * k * k, k, I, O, prec=prec, w_exp=w_exp)


def pack_matrix_in_out(W_in_out):
    """NIN weight W[in,out] -> packed 1x1 operand."""
    I, O = W_in_out.shape
    return pack_weights(_f32(W_in_out.detach().contiguous()), 0, O, 1, 1, I, O)


def direct_weight(w_oihw):
    """[O,I,k,k] -> fp32 [k,k,I,O] for mud_conv2d_direct."""
    return w_oihw.detach().permute(2, 3, 1, 0).contiguous()


# ---- arithmetic plan per 3x3 launch (mud_conv_args.prec).  MUD_PREC_PLAN: 'auto' (default) = the fp16 + e4m3-cross-term plan
# (MUD_PREC_FP8X) for every launch the library has it for (fp8x_pays), fp16 x 3 elsewhere (small grids, 1x1, the exact head / tail
# kernels); 'off' = every launch fp16 x 3; 'all' = wherever the library has the plan, whatever fp8x_pays says.
PREC_PLAN = os.environ.get('MUD_PREC_PLAN', 'auto')


def fp8x_pays(B, H, W, cin, cout):
    """Should a launch the library has MUD_PREC_FP8X for take it?  Per launch in isolation (scripts/ab_prec.py, batch 16,
    profiles/r03_g_ab_prec_b16.txt) the plan is 0.86-0.99x of the fp16 x 3 launch on every shape it is built for (the 64 -> 64
    layers at 256x256 included, on their one-row tile), and the whole bench line alternated on one box agrees
    (profiles/r03_d_plan_alternation.txt): no shape is excluded.  The hook stays for shapes a later measurement finds to lose."""
    return True


def conv_prec_supported(x: View, cout, pro_mode, prec, skip=False, sub2=False):
    a = ConvArgs()
    a.x, a.B, a.H, a.W, a.Cin, a.ldx, a.ks, a.stride, a.pad = x.ptr, x.B, x.H, x.W, x.C, x.ld, 3, 1, 1
    a.out, a.Cout, a.ldo, a.pro_mode, a.sub2 = x.ptr, cout, (cout + 3) & ~3, pro_mode, int(sub2)
    if skip:
        a.skip_w = x.ptr
    return bool(load().mud_conv2d_mfma_prec_supported(C.byref(a), prec))


def choose_prec(x: View, cout, pro_mode, *, skip=False, sub2=False, w_bstride=0):
    """The plan a 3x3 mud_conv2d_mfma launch of this shape runs with."""
    if PREC_PLAN == 'off' or w_bstride or x.C % 4:
        return PREC_16X3
    if PREC_PLAN != 'all' and not fp8x_pays(x.B, x.H, x.W, x.C, cout):
        return PREC_16X3
    return PREC_FP8X if conv_prec_supported(x, cout, pro_mode, PREC_FP8X, skip=skip, sub2=sub2) else PREC_16X3


def conv(x: View, w, ks, Cout, *, mfma, stride=1, pad=None, pro=None, bias=None, bias2=None, res: View = None,
         out_scale=1.0, act=ACT_NONE, out: View = None, w_bstride=0, arena=None, sub2=False, emul: View = None, gate=None, emul_cout=0,
         skip=None, prec=PREC_16X3, w_exp=0):
    """One fused convolution launch.  pro = (scale [B,Cin], shift [B,Cin], mode).
    skip = (packed 1x1 weights, bias or None, out View): the same launch also writes the 1x1 convolution of the RAW input
    (the residual block's Conv_2) - see fused_skip_ok().  prec / w_exp: the arithmetic plan `w` was packed for."""
    lib = load()
    pad = ks // 2 if pad is None else pad
    Ho = (x.H + 2 * pad - ks) // stride + 1
    Wo = (x.W + 2 * pad - ks) // stride + 1
    if sub2:      # stride-2 pad-0 result obtained from the stride-1 pad-1 kernel by keeping odd positions
        assert mfma and ks == 3 and stride == 1 and x.H % 2 == 1 and x.W % 2 == 1
        Ho, Wo = x.H // 2, x.W // 2
    if out is None:
        out = View.empty(x.B, Ho, Wo, Cout, x.device, arena)
    assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, Cout), ((out.B, out.H, out.W, out.C), (x.B, Ho, Wo, Cout))
    a = ConvArgs()
    a.x, a.B, a.H, a.W, a.Cin, a.ldx = x.ptr, x.B, x.H, x.W, x.C, x.ld
    a.w, a.w_bstride = ptr(w), w_bstride
    a.ks, a.stride, a.pad = ks, stride, pad
    keep = None          # tensors the launch reads that nothing else references (lazy GroupNorm operands, split-K slabs)
    if pro is not None and isinstance(pro[0], LazyGN) and not (mfma and x.C <= 1024 and pro[0].x.stats is not None):
        pro = resolve_pro(pro)
    if pro is not None and pro[2] == PRO_LRELU:
        a.pro_mode = PRO_LRELU
    elif pro is not None and isinstance(pro[0], LazyGN):      # finalised inside the kernel's prologue
        gn = keep = pro[0]
        assert gn.x.C == x.C and gn.x.B == x.B
        a.pro_mode = pro[2]
        a.gn_sums, a.gn_sums_ld, a.gn_G, a.gn_eps, a.gn_count = gn.x.stats_ptr, gn.x.stats.shape[1], gn.G, gn.eps, float(gn.x.H * gn.x.W)
        a.gn_gamma, a.gn_beta, a.gn_bstride = ptr(gn.gamma), ptr(gn.beta), gn.bstride
    elif pro is not None:
        sc, sh, mode = pro
        assert sc.shape == (x.B, x.C) and sc.stride(1) == 1 and sh.stride() == sc.stride()
        a.pro_scale, a.pro_shift, a.pro_ld, a.pro_mode = ptr(sc), ptr(sh), sc.stride(0), mode
    else:
        a.pro_mode = PRO_NONE
    if bias is not None:
        assert bias.numel() == Cout and bias.is_contiguous()
        a.bias = ptr(bias)
    if bias2 is not None:
        assert bias2.shape == (x.B, Cout) and bias2.stride(1) == 1
        a.bias2, a.bias2_ld = ptr(bias2), bias2.stride(0)
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (x.B, Ho, Wo, Cout)
        a.res, a.ldr = res.ptr, res.ld
    a.out_scale, a.act = out_scale, act
    a.sub2 = 1 if sub2 else 0
    if emul is not None:          # v *= emul (on the first emul_cout output channels only, when given)
        assert (emul.B, emul.H, emul.W, emul.C) == (x.B, Ho, Wo, emul_cout or Cout)
        a.emul, a.ld_emul, a.emul_cout = emul.ptr, emul.ld, emul_cout
    if gate is not None:          # v = g*v + (1-g)*other
        gv, ov = gate
        assert (gv.B, gv.H, gv.W, gv.C) == (x.B, Ho, Wo, Cout) == (ov.B, ov.H, ov.W, ov.C)
        a.egate, a.ld_egate, a.eother, a.ld_eother = gv.ptr, gv.ld, ov.ptr, ov.ld
    a.out, a.Cout, a.ldo = out.ptr, Cout, out.ld
    a.prec, a.w_exp = prec, w_exp
    if out.stats is not None:
        a.stats, a.stats_ld = out.stats_ptr, out.stats.shape[1]
    skip_flops = 0.0
    if skip is not None:
        sw, sb, so = skip
        assert mfma and ks == 3 and res is None and (so.B, so.H, so.W, so.C) == (x.B, Ho, Wo, Cout)
        a.skip_w, a.skip_bias, a.skip_out, a.skip_ldo = ptr(sw), ptr(sb), so.ptr, so.ld
        skip_flops = 2.0 * x.B * Ho * Wo * Cout * x.C
    if mfma and ks == 3:          # small grids (one slice at a time): split-K slabs, stream-ordered scratch (graph-capture safe)
        cnt = splitk_counters(x.device)
        a.splitk_counters, a.splitk_ncounters = ptr(cnt), cnt.numel()
        nws = lib.mud_conv2d_mfma_splitk_bytes(C.byref(a))
        if nws > 0:
            keep = (keep, torch.empty(nws, device=x.device, dtype=torch.uint8))
            a.splitk_ws, a.splitk_ws_bytes = ptr(keep[1]), nws
    fn = lib.mud_conv2d_mfma if mfma else lib.mud_conv2d_direct
    name = (f'conv_mfma_k{ks}' if mfma else f'conv_direct_k{ks}') + ('_fp8x' if prec == PREC_FP8X else '')
    flops = 2.0 * x.B * Ho * Wo * Cout * x.C * ks * ks + skip_flops     # algorithmic (sub2 issues 4x this)
    nbytes = 4.0 * (x.npix * x.C + out.npix * Cout * ((2 if res is not None else 1) + (1 if skip is not None else 0))) + (w.numel() * w.element_size() if w_bstride == 0 else x.B * w_bstride)
    _launch(name, x.device, fn, C.byref(a), STREAM, flops=flops, nbytes=nbytes)
    return out


# ---------------------------------------------------------------------------------------------------
def upfirdn2d_planes(x, kernel, up, down, pad):
    """The reference's native-op boundary: x [N,C,H,W] (any float dtype is computed in fp32)."""
    require_gpu(x, kernel)
    N, Cc, H, W = x.shape
    xin = _f32(x.float().contiguous())
    k = _f32(kernel.float().contiguous())
    kh, kw = k.shape
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad
    Ho = (H * uy + py0 + py1 - kh) // dy + 1
    Wo = (W * ux + px0 + px1 - kw) // dx + 1
    out = torch.empty(N, Cc, Ho, Wo, device=x.device, dtype=torch.float32)
    _launch('upfirdn2d', x.device, load().mud_upfirdn2d, ptr(xin), N * Cc, H, W, ptr(k), kh, kw, ux, uy, dx, dy, px0, px1, py0, py1, ptr(out), STREAM)
    return out.to(x.dtype)


def fir_nhwc(x: View, kernel2d, up, down, pad, pro=None, want_h=True, want_x=False):
    """kernel2d: host list of lists / numpy [kh,kw].  -> (out_h or None, out_x or None)."""
    import numpy as np
    k = np.ascontiguousarray(kernel2d, dtype=np.float32)
    kh, kw = k.shape
    Ho = (x.H * up + pad[0] + pad[1] - kh) // down + 1
    Wo = (x.W * up + pad[0] + pad[1] - kw) // down + 1
    oh = View.empty(x.B, Ho, Wo, x.C, x.device) if want_h else None
    ox = View.empty(x.B, Ho, Wo, x.C, x.device) if want_x else None
    sc = sh = None
    ld = mode = 0
    if pro is not None:
        sc, sh, mode = resolve_pro(pro)
        ld = sc.stride(0)
    _launch('fir_nhwc', x.device, load().mud_fir_nhwc, x.ptr, x.B, x.H, x.W, x.C, x.ld, k.ctypes.data_as(C.POINTER(C.c_float)), kh, kw, up, down,
            pad[0], pad[1], ptr(sc), ptr(sh), ld, mode, oh.ptr if oh else None, oh.ld if oh else 0, ox.ptr if ox else None,
            ox.ld if ox else 0, STREAM, nbytes=4.0 * x.C * (x.npix + x.B * Ho * Wo * (int(want_h) + int(want_x))))
    return oh, ox


def minibatch_stddev(x: View, group):
    """-> [B] tensor: the critic's minibatch-stddev scalar of every sample's group."""
    out = torch.empty(x.B, device=x.device, dtype=torch.float32)
    _launch('minibatch_stddev', x.device, load().mud_minibatch_stddev, x.ptr, x.B, x.H * x.W, x.C, x.ld, group, ptr(out), STREAM)
    return out


def attention_supported(C_):
    return bool(load().mud_attention_supported(C_))


def attention(qkv: View, C_, scale):
    """qkv: view [B,H,W,3C] (q | k | v, contiguous rows) -> View [B,H,W,C]."""
    assert qkv.C == 3 * C_
    n = qkv.H * qkv.W
    out = View.empty(qkv.B, qkv.H, qkv.W, C_, qkv.device)
    nws = load().mud_attention_ws_bytes(qkv.B, n, C_)        # > 0: few workgroups, the keys are split and merged
    ws = torch.empty(nws // 4, device=qkv.device, dtype=torch.float32) if nws else None
    _launch('attention', qkv.device, load().mud_attention, qkv.ptr, qkv.B, n, C_, qkv.ld, float(scale), out.ptr, out.ld, ptr(ws) if nws else None,
            STREAM, flops=4.0 * qkv.B * n * n * C_)
    return out


def softmax_rows_(s, n):
    """in-place softmax over the last axis of a contiguous [..., n] tensor."""
    rows = s.numel() // n
    _launch('softmax_rows', s.device, load().mud_softmax_rows, ptr(s), rows, n, n, STREAM)
    return s


def mul(a: View, b: View, out: View = None):
    out = View.empty(a.B, a.H, a.W, a.C, a.device) if out is None else out
    _launch('mul', a.device, load().mud_mul, a.ptr, a.ld, b.ptr, b.ld, out.ptr, out.ld, a.npix, a.C, STREAM)
    return out


def gate_mix(g: View, att: View, other: View, out: View):
    _launch('gate_mix', g.device, load().mud_gate_mix, g.ptr, g.ld, att.ptr, att.ld, other.ptr, other.ld, out.ptr, out.ld, g.B, g.H * g.W, g.C,
            out.stats_ptr, out.stats.shape[1] if out.stats is not None else 0, STREAM)
    return out


def resize_bilinear(x, size):
    """x [..., H, W] planes -> [..., Ho, Wo]; `F.interpolate(x, size, mode='bilinear', align_corners=False)` semantics
    (reference engine/test_volume.py:274, engine/train.py:959)."""
    require_gpu(x)
    H, W = x.shape[-2:]
    Ho, Wo = int(size[0]), int(size[1])
    xin = _f32(x.float().contiguous())
    out = torch.empty(*x.shape[:-2], Ho, Wo, device=x.device, dtype=torch.float32)
    _launch('resize_bilinear', x.device, load().mud_resize_bilinear, ptr(xin), xin.numel() // (H * W), H, W, Ho, Wo, ptr(out), STREAM)
    return out


def affine_clamp(x, scale, shift, lo, hi):
    """clamp(x*scale + shift, lo, hi) elementwise (fp32)."""
    require_gpu(x)
    xin = _f32(x.float().contiguous())
    out = torch.empty_like(xin)
    _launch('affine_clamp', x.device, load().mud_affine_clamp, ptr(xin), xin.numel(), float(scale), float(shift), float(lo), float(hi), ptr(out), STREAM)
    return out


def to_range_0_1(x):
    """[-1,1] -> [0,1] with clipping: ((x + 1) / 2).clamp(0, 1) (reference engine/test_volume.py:281)."""
    return affine_clamp(x, 0.5, 0.5, 0.0, 1.0)


INV_SQRT2 = 1.0 / math.sqrt(2.0)
