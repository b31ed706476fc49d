"""How much operand precision do the 3x3 convolutions need?  Runs the CPU oracle on the reference's own config-2 fixture
(tests/golden/full_cfg2.npz) with the INPUT ACTIVATIONS and / or WEIGHTS of every 3x3 convolution that the MFMA kernel serves
(Cin % 4 == 0, Cin >= 16) rounded to a shorter format, and reports the max-abs deviation per diffusion step against the
reference's recorded outputs (bar: 1e-3).  CPU only.    python scripts/exp_operand_rounding.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import bench
from oracle import mudiff_oracle as O

from tests.helpers import demo_conds, load_golden, sampler_inputs

ref = load_golden('full_cfg2.npz')
cfg = O.default_config()
sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
conds = demo_conds()
x_init, zs, noises = sampler_inputs(cfg, 1)
coef = O.PosteriorCoefficients(cfg)


def rnd(t, fmt):
    if fmt == 'fp32':
        return t
    if fmt == 'fp16':
        return t.half().float()
    if fmt == 'bf16':
        return t.bfloat16().float()
    if fmt == 'tf32':                      # 10 mantissa bits, round to nearest even on the bit pattern
        i = t.contiguous().view(torch.int32)
        i = (i + 0x0FFF + ((i >> 13) & 1)) & ~0x1FFF
        return i.view(torch.float32)
    if fmt == 'bf16x2':                    # hi + lo in bf16: 16 bits (what the shipped kernel keeps of each operand)
        h = t.bfloat16().float()
        return h + (t - h).bfloat16().float()
    raise ValueError(fmt)


def q_elem(x, fmt):
    """round to nearest representable value of an MX element format (already divided by the block scale)"""
    if fmt == 'e4m3':                      # OCP e4m3fn: max 448, 3 mantissa bits, subnormals to 2^-9
        return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    if fmt == 'e5m2':
        return x.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).float()
    if fmt == 'e2m3':                      # fp6: max 7.5, 3 mantissa bits, exponents 2^0..2^2, subnormal step 1/8
        a = x.abs().clamp(max=7.5)
        step = torch.where(a < 2.0, torch.full_like(a, 0.125), torch.where(a < 4.0, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
        return torch.sign(x) * torch.round(a / step) * step
    raise ValueError(fmt)


EMAX = {'e4m3': 8, 'e5m2': 15, 'e2m3': 2}
MAXVAL = {'e4m3': 448.0, 'e5m2': 57344.0, 'e2m3': 7.5}
RULE = os.environ.get('MX_SCALE_RULE', 'fit')      # 'fit': smallest power of two that keeps the block maximum representable; 'ocp': floor(log2(amax)) - emax (the maximum may saturate)


def mx_quant(t, fmt, dim=1, block=32):
    """OCP MX block quantisation along `dim` (the K dimension of the MFMA): one power-of-two scale per 32 elements."""
    t = t.movedim(dim, -1)
    shp = t.shape
    K = shp[-1]
    pad = (-K) % block
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, block)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -120)
    scale = torch.exp2(torch.ceil(torch.log2(amax / MAXVAL[fmt]))) if RULE == 'fit' else torch.exp2(torch.floor(torch.log2(amax)) - EMAX[fmt])
    qd = q_elem(b / scale, fmt) * scale
    return qd.reshape(*t.shape)[..., :K].reshape(shp).movedim(-1, dim)


orig = F.conv2d
state = {'a': 'fp32', 'w': 'fp32'}


def patched(x, w, *args, **kw):
    if w.shape[-1] == 3 and w.shape[1] % 4 == 0 and w.shape[1] >= 16:
        if state['a'] == 'crossc:e4m3':              # the same with CONSTANT power-of-two scales per operand class (tap-packed K: no per-pixel block scale)
            bias = args[0] if args else kw.pop('bias', None)
            rest = args[1:] if args else ()
            xh, wh = x.half().float(), w.half().float()
            xl, wl = x - xh, w - wh
            kw_ = float(2.0 ** (7 - torch.ceil(torch.log2(w.abs().max())).item()))      # max |w| * kw_ in (64, 128]
            qc = lambda t, sc: q_elem(t * sc, 'e4m3') / sc
            y = orig(xh, wh, bias, *rest, **kw)
            y = y + orig(qc(xl, 2.0 ** 14), qc(wh, kw_), None, *rest, **kw)
            return y + orig(qc(xh, 4.0), qc(wl, kw_ * 4096.0), None, *rest, **kw)
        if state['a'].startswith('cross:'):          # fp16 hi.hi + the two cross terms in an MX element format (the proposal of DESIGN section 9)
            fmt = state['a'].split(':')[1]
            bias = args[0] if args else kw.pop('bias', None)
            rest = args[1:] if args else ()
            xh, wh = x.half().float(), w.half().float()
            xl, wl = x - xh, w - wh
            y = orig(xh, wh, bias, *rest, **kw)
            y = y + orig(mx_quant(xl, fmt), mx_quant(wh, fmt), None, *rest, **kw)
            return y + orig(mx_quant(xh, fmt), mx_quant(wl, fmt), None, *rest, **kw)
        x, w = rnd(x, state['a']), rnd(w, state['w'])
    return orig(x, w, *args, **kw)


O.F.conv2d = patched


def main():
    print('activations / weights of the 3x3 convs rounded to ...   max-abs per step vs the reference\'s recorded outputs (x01, x02, x_new)')
    cases = (('fp32', 'fp32'), ('bf16x2', 'bf16x2'), ('cross:e4m3', '-'), ('crossc:e4m3', '-'), ('cross:e2m3', '-'), ('cross:e5m2', '-'), ('fp16', 'fp32'), ('fp32', 'fp16'), ('fp16', 'fp16'), ('tf32', 'tf32'), ('bf16', 'fp32'))
    if os.environ.get('ONLY'):
        cases = tuple(c for c in cases if c[0] in os.environ['ONLY'].split(','))
    for fa, fw in cases:
        state.update(a=fa, w=fw)
        _, steps = O.sample_from_model(coef, sd1, sd2, cfg, *conds, x_init, zs, noises, return_steps=True)
        per = [max(float((v - ref[f'step{k}.{nm}']).abs().max()) for nm, v in zip(('x01', 'x02', 'xnew'), stp)) for k, stp in enumerate(steps)]
        print(f'  a={fa:10s} w={fw:7s}  ' + '  '.join(f'{e:.2e}' for e in per), flush=True)


if __name__ == '__main__':
    main()
