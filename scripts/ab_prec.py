"""Per layer shape, in one process, interleaved rounds, random data: the 16-bit x 3 plan against the fp16 + e4m3-cross-term plan
(MUD_PREC_FP8X) of the 3x3 kernel, and - for the residual blocks' Conv_0 that change the channel count - the fused skip conv
(16x3, one launch) against fp8x + the 1x1 skip conv as its own launch.  The table the 'auto' plan of ops.fp8x_pays is read off.
    python scripts/ab_prec.py [B] [rounds]"""
import math, sys
sys.path[:0] = ['/root/repo', '/root/repo/mu-diff_amd']
import numpy as np, torch
from mudiff_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda:0'
# (H, Cin, Cout, pro, residual, skip conv, launches per G1+G2 pass): the 3x3 MFMA launches of profiles/r02_p_layer_times_b16.txt
shapes = [(256, 192, 384, 0, 0, 0, 1), (256, 320, 64, 2, 0, 1, 2), (256, 256, 64, 2, 0, 1, 2), (256, 192, 64, 2, 0, 1, 2), (256, 128, 64, 2, 0, 1, 2),
          (256, 128, 128, 2, 1, 0, 2), (256, 128, 128, 0, 0, 0, 2), (256, 64, 64, 2, 0, 0, 12), (256, 64, 64, 2, 1, 0, 10),
          (128, 256, 256, 2, 1, 0, 2), (128, 256, 256, 0, 0, 0, 2), (128, 384, 128, 2, 0, 1, 2), (128, 256, 128, 2, 0, 1, 2), (128, 192, 128, 2, 0, 1, 2),
          (128, 128, 128, 2, 1, 0, 10), (128, 64, 128, 2, 0, 1, 2),
          (64, 512, 256, 2, 0, 1, 4), (64, 384, 256, 2, 0, 1, 2), (64, 256, 256, 2, 1, 0, 14), (64, 256, 256, 2, 0, 0, 6), (64, 128, 256, 2, 0, 1, 2)]


def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {}
for H, Cin, Cout, pro, res, skip, cnt in shapes:
    g = torch.Generator(device=dev).manual_seed(H + Cin + Cout)
    x = ops.View(torch.randn(B, H, H, Cin, device=dev, generator=g), B, H, H, Cin)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(Cin * 9)
    w1 = torch.randn(Cout, Cin, 1, 1, device=dev, generator=g) / math.sqrt(Cin)
    sc, sh = torch.rand(B, Cin, device=dev, generator=g) + 0.5, torch.randn(B, Cin, device=dev, generator=g)
    prol = (sc, sh, ops.PRO_AFFINE_SILU) if pro == 2 else None
    r = ops.View(torch.randn(B, H, H, Cout, device=dev, generator=g), B, H, H, Cout) if res else None
    b2 = torch.randn(B, Cout, device=dev, generator=g)
    arena = ops.StatsArena(dev)
    out, out8, so = (ops.View.empty(B, H, H, Cout, dev, arena) for _ in range(3))
    p16, p1 = ops.pack_conv_weight(w), ops.pack_conv_weight(w1)
    we = ops.fp8x_weight_exponent(w)
    p8 = ops.pack_conv_weight(w, prec=ops.PREC_FP8X, w_exp=we)
    sup = ops.conv_prec_supported(x, Cout, pro, ops.PREC_FP8X)
    kw = dict(mfma=True, pro=prol, bias2=b2, res=r, out_scale=0.7071 if res else 1.0)
    runs = {'16x3': lambda: ops.conv(x, p16, 3, Cout, out=out, **kw)}
    if sup:
        runs['fp8x'] = lambda: ops.conv(x, p8, 3, Cout, out=out8, prec=ops.PREC_FP8X, w_exp=we, **kw)
    if skip:
        runs['16x3+fused skip'] = lambda: ops.conv(x, p16, 3, Cout, out=out, skip=(p1, None, so), **kw)
        runs['16x3, skip apart'] = lambda: (ops.conv(x, p16, 3, Cout, out=out, **kw), ops.conv(x, p1, 1, Cout, mfma=True, out=so))
        if ops.conv_prec_supported(x, Cout, pro, ops.PREC_FP8X, skip=True):
            runs['fp8x+fused skip'] = lambda: ops.conv(x, p8, 3, Cout, out=out8, prec=ops.PREC_FP8X, w_exp=we, skip=(p1, None, so), **kw)
        if sup:
            runs['fp8x, skip apart'] = lambda: (ops.conv(x, p8, 3, Cout, out=out8, prec=ops.PREC_FP8X, w_exp=we, **kw), ops.conv(x, p1, 1, Cout, mfma=True, out=so))
    for fn in runs.values():
        fn()
    torch.cuda.synchronize()
    diff = float((out.base - out8.base).abs().max()) if sup else float('nan')
    ts = {k: [] for k in runs}
    for _ in range(rounds):
        for k, fn in runs.items():
            ts[k].append(timed(fn))
    med = {k: float(np.median(v)) for k, v in ts.items()}
    base = med['16x3+fused skip'] if skip else med['16x3']
    best = min((k for k in med if (('skip' in k) == bool(skip))), key=lambda k: med[k])
    for k, v in med.items():
        if ('skip' in k) == bool(skip):
            tot.setdefault(k.replace('16x3+fused skip', 'base').replace('16x3', 'base') if k in ('16x3', '16x3+fused skip') else k, 0.0)
    tot['shipped'] = tot.get('shipped', 0.0) + base * cnt
    tot['best'] = tot.get('best', 0.0) + med[best] * cnt
    fl = 2.0 * B * H * H * Cout * Cin * 9
    print(f'{H:4d}^2 {Cin:4d}->{Cout:4d} pro{pro} res{res} skip{skip} x{cnt:2d}: ' + '  '.join(f'{k} {v:7.1f}us ({v / base:.3f})' for k, v in med.items())
          + f'  | best: {best}; max|fp8x - 16x3| {diff:.2e}; {fl / base / 1e6:.0f} TF', flush=True)
print(f'weighted per G1+G2 pass: shipped {tot["shipped"] / 1e3:.2f} ms, best plan per shape {tot["best"] / 1e3:.2f} ms ({tot["best"] / tot["shipped"]:.3f})')
