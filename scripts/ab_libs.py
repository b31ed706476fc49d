"""Per layer shape, ONE process, interleaved rounds, random data: the shipped library against a variant build
(scripts/build_variants.py) on the 3x3 launches of a G1 + G2 pass under the 'auto' plan (fp8 cross terms where built, fused skip conv
where the block has one).  Both libraries are loaded side by side and the loader's handle is swapped between timed runs, so box-to-box
and minute-to-minute clock drift cancels.
    MUDIFF_ALLOW_VARIANT=1 python scripts/ab_libs.py <variant name> [B] [rounds]"""
import math, os, sys
sys.path[:0] = ['/root/repo', '/root/repo/mu-diff_amd']
import numpy as np, torch
import mudiff_hip as M
from mudiff_hip import ops
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
dev = 'cuda:0'
os.environ['MUDIFF_ALLOW_VARIANT'] = '1'
libA = M.load()
M._lib, M._LIB_PATH = None, f'/root/repo/mu-diff_amd/mudiff_hip/variants/lib_{name}.so'
libB = M.load()
print('A:', M._SHIPPED, '| B:', M._LIB_PATH, libB.mud_build_flags())
# (H, Cin, Cout, pro, residual, skip conv, launches per G1+G2 pass): the 3x3 MFMA launches of a pass (as scripts/ab_prec.py)
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


tot = [0.0, 0.0]
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
    outs = [ops.View.empty(B, H, H, Cout, dev, arena) for _ in range(2)]
    so = ops.View.empty(B, H, H, Cout, dev, arena)
    M._lib = libA
    p1 = ops.pack_conv_weight(w1)
    sup = ops.conv_prec_supported(x, Cout, pro, ops.PREC_FP8X, skip=bool(skip))
    we = ops.fp8x_weight_exponent(w) if sup else 0
    pw = ops.pack_conv_weight(w, prec=ops.PREC_FP8X, w_exp=we) if sup else ops.pack_conv_weight(w)
    kw = dict(mfma=True, pro=prol, bias2=b2, res=r, out_scale=0.7071 if res else 1.0)
    if sup:
        kw.update(prec=ops.PREC_FP8X, w_exp=we)
    if skip:
        kw.update(skip=(p1, None, so))

    def run(lib, out):
        M._lib = lib
        ops.conv(x, pw, 3, Cout, out=out, **kw)
    run(libA, outs[0]); run(libB, outs[1])
    torch.cuda.synchronize()
    diff = float((outs[0].base - outs[1].base).abs().max())
    ts = [[], []]
    for _ in range(rounds):
        for i, lib in enumerate((libA, libB)):
            ts[i].append(timed(lambda: run(lib, outs[i])))
    a, b = float(np.median(ts[0])), float(np.median(ts[1]))
    tot[0] += a * cnt; tot[1] += b * cnt
    print(f'{H:4d}^2 {Cin:4d}->{Cout:4d} pro{pro} res{res} skip{skip} x{cnt:2d} {"fp8x" if sup else "16x3"}: shipped {a:7.1f}us  {name} {b:7.1f}us ({b / a:.3f})  max|diff| {diff:.1e}', flush=True)
print(f'weighted per G1+G2 pass: shipped {tot[0] / 1e3:.2f} ms, {name} {tot[1] / 1e3:.2f} ms ({tot[1] / tot[0]:.3f})')
