"""Numerics of an experimental 3x3 conv library variant against fp64 (and against the shipped library's error on the same problem).
    MUDIFF_HIP_LIB=mu-diff_amd/mudiff_hip/variants/lib_fp8x.so python scripts/check_fp8x.py"""
import math, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch, torch.nn.functional as F
from mudiff_hip import ops
dev = 'cuda:0'
print('library:', os.environ.get('MUDIFF_HIP_LIB', '(shipped)'))
for B, H, W, Cin, Cout, res in [(2, 64, 64, 256, 256, True), (1, 256, 256, 64, 64, True), (2, 128, 128, 128, 128, False), (1, 256, 256, 192, 384, False),
                                (2, 33, 70, 320, 64, False), (1, 64, 64, 512, 256, False), (3, 16, 16, 24, 40, True)]:
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    sc, sh = torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g) if res else None
    xv = ops.View.from_nchw(x.to(dev))
    out = ops.conv(xv, ops.pack_conv_weight(w.to(dev)), 3, Cout, mfma=True, pro=(sc.to(dev), sh.to(dev), ops.PRO_AFFINE_SILU), bias=bias.to(dev),
                   res=ops.View.from_nchw(r.to(dev)) if res else None)
    h = F.silu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None])
    ref = F.conv2d(h, w.double(), bias.double(), padding=1) + (r.double() if res else 0)
    y = out.to_nchw().cpu().double()
    err = (y - ref).abs()
    print(f'{B}x{H}x{W} {Cin:4d}->{Cout:4d} res{int(res)}: max-abs {err.max():.3e}  rms {err.pow(2).mean().sqrt():.3e}  (|ref| max {ref.abs().max():.2f}, rms {ref.pow(2).mean().sqrt():.2f})  nan {int(torch.isnan(y).sum())}')
