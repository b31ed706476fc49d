"""A few launches of the 3x3 MFMA conv on representative layer shapes (both arithmetic plans) and of the attention kernel, for
rocprofv3 --pmc runs.    python scripts/pmc_conv.py [B] [16x3|fp8x|both]"""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
dev = 'cuda:0'
for H, Cin, Cout in [(128, 128, 128), (256, 64, 64), (64, 256, 256), (128, 384, 128), (256, 320, 64)]:
    x = ops.View(torch.randn(B, H, H, Cin, device=dev), B, H, H, Cin)
    wt = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    out = ops.View.empty(B, H, H, Cout, dev)
    if which in ('16x3', 'both'):
        w = ops.pack_conv_weight(wt)
        for _ in range(3):
            ops.conv(x, w, 3, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), out=out)
    if which in ('fp8x', 'both') and ops.conv_prec_supported(x, Cout, ops.PRO_AFFINE_SILU, ops.PREC_FP8X):
        we = ops.fp8x_weight_exponent(wt)
        w8 = ops.pack_conv_weight(wt, prec=ops.PREC_FP8X, w_exp=we)
        for _ in range(3):
            ops.conv(x, w8, 3, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), out=out, prec=ops.PREC_FP8X, w_exp=we)
    torch.cuda.synchronize()
qkv = ops.View(torch.randn(B, 64, 64, 768, device=dev), B, 64, 64, 768)
for _ in range(3):
    ops.attention(qkv, 256, 256 ** -0.5)
torch.cuda.synchronize()
