"""A few launches of the 3x3 MFMA conv on representative layer shapes and of the attention kernel, for rocprofv3 --pmc runs."""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = 'cuda:0'
for H, Cin, Cout in [(128, 128, 128), (256, 64, 64), (64, 256, 256), (128, 384, 128), (256, 320, 64)]:
    x = ops.View(torch.randn(B, H, H, Cin, device=dev), B, H, H, Cin)
    w = ops.pack_conv_weight(torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9))
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    out = ops.View.empty(B, H, H, Cout, dev)
    for _ in range(3):
        ops.conv(x, w, 3, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), out=out)
    torch.cuda.synchronize()
qkv = ops.View(torch.randn(B, 64, 64, 768, device=dev), B, 64, 64, 768)
for _ in range(3):
    ops.attention(qkv, 256, 256 ** -0.5)
torch.cuda.synchronize()
