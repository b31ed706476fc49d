"""Run ONE conv shape repeatedly (for rocprofv3 --pmc):  python scripts/one_conv.py H Cin Cout ks B reps"""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
H, Cin, Cout, ks, B, reps = [int(v) for v in sys.argv[1:7]]
dev = 'cuda:0'
x = ops.View(torch.randn(B, H, H, Cin, device=dev), B, H, H, Cin)
w = ops.pack_conv_weight(torch.randn(Cout, Cin, ks, ks, device=dev) / math.sqrt(Cin * ks * ks))
sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
out = ops.View.empty(B, H, H, Cout, dev)
for _ in range(reps):
    ops.conv(x, w, ks, Cout, mfma=True, pro=(sc, sh, ops.PRO_AFFINE_SILU), out=out)
torch.cuda.synchronize()
