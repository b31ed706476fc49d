"""Micro-benchmark of the MFMA conv kernel on the layer shapes of config 2 (B slices)."""
import sys, math
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = [ s for s in [  # H, Cin, Cout, ks
    (256, 64, 64, 3), (256, 256, 64, 3), (256, 192, 64, 3), (256, 128, 64, 3), (256, 320, 64, 3), (256, 192, 384, 3),
    (128, 64, 128, 3), (128, 128, 128, 3), (128, 384, 128, 3), (128, 256, 128, 3), (128, 192, 128, 3),
    (64, 128, 256, 3), (64, 256, 256, 3), (64, 512, 256, 3), (64, 384, 256, 3),
    (256, 256, 64, 1), (128, 64, 128, 1), (64, 128, 256, 1), (64, 256, 768, 1), (64, 256, 256, 1), (64, 512, 256, 1), (128, 384, 128, 1),
] if (len(sys.argv) < 3 or (sys.argv[2] == "only128" and s[2] % 128 == 0) or (sys.argv[2] == "only64" and s[2] == 64 and s[3] == 3)) ]
dev = 'cuda:0'
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = 0
for H, Cin, Cout, ks in shapes:
    x = ops.View(torch.randn(B, H, H, Cin, device=dev), B, H, H, Cin)
    w = ops.pack_conv_weight(torch.randn(Cout, Cin, ks, ks, device=dev) / math.sqrt(Cin * ks * ks))
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    out = ops.View.empty(B, H, H, Cout, dev)
    for pro in (None, (sc, sh, ops.PRO_AFFINE_SILU)):
        ms = timeit(lambda: ops.conv(x, w, ks, Cout, mfma=True, pro=pro, out=out))
        fl = 2.0 * B * H * H * Cout * Cin * ks * ks
        print(f'{H:4d}^2 {Cin:4d}->{Cout:4d} k{ks} pro={"silu" if pro else "none"}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF alg  ({3*fl/ms/1e9/2500*100:5.1f}% of the 16-bit MFMA peak issued)')
# attention-shaped GEMMs
N, C = 4096, 256
q = ops.View(torch.randn(B, 1, N, 3 * C, device=dev), B, 1, N, C, 3 * C, 0)
kp = ops.pack_weights(q.base, 0, 1, 3 * C, 1, C, N, nbatch=B, src_bstride=N * 3 * C, src_offset=C)
ms = timeit(lambda: ops.conv(q, kp, 1, N, mfma=True, w_bstride=kp.shape[1]))
print(f'QK^T  N={N} C={C}: {ms*1e3:8.1f} us {2.0*B*N*N*C/ms/1e9:7.1f} TF alg')
s = ops.View.empty(B, 1, N, N, dev)
vp = ops.pack_weights(q.base, 0, 3 * C, 1, 1, N, C, nbatch=B, src_bstride=N * 3 * C, src_offset=2 * C)
ms = timeit(lambda: ops.conv(s, vp, 1, C, mfma=True, w_bstride=vp.shape[1]))
print(f'PV    N={N} C={C}: {ms*1e3:8.1f} us {2.0*B*N*N*C/ms/1e9:7.1f} TF alg')
