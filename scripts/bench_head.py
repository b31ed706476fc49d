"""Head conv (Cin=1 -> 64, 256^2) and FIR timings with / without the statistics epilogue."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
from backbones import up_or_down_sampling as ud
B, dev = 16, 'cuda:0'
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = ops.View(torch.randn(B, 256, 256, 1, device=dev), B, 256, 256, 1)
w = ops.direct_weight(torch.randn(64, 1, 3, 3, device=dev))
bias = torch.randn(64, device=dev)
out = ops.View.empty(B, 256, 256, 64, dev)
print('head no stats  %.1f us' % timeit(lambda: ops.conv(x, w, 3, 64, mfma=False, bias=bias, out=out)))
arena = ops.StatsArena(dev)
outs = ops.View.empty(B, 256, 256, 64, dev, arena)
print('head + stats   %.1f us' % timeit(lambda: ops.conv(x, w, 3, 64, mfma=False, bias=bias, out=outs)))
for C, H in ((64, 256), (128, 128)):
    xx = ops.View(torch.randn(B, H, H, C, device=dev), B, H, H, C)
    sc, sh = torch.rand(B, C, device=dev) + 0.5, torch.randn(B, C, device=dev)
    kk, up, down, pad = ud.fir_params('down', (1, 3, 3, 1))
    nb = 4 * B * H * H * C * (1 + 0.5)
    t = timeit(lambda: ops.fir_nhwc(xx, kk, up, down, pad, pro=(sc, sh, ops.PRO_AFFINE_SILU), want_h=True, want_x=True)); print(f'fir down dual silu C={C} H={H}: {t:.1f} us {nb/t/1e6:.2f} TB/s')
    t = timeit(lambda: ops.fir_nhwc(xx, kk, up, down, pad, pro=None, want_h=True, want_x=True)); print(f'fir down dual none C={C} H={H}: {t:.1f} us {nb/t/1e6:.2f} TB/s')
    t = timeit(lambda: ops.fir_nhwc(xx, kk, up, down, pad, pro=None, want_h=True, want_x=False)); print(f'fir down single none C={C} H={H}: {t:.1f} us')
    kk, up, down, pad = ud.fir_params('up', (1, 3, 3, 1))
    nb = 4 * B * H * H * C * (1 + 4)
    t = timeit(lambda: ops.fir_nhwc(xx, kk, up, down, pad, pro=(sc, sh, ops.PRO_AFFINE_SILU))); print(f'fir up silu C={C} H={H}: {t:.1f} us {nb/t/1e6:.2f} TB/s')
    t = timeit(lambda: ops.fir_nhwc(xx, kk, up, down, pad)); print(f'fir up none C={C} H={H}: {t:.1f} us {nb/t/1e6:.2f} TB/s')
