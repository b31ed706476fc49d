"""Throughput + per-layer table of BASELINE config 5 shapes (8 reverse steps, ch_mult 1-1-2-2-4, attention at 16x16 in the down / up
paths and at the bottom) through the captured sampler.   python scripts/bench_config5.py [batch]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
import bench
from mudiff_hip import ops, sampling as S
from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
cfg = bench.bench_config()
cfg.ch_mult, cfg.num_timesteps = [1, 1, 2, 2, 4], 8
g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
bench.random_weights_(g1, 1); bench.random_weights_(g2, 2)
g1, g2 = g1.to(dev).eval(), g2.to(dev).eval()
coef = S.Posterior_Coefficients(cfg, dev)
c1, c2, c3 = bench.synthetic_batch(cfg, B, dev, seed=3)
x = torch.randn(B, 1, 256, 256, device=dev)
smp = S.GraphSampler(coef, g1, g2, cfg, B, 256, 256, dev)
for _ in range(2): smp.sample(c1, c2, c3, x, 8)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 4
for _ in range(n): smp.sample(c1, c2, c3, x, 8)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'config 5 shapes, batch {B}: {n * B / dt:.2f} slices/s ({1e3 * dt / n:.1f} ms per 8-step batch)')
ops.PROFILE.enable()
t = torch.full((B,), 3, dtype=torch.int64, device=dev); z = torch.randn(B, cfg.nz, device=dev)
orig = ops._launch
def tagged(name, dev_, fn, *args, flops=0.0, nbytes=0.0):
    if name.startswith('conv_'):
        a = args[0]._obj
        name = f'{name} {a.H:3d}^2 {a.Cin:4d}->{a.Cout:4d} pro{a.pro_mode} res{int(bool(a.res))} skip{int(bool(a.skip_w))}'
    return orig(name, dev_, fn, *args, flops=flops, nbytes=nbytes)
ops._launch = tagged
y1 = g1(x, c1, c2, c3, t, z); y2 = g2(x, c1, c2, c3, t, z, y1)
prof = ops.PROFILE.summary(); ops.PROFILE.disable()
tot = sum(v['ms'] for v in prof.values())
print(f'total {tot:.2f} ms for G1+G2 at B={B}')
for nme, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:45]:
    tf = v['flops'] / v['ms'] / 1e9 if v['flops'] else 0
    print(f'{v["ms"]:8.3f} ms  n={v["n"]:3d}  {v["ms"]/v["n"]*1e3:8.1f} us/launch  {tf:7.1f} TF  {nme}')
