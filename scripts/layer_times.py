"""Per-layer kernel times of one G1 + G2 forward (B slices), from HIP events around every launch."""
import sys, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from oracle import mudiff_oracle as O
from mudiff_hip import ops, sampling as S
from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = 'cuda:0'
cfg = O.default_config()
g1, g2 = NCSNpp(cfg).to(dev).eval(), NCSNpp_adaptive(cfg).to(dev).eval()
g1.load_state_dict(O.make_state_dict(cfg, 'g1', 1234)); g2.load_state_dict(O.make_state_dict(cfg, 'g2', 1234))
x = torch.randn(B, 1, 256, 256, device=dev); c = [torch.randn(B, 1, 256, 256, device=dev) for _ in range(3)]
t = torch.full((B,), 3, dtype=torch.int64, device=dev); z = torch.randn(B, cfg.nz, device=dev)
orig = ops._launch
def tagged(name, dev_, fn, *args, flops=0.0, nbytes=0.0):
    if name.startswith('conv_'):
        a = args[0]._obj
        name = f'{name} {a.H:3d}^2 {a.Cin:4d}->{a.Cout:4d} pro{a.pro_mode} res{int(bool(a.res))} st{int(bool(a.stats))} ld{a.ldx}/{a.ldo}'
    return orig(name, dev_, fn, *args, flops=flops, nbytes=nbytes)
ops._launch = tagged
for it in range(2):
    ops.PROFILE.enable()
    y1 = g1(x, *c, t, z); y2 = g2(x, *c, t, z, y1)
    torch.cuda.synchronize()
    prof = ops.PROFILE.summary(); ops.PROFILE.disable()
tot = sum(v['ms'] for v in prof.values())
print(f'total {tot:.2f} ms for G1+G2 at B={B}')
for n, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:60]:
    tf = v['flops'] / v['ms'] / 1e9 if v['flops'] else 0
    print(f'{v["ms"]:8.3f} ms  n={v["n"]:3d}  {v["ms"]/v["n"]*1e3:8.1f} us/launch  {tf:7.1f} TF  {v["bytes"]/v["ms"]/1e9:7.2f} TB/s  {n}')
