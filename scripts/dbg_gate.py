import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
gen=torch.Generator().manual_seed(1)
for C,H,W in ((48,9,7),(48,16,16),(16,9,7),(192,9,7)):
    a,b,c=(torch.randn(2,C,H,W,generator=gen) for _ in range(3))
    arena=ops.StatsArena(torch.device('cuda:0'))
    out=ops.View.empty(2,H,W,C,'cuda:0',arena)
    ref=a*b+(1-a)*c
    ops.gate_mix(*(ops.View.from_nchw(t.cuda()) for t in (a,b,c)), out)
    d=(out.stats[...,0].cpu()-ref.sum(dim=(2,3)).double()).abs()
    print(C,H,W,'bad channels', (d>1e-3).nonzero().tolist()[:40], 'max', d.max().item())
    # ones test: count per channel
    one=torch.ones(2,C,H,W)
    out2=ops.View.empty(2,H,W,C,'cuda:0',arena)
    ops.gate_mix(*(ops.View.from_nchw(t.cuda()) for t in (one,one,one)), out2)
    print('  counts', out2.stats[0,:,0].cpu().tolist()[:24], 'expected', H*W)
