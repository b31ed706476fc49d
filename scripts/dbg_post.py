import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/mu-diff_amd')
import torch, numpy as np
from helpers import load_golden
from oracle import mudiff_oracle as O
from mudiff_hip import sampling as S
gd=load_golden('elementwise.npz'); cfg=O.default_config(); coef=S.Posterior_Coefficients(cfg,'cuda:0')
g=lambda t: t.cuda()
x01,x02,xt,t,nz=[g(gd[k]) for k in ('x01','x02','xt','t','noise')]
out=S.sample_posterior_combine(coef,x01,x02,xt,t,nz).cpu()
ref=gd['posterior_combine']
def chain(dev):
    c=S.Posterior_Coefficients(cfg,dev)
    a,b,x,tt,n=[gd[k].to(dev) for k in ('x01','x02','xt','t','noise')]
    c1=S.extract(c.posterior_mean_coef1,tt,x.shape); c2=S.extract(c.posterior_mean_coef2,tt,x.shape)
    m1=c1*a+c2*x; m2=c1*b+c2*x; mean=(m1+m2)/2
    lv=S.extract(c.posterior_log_variance_clipped,tt,x.shape)
    mask=(1-(tt==0).type(torch.float32))
    sdn=mask[:,None,None,None]*torch.exp(0.5*lv)
    return mean.cpu(), sdn.cpu(), (mean+sdn*n).cpu()
mg,sg,og=chain('cuda:0'); mc,sc,oc=chain('cpu')
print('cpu chain == golden', torch.equal(oc,ref))
print('gpu-eager mean==cpu', torch.equal(mg,mc), 'sdn', sg.flatten().tolist(), sc.flatten().tolist())
print('gpu-eager out == golden', torch.equal(og,ref), 'kernel == gpu-eager', torch.equal(out,og))
print('std table', S._std_table(coef).cpu().tolist())
