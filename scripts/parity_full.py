"""Per-step parity of the HIP path against ALL full-size reference-made fixtures (config 2, wide config 3, config 5), one line
each - the numbers a precision change is judged on.  Honours MUDIFF_HIP_LIB (+ MUDIFF_ALLOW_VARIANT=1).
    python scripts/parity_full.py [cfg2] [cfg3w] [cfg5]"""
import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/mu-diff_amd', '/root/repo/tests']
import numpy as np, torch
from helpers import demo_conds, load_golden, sampler_inputs, wide_cfg3_case
from oracle import mudiff_oracle as O
from mudiff_hip import sampling as S
import mudiff_hip
from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
dev = 'cuda:0'
which = sys.argv[1:] or ['cfg2', 'cfg3w', 'cfg5']
print('library:', mudiff_hip.lib_path(), getattr(mudiff_hip.load(), 'mud_build_flags', lambda: b'?')())


def build(cfg):
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    g1.load_state_dict(O.make_state_dict(cfg, 'g1', 1234)); g2.load_state_dict(O.make_state_dict(cfg, 'g2', 1234))
    return g1.to(dev).eval(), g2.to(dev).eval()


g = lambda t: t.to(dev)
for name in which:
    if name == 'cfg3w':
        cfg = O.default_config()
        g1, g2 = build(cfg)
        case = wide_cfg3_case(cfg, copies=2)
        sm = S.GraphSampler(S.Posterior_Coefficients(cfg, dev), g1, g2, cfg, 32, 256, 256, dev)
        out, steps = sm.sample(*[g(c) for c in case['conds']], g(case['x_init']), 4, zs=[g(z) for z in case['zs']], noises=[g(n) for n in case['noises']], return_steps=True)
        errs = []
        for k, st in enumerate(steps):
            e = (st[2].cpu().view(2, 16, -1) - case['refs'][k].view(1, 16, -1)).abs().amax(dim=2).amax(dim=0)
            errs.append(e)
        E = torch.stack(errs)      # [step, slice]
        print('cfg3w per-step worst over 16 slices:', ' '.join(f'{float(v):.2e}' for v in E.amax(dim=1)), '| per-slice worst:', ' '.join(f'{float(v) * 1e4:.1f}' for v in E.amax(dim=0)), '(x1e-4)')
        del sm
    else:
        cfg = O.default_config() if name == 'cfg2' else O.default_config(ch_mult=[1, 1, 2, 2, 4], num_timesteps=8, attn_resolutions=(16,))
        gd = load_golden('full_cfg2.npz' if name == 'cfg2' else 'full_cfg5.npz')
        g1, g2 = build(cfg)
        x_init, zs, noises = sampler_inputs(cfg, 1)
        conds = [g(c) for c in demo_conds()]
        x, steps = S.sample_from_model(S.Posterior_Coefficients(cfg, dev), g1, conds[0], g2, conds[1], conds[2], cfg.num_timesteps, g(x_init), None, cfg,
                                       zs=[g(z) for z in zs], noises=[g(n) for n in noises], return_steps=True)
        per = []
        for k, st in enumerate(steps):
            per.append(max(float((v.cpu() - gd[f'step{k}.{nm}']).abs().max()) for nm, v in zip(('x01', 'x02', 'xnew'), st) if f'step{k}.{nm}' in gd))
        print(f'{name} per-step max-abs:', ' '.join(f'{e:.2e}' for e in per))
    del g1, g2
    torch.cuda.empty_cache()
