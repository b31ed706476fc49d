"""GPU parity: the HIP path (through the C ABI of libmudiff_hip.so) against the CPU oracle and the
golden vectors recorded from the reference.  Tolerances: bit-exact for the posterior / q_sample
kernels (un-contracted fp32 in reference order); <= 1e-3 max-abs per diffusion step for the generators
(BASELINE.json north_star) - the measured error is printed and asserted well below that."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import SMALL_CFGS, demo_conds, load_golden, sampler_inputs, small_conds
from oracle import mudiff_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


def _imports():
    from mudiff_hip import ops, sampling
    from backbones import layerspp, up_or_down_sampling
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    return ops, sampling, layerspp, up_or_down_sampling, NCSNpp, NCSNpp_adaptive


def g(t):
    return t.to(DEV)


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


# ----------------------------------------------------------------------------------------------
def test_library_loaded_is_in_tree():
    import os
    import mudiff_hip
    lib = mudiff_hip.load()
    assert lib.mud_version() >= 110
    assert lib.mud_build_flags() == b'', 'the GPU suite must run on the clean in-tree build, not on an experiment variant'
    assert os.path.realpath(mudiff_hip.lib_path()) == os.path.realpath(os.path.join(os.path.dirname(mudiff_hip.__file__), 'libmudiff_hip.so'))
    with open('/proc/self/maps') as f:
        assert 'libmudiff_hip.so' in f.read()


def test_posterior_and_q_sample_bit_exact():
    """The kernels reproduce the reference's chain of separately-rounded fp32 mul/add exactly: compared
    bit-for-bit with the oracle evaluated on THIS host from the same tables.  (Host-side tables use
    float64 exp/log whose last bit depends on the CPU's SIMD dispatch, so against the golden recorded in
    the build container the comparison allows 1 ulp.)"""
    ops, S, *_ = _imports()
    gd = load_golden('elementwise.npz')
    cfg = O.default_config()
    coef, dcoef = S.Posterior_Coefficients(cfg, DEV), S.Diffusion_Coefficients(cfg, DEV)
    ocoef, odcoef = O.PosteriorCoefficients(cfg), O.DiffusionCoefficients(cfg)
    kat = load_golden('kat_schedules.npz')
    for f in ('posterior_mean_coef1', 'posterior_mean_coef2', 'posterior_log_variance_clipped'):
        assert torch.equal(getattr(coef, f).cpu(), getattr(ocoef, f))
        torch.testing.assert_close(getattr(coef, f).cpu(), kat[f'T4.{f}'], rtol=3e-7, atol=0)
    x01, x02, xt, t, nz = gd['x01'], gd['x02'], gd['xt'], gd['t'], gd['noise']
    out = S.sample_posterior_combine(coef, g(x01), g(x02), g(xt), g(t), g(nz)).cpu()
    assert torch.equal(out, O.sample_posterior_combine(ocoef, x01, x02, xt, t, nz))
    assert maxdiff(out, gd['posterior_combine']) <= 2.4e-7
    out = S.sample_posterior(coef, g(x01), g(xt), g(t), g(nz)).cpu()
    assert torch.equal(out, O.sample_posterior(ocoef, x01, xt, t, nz)) and maxdiff(out, gd['posterior']) <= 2.4e-7
    out = S.q_sample(dcoef, g(x01), g(t), noise=g(nz)).cpu()
    assert torch.equal(out, O.q_sample(odcoef, x01, t, nz)) and maxdiff(out, gd['q_sample']) <= 2.4e-7
    a, b = S.q_sample_pairs(dcoef, g(x01), g(t), noise=g(gd['noise_outer']), noise_inner=g(gd['noise_inner']))
    oa, ob = O.q_sample_pairs(odcoef, x01, t, gd['noise_inner'], gd['noise_outer'])
    assert torch.equal(a.cpu(), oa) and torch.equal(b.cpu(), ob)
    assert maxdiff(a, gd['q_pair0']) <= 2.4e-7 and maxdiff(b, gd['q_pair1']) <= 4.8e-7
    # ragged / empty / unaligned sizes
    for B, shape in ((3, (1, 7, 9)), (1, (1, 1, 1)), (0, (1, 8, 8)), (2, (1, 256, 256))):
        gen = torch.Generator().manual_seed(B + shape[1])
        xs = [torch.randn(B, *shape, generator=gen) for _ in range(4)]
        tt = torch.randint(0, 4, (B,), generator=gen)
        ref = O.sample_posterior_combine(ocoef, xs[0], xs[1], xs[2], tt, xs[3])
        out = S.sample_posterior_combine(coef, g(xs[0]), g(xs[1]), g(xs[2]), g(tt), g(xs[3]))
        assert torch.equal(out.cpu(), ref)


def test_embeddings_and_dense():
    ops, *_ = _imports()
    t = torch.tensor([0, 1, 2, 3, 7, 999])
    assert maxdiff(ops.timestep_embedding(g(t), 64), O.timestep_embedding(t, 64)) < 2e-4   # |arg| up to 999
    assert maxdiff(ops.timestep_embedding(g(t[:4]), 64), O.timestep_embedding(t[:4], 64)) < 1e-6
    gen = torch.Generator().manual_seed(0)
    z = torch.randn(5, 100, generator=gen)
    ref = z / torch.sqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    assert maxdiff(ops.pixel_norm(g(z)), ref) < 1e-6
    for B, K, N in ((5, 100, 256), (1, 256, 1024), (9, 64, 33), (32, 256, 512), (20, 256, 11520), (3, 102, 70), (2, 1024, 65)):
        x, W, b = torch.randn(B, K, generator=gen), torch.randn(N, K, generator=gen) / math.sqrt(K), torch.randn(N, generator=gen)
        assert maxdiff(ops.dense(g(x), g(W), g(b)), F.linear(x, W, b)) < 5e-6
        assert maxdiff(ops.dense(g(x), g(W), g(b), act_in=ops.ACT_SILU, act_out=ops.ACT_SILU), F.silu(F.linear(F.silu(x), W, b))) < 5e-6


@pytest.mark.parametrize('B,dims,pn,act_last', [(3, [100, 256, 256, 256, 256], True, True), (1, [64, 256, 256], False, False), (17, [50, 64, 64], True, True),
                                                 (2, [37, 19, 5], False, True)])
def test_mlp_chain_one_launch(B, dims, pn, act_last):
    """mud_mlp_chain: PixelNorm + the z-mapping MLP (SiLU after every layer) / the timestep MLP (Linear, SiLU, Linear) as ONE launch
    against torch in fp64 (reference ncsnpp_generator_adagn_feat.py:44-49,271-277,301-305)."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(sum(dims) + B)
    x = torch.randn(B, dims[0], generator=gen)
    layers = [(torch.randn(n, k, generator=gen) / math.sqrt(k), torch.randn(n, generator=gen)) for k, n in zip(dims[:-1], dims[1:])]
    out = ops.mlp_chain(g(x), [(g(w), g(b)) for w, b in layers], pixel_norm=pn, act=ops.ACT_SILU, act_last=act_last)
    h = x.double()
    if pn:
        h = h * torch.rsqrt((h * h).mean(dim=1, keepdim=True) + 1e-8)
    for l, (w, b) in enumerate(layers):
        h = h @ w.double().t() + b.double()
        if l + 1 < len(layers) or act_last:
            h = F.silu(h)
    assert out.shape == h.shape and maxdiff(out, h) <= 2e-5 * max(1.0, float(h.abs().max()))


def test_mlp_chains_side_by_side():
    """mud_mlp_chains: independent chains (different widths, depths, batch sizes) in one launch give exactly what the separate
    launches give (a generator's z-mapping network and timestep MLP run this way)."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(11)
    specs = [(3, [100, 256, 256, 256, 256], True, True), (3, [128, 512, 512], False, False), (1, [37, 19, 5], False, True), (5, [64, 64], True, False)]
    chains = []
    for B, dims, pn, al in specs:
        layers = [(g(torch.randn(n, k, generator=gen) / math.sqrt(k)), g(torch.randn(n, generator=gen))) for k, n in zip(dims[:-1], dims[1:])]
        chains.append(dict(x=g(torch.randn(B, dims[0], generator=gen)), layers=layers, pixel_norm=pn, act=ops.ACT_SILU, act_last=al))
    for n in (2, 4):
        outs = ops.mlp_chains(chains[:n])
        for c, o in zip(chains[:n], outs):
            assert torch.equal(o, ops.mlp_chain(c['x'], c['layers'], pixel_norm=c['pixel_norm'], act=c['act'], act_last=c['act_last']))
    with pytest.raises(AssertionError):
        ops.mlp_chains(chains + chains[:1])


@pytest.mark.parametrize('B,H,W,C,G', [(2, 8, 8, 16, 4), (1, 64, 64, 256, 32), (3, 17, 5, 24, 6), (2, 32, 32, 192, 32),
                                       (1, 256, 256, 64, 16), (2, 16, 16, 320, 32)])
def test_groupnorm_scale_shift(B, H, W, C, G):
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(B, C, H, W, generator=gen) * 1.7 + 0.6
    gamma, beta = torch.randn(B, C, generator=gen), torch.randn(B, C, generator=gen)
    xv = ops.View.from_nchw(g(x))
    sc, sh = ops.gn_scale_shift(xv, G, g(gamma), g(beta))
    got = x * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
    ref = gamma[:, :, None, None] * F.group_norm(x, G, eps=1e-6) + beta[:, :, None, None]
    assert maxdiff(got, ref) < 2e-5
    # view with a channel offset inside a wider buffer
    wide = torch.zeros(B, H, W, C + 8, device=DEV)
    wide[..., 4:4 + C] = g(x).permute(0, 2, 3, 1)
    v = ops.View(wide, B, H, W, C + 8).slice(4, C)
    sc2, sh2 = ops.gn_scale_shift(v, G)
    ref2 = F.group_norm(x, G, eps=1e-6)
    assert maxdiff(x * sc2.cpu()[:, :, None, None] + sh2.cpu()[:, :, None, None], ref2) < 2e-5
    assert maxdiff(ops.channel_mean(v), x.mean(dim=(2, 3))) < 1e-6


@pytest.mark.parametrize('B,H,W,Cin,Cout,ks,stride,pad', [(2, 12, 20, 1, 64, 3, 1, 1), (1, 9, 7, 64, 1, 3, 1, 1), (2, 13, 13, 8, 16, 3, 2, 0),
                                                           (1, 8, 8, 3, 5, 3, 1, 1), (2, 6, 6, 16, 8, 1, 1, 0)])
def test_conv_direct(B, H, W, Cin, Cout, ks, stride, pad):
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, ks, ks, generator=gen) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=gen)
    out = ops.conv(ops.View.from_nchw(g(x)), ops.direct_weight(g(w)), ks, Cout, mfma=False, stride=stride, pad=pad, bias=g(b))
    assert maxdiff(out.to_nchw(), F.conv2d(x, w, b, stride=stride, padding=pad)) < 1e-5


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 19, 45, 64, 1), (1, 8, 32, 16, 2), (3, 33, 70, 48, 3), (1, 9, 7, 32, 4)])
def test_tail_conv_kernel(B, H, W, Cin, Cout):
    """The LDS-tiled C_out <= 4 kernel (output / pyramid convolutions): ragged tiles, GroupNorm+SiLU prologue, residual, tanh,
    against torch fp32; and the strip kernel on the stride-2 pyramid convolution (C_in = 1) with residual + statistics."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, W, generator=gen)
    sc, sh = torch.rand(B, Cin, generator=gen) + 0.5, torch.randn(B, Cin, generator=gen)
    xv = ops.View.from_nchw(g(x))
    out = ops.conv(xv, ops.direct_weight(g(w)), 3, Cout, mfma=False, bias=g(b))
    assert maxdiff(out.to_nchw(), F.conv2d(x, w, b, padding=1)) < 1e-5
    out = ops.conv(xv, ops.direct_weight(g(w)), 3, Cout, mfma=False, bias=g(b), pro=(g(sc), g(sh), ops.PRO_AFFINE_SILU),
                   res=ops.View.from_nchw(g(res)), out_scale=0.7, act=ops.ACT_TANH)
    ref = torch.tanh((F.conv2d(F.silu(x * sc[:, :, None, None] + sh[:, :, None, None]), w, b, padding=1) + res) * 0.7)
    assert maxdiff(out.to_nchw(), ref) < 2e-5
    # stride-2 pad-0 conv of a single-channel image (the input pyramid after its FIR): strip kernel, residual, statistics
    Hs, Ws = 2 * H + 1, 2 * W + 1
    xs = torch.randn(B, 1, Hs, Ws, generator=gen)
    ws = torch.randn(64, 1, 3, 3, generator=gen) / 3
    bs = torch.randn(64, generator=gen)
    rs = torch.randn(B, 64, H, W, generator=gen)
    arena = ops.StatsArena(DEV)
    o = ops.View.empty(B, H, W, 64, DEV, arena)
    ops.conv(ops.View.from_nchw(g(xs)), ops.direct_weight(g(ws)), 3, 64, mfma=False, stride=2, pad=0, bias=g(bs), res=ops.View.from_nchw(g(rs)),
             out_scale=0.5, out=o)
    refs = (F.conv2d(xs, ws, bs, stride=2) + rs) * 0.5
    assert maxdiff(o.to_nchw(), refs) < 1e-5
    assert maxdiff(o.stats[..., 0], refs.sum(dim=(2, 3))) < 1e-3 and maxdiff(o.stats[..., 1], (refs * refs).sum(dim=(2, 3))) < 1e-3


@pytest.mark.parametrize('B,H,W,Cin,Cout,ks', [(2, 8, 32, 32, 64, 3), (1, 20, 37, 48, 96, 3), (2, 16, 16, 80, 16, 3), (1, 64, 64, 256, 256, 3),
                                              (1, 33, 9, 8, 24, 3), (2, 16, 16, 64, 128, 1), (1, 5, 7, 36, 40, 1), (1, 64, 64, 256, 768, 1)])
def test_conv_mfma_split_precision(B, H, W, Cin, Cout, ks):
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin + 3 * Cout + ks)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, ks, ks, generator=gen) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=gen)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    out = ops.conv(ops.View.from_nchw(g(x)), ops.pack_conv_weight(g(w)), ks, Cout, mfma=True, bias=g(b)).to_nchw()
    err = maxdiff(out, ref)
    fp32_err = maxdiff(F.conv2d(x, w, b, padding=ks // 2), ref)
    print(f'conv_mfma {B}x{H}x{W} {Cin}->{Cout} k{ks}: split-fp16 err {err:.2e}, torch fp32 err {fp32_err:.2e}')
    assert err < 3e-5
    # fused prologue (AdaGN affine + SiLU), time-embedding bias, residual, rescale, activation
    sc, sh = torch.randn(B, Cin, generator=gen), torch.randn(B, Cin, generator=gen)
    b2, res = torch.randn(B, Cout, generator=gen), torch.randn(B, Cout, H, W, generator=gen)
    xin = F.silu(x * sc[:, :, None, None] + sh[:, :, None, None])
    ref = torch.tanh((F.conv2d(xin, w, b, padding=ks // 2) + b2[:, :, None, None] + res) * 0.5)
    out = ops.conv(ops.View.from_nchw(g(x)), ops.pack_conv_weight(g(w)), ks, Cout, mfma=True, bias=g(b), bias2=g(b2),
                   pro=(g(sc), g(sh), ops.PRO_AFFINE_SILU), res=ops.View.from_nchw(g(res)), out_scale=0.5, act=ops.ACT_TANH).to_nchw()
    assert maxdiff(out, ref) < 3e-5


@pytest.mark.parametrize('B,H,W,Cin,Cout,pro,res', [(8, 64, 64, 256, 256, True, True), (4, 128, 128, 128, 128, True, False), (1, 256, 256, 192, 384, False, False),
                                                     (16, 70, 33, 96, 128, True, True), (16, 128, 128, 192, 64, True, False),
                                                     (8, 128, 128, 64, 64, True, True), (8, 128, 128, 64, 64, False, False)])
def test_conv_fp8x_plan_vs_fp64(B, H, W, Cin, Cout, pro, res):
    """MUD_PREC_FP8X (fp16 hi.hi + both cross terms on the block-scaled e4m3 MFMA, per-layer weight exponent): against fp64, next to
    the 16-bit x 3 plan on the same problem.  ~2^-15 per product: rms error <= 4e-5 of the convolution's rms (the fp16 x 3 plan: <= 4e-6);
    weights far from unit scale exercise the per-layer exponent."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(B + H + Cin + Cout)
    for wscale in (1.0, 37.0, 1.0 / 512):
        x = torch.randn(B, Cin, H, W, generator=gen)
        w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5 * wscale
        bias = torch.randn(Cout, generator=gen)
        sc, sh = torch.rand(B, Cin, generator=gen) + 0.5, torch.randn(B, Cin, generator=gen)
        r = torch.randn(B, Cout, H, W, generator=gen) if res else None
        xv = ops.View.from_nchw(g(x))
        if not ops.conv_prec_supported(xv, Cout, ops.PRO_AFFINE_SILU if pro else ops.PRO_NONE, ops.PREC_FP8X):
            raise AssertionError('test shape must be one the plan is built for')
        kw = dict(mfma=True, pro=(g(sc), g(sh), ops.PRO_AFFINE_SILU) if pro else None, bias=g(bias), res=ops.View.from_nchw(g(r)) if res else None)
        we = ops.fp8x_weight_exponent(w)
        assert float(w.abs().max()) * 2.0 ** we <= 448.0 < float(w.abs().max()) * 2.0 ** (we + 1)
        y8 = ops.conv(xv, ops.pack_conv_weight(g(w), prec=ops.PREC_FP8X, w_exp=we), 3, Cout, prec=ops.PREC_FP8X, w_exp=we, **kw).to_nchw().cpu().double()
        y16 = ops.conv(xv, ops.pack_conv_weight(g(w)), 3, Cout, **kw).to_nchw().cpu().double()
        h = x.double()
        if pro:
            h = torch.nn.functional.silu(h * sc.double()[:, :, None, None] + sh.double()[:, :, None, None])
        ref = torch.nn.functional.conv2d(h, w.double(), bias.double(), padding=1) + (r.double() if res else 0)
        rms = lambda e: float(e.pow(2).mean().sqrt())
        conv_rms = rms(ref - (r.double() if res else 0) - bias.double()[None, :, None, None])
        e8, e16 = rms(y8 - ref), rms(y16 - ref)
        print(f'{B}x{H}x{W} {Cin}->{Cout} w x{wscale:g} (2^{we}): rms err fp8x {e8:.2e} 16x3 {e16:.2e}; conv rms {conv_rms:.2f}; max-abs fp8x {float((y8 - ref).abs().max()):.2e}')
        assert not torch.isnan(y8).any() and e8 <= 4e-5 * conv_rms + 1e-6 and e16 <= 4e-6 * conv_rms + 1e-6


def test_fp8x_plan_degrades_gracefully_out_of_range():
    """Activations outside the e4m3 images' range (|a| > 112 or < 5e-4) only lose their CROSS terms (those products fall back to
    one 11-bit piece: <= 2^-10 relative), values beyond fp16's range saturate at +-65504; nothing becomes NaN or inf."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(5)
    B, H, Cin, Cout = 4, 128, 128, 128
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / (Cin * 9) ** 0.5
    we = ops.fp8x_weight_exponent(w)
    w8 = ops.pack_conv_weight(g(w), prec=ops.PREC_FP8X, w_exp=we)
    for scale, tol, abs_tol in ((300.0, 2.0 ** -10, None), (1.0, 4e-5, None), (1e-5, None, 1e-7)):
        # (inputs below fp16's normal range, 6e-5, keep only ABSOLUTE precision - the pieces are fp16 subnormals, <= 2^-25 each:
        #  irrelevant next to O(1) activations, so that case is judged on the absolute error)
        x = torch.randn(B, Cin, H, H, generator=gen) * scale
        xv = ops.View.from_nchw(g(x))
        assert ops.conv_prec_supported(xv, Cout, ops.PRO_NONE, ops.PREC_FP8X)
        y = ops.conv(xv, w8, 3, Cout, mfma=True, prec=ops.PREC_FP8X, w_exp=we).to_nchw().cpu().double()
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
        err = float((y - ref).pow(2).mean().sqrt())
        rel = err / float(ref.pow(2).mean().sqrt())
        print(f'fp8x plan, inputs x{scale:g}: rms error {err:.2e} = {rel:.2e} of the output rms')
        assert torch.isfinite(y).all() and (rel <= tol if tol is not None else err <= abs_tol)
    x = torch.full((B, Cin, H, H), 1e6)
    y = ops.conv(ops.View.from_nchw(g(x)), w8, 3, Cout, mfma=True, prec=ops.PREC_FP8X, w_exp=we).to_nchw().cpu()
    y16 = ops.conv(ops.View.from_nchw(g(x)), ops.pack_conv_weight(g(w)), 3, Cout, mfma=True).to_nchw().cpu()
    assert torch.isfinite(y).all() and torch.isfinite(y16).all()          # saturated at 65504, not inf / NaN (both plans)


def test_c_abi_refuses_a_plan_that_is_not_built_for_the_launch():
    """Weights packed for MUD_PREC_FP8X cannot be read by another plan, so mud_conv2d_mfma must fail loudly (nothing launched)
    where the plan does not exist: small grids, the fused skip conv, 1x1 kernels."""
    import ctypes as C
    import mudiff_hip
    ops, *_ = _imports()
    lib = mudiff_hip.load()
    x = ops.View(torch.randn(1, 16, 16, 64, device=DEV), 1, 16, 16, 64)
    assert not ops.conv_prec_supported(x, 64, ops.PRO_AFFINE_SILU, ops.PREC_FP8X)
    big = ops.View(torch.randn(4, 128, 128, 128, device=DEV), 4, 128, 128, 128)
    assert ops.conv_prec_supported(big, 128, ops.PRO_AFFINE_SILU, ops.PREC_FP8X) and ops.conv_prec_supported(big, 128, ops.PRO_AFFINE_SILU, ops.PREC_FP8X, skip=True)
    assert not ops.conv_prec_supported(big, 128, ops.PRO_NONE, ops.PREC_FP8X, skip=True)        # the fused skip conv is the residual blocks' (AdaGN + SiLU prologue)
    assert not ops.conv_prec_supported(big, 128, ops.PRO_AFFINE, ops.PREC_FP8X)
    w = torch.randn(64, 64, 3, 3, device=DEV)
    with pytest.raises(mudiff_hip.MudiffHipError, match='MUD_PREC_FP8X is not built'):
        ops.conv(x, ops.pack_conv_weight(w, prec=ops.PREC_FP8X, w_exp=3), 3, 64, mfma=True, prec=ops.PREC_FP8X, w_exp=3)
    dst = torch.empty(lib.mud_packed_weight_bytes(1, 64, 64), device=DEV, dtype=torch.uint8)
    w1 = torch.randn(64, 64, device=DEV)
    assert lib.mud_pack_weights_prec(C.c_void_p(w1.data_ptr()), 0, 64, 1, 0, 1, 64, 64, 1, 1, 0, C.c_void_p(dst.data_ptr()), None) != 0    # ks == 1 has no fp8x form


def test_c_abi_refuses_prologue_arrays_wider_than_its_lds_image():
    """mud_conv2d_mfma keeps the prologue scale / shift of a sample in LDS (2 x 1024 floats): a launch with more input channels
    and pro_mode AFFINE / AFFINE_SILU must fail loudly, not read past the image; 1024 channels still run and match fp64."""
    import mudiff_hip
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(7)
    for Cin, ok in ((1024, True), (1028, False)):
        x = torch.randn(1, Cin, 8, 8, generator=gen)
        w = torch.randn(8, Cin, 1, 1, generator=gen) / math.sqrt(Cin)
        sc, sh = torch.rand(1, Cin, generator=gen) + 0.5, torch.randn(1, Cin, generator=gen)
        run = lambda: ops.conv(ops.View.from_nchw(g(x)), ops.pack_conv_weight(g(w)), 1, 8, mfma=True, pro=(g(sc), g(sh), ops.PRO_AFFINE))
        if ok:
            ref = torch.nn.functional.conv2d((x * sc[:, :, None, None] + sh[:, :, None, None]).double(), w.double())
            assert float((run().to_nchw().cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
        else:
            with pytest.raises(mudiff_hip.MudiffHipError, match='Cin <= 1024'):
                run()


@pytest.mark.parametrize('mfma,Cin,Cout,H,W', [(True, 32, 96, 20, 37), (True, 64, 64, 64, 64), (False, 1, 64, 31, 17), (False, 8, 24, 16, 16)])
def test_epilogue_statistics_equal_two_pass_groupnorm(mfma, Cin, Cout, H, W):
    """A producer's epilogue accumulates per-channel (sum, sumsq) of what it stores; GroupNorm from those
    sums must equal GroupNorm from a pass over the tensor (and torch)."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin + Cout)
    B = 3
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=gen)
    arena = ops.StatsArena(torch.device(DEV))
    wide = ops.View.empty(B, H, W, Cout + 16, DEV, arena)             # the output is a channel slice of a wider buffer
    out = wide.slice(8, Cout)
    wp = ops.pack_conv_weight(g(w)) if mfma else ops.direct_weight(g(w))
    ops.conv(ops.View.from_nchw(g(x)), wp, 3, Cout, mfma=mfma, bias=g(b), out=out)
    G = min(Cout // 4, 32)
    gamma, beta = g(torch.randn(B, Cout, generator=gen)), g(torch.randn(B, Cout, generator=gen))
    sc1, sh1 = ops.gn_scale_shift(out, G, gamma, beta)                 # from the epilogue sums
    plain = ops.View(wide.base, B, H, W, Cout, wide.ld, 8)             # same memory, no stats -> two-pass kernel
    sc2, sh2 = ops.gn_scale_shift(plain, G, gamma, beta)
    assert maxdiff(sc1, sc2) < 2e-6 and maxdiff(sh1, sh2) < 2e-6
    y = out.to_nchw().cpu()
    ref = gamma.cpu()[:, :, None, None] * F.group_norm(y, G, eps=1e-6) + beta.cpu()[:, :, None, None]
    assert maxdiff(y * sc1.cpu()[:, :, None, None] + sh1.cpu()[:, :, None, None], ref) < 2e-5


@pytest.mark.parametrize('B,H,W,C,Cout,ks,silu', [(2, 20, 37, 48, 96, 3, True), (3, 16, 16, 320, 64, 3, True), (1, 64, 64, 256, 256, 3, True),
                                                    (2, 24, 24, 192, 128, 3, False), (2, 16, 16, 64, 192, 1, False), (1, 9, 7, 24, 16, 3, True)])
def test_groupnorm_folded_into_conv_prologue(B, H, W, C, Cout, ks, silu):
    """mud_conv_args.gn_*: the consumer conv finalises the GroupNorm of its input from the producer-accumulated sums in its own
    prologue (no gn_from_sums launch).  Must equal the materialised scale / shift path on the same sums, and torch's
    group_norm -> (silu) -> conv in fp64 (AdaptiveGroupNorm + conv, reference layerspp.py:37-54,293-318)."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(C + Cout + ks)
    src = torch.randn(B, C, H, W, generator=gen) * 1.5 + 0.3
    G = min(C // 4, 32)
    arena = ops.StatsArena(torch.device(DEV))
    wide = ops.View.empty(B, H, W, C + 8, DEV, arena)                  # the input is a channel slice of a concat buffer
    x = wide.slice(4, C)
    ops.fir_nhwc(ops.View.from_nchw(g(src)), [[1.0]], 1, 1, (0, 0))[0]       # (warm the FIR path; the sums come from a conv below)
    eye = torch.zeros(C, C, 1, 1)
    eye[torch.arange(C), torch.arange(C)] = 1.0
    ops.conv(ops.View.from_nchw(g(src)), ops.pack_conv_weight(g(eye)), 1, C, mfma=True, out=x)      # producer: copies src, accumulates sums
    xs = x.to_nchw().cpu()
    gamma, beta = g(torch.randn(B, C, generator=gen)), g(torch.randn(B, C, generator=gen))
    w = torch.randn(Cout, C, ks, ks, generator=gen) / math.sqrt(C * ks * ks)
    bias = torch.randn(Cout, generator=gen)
    wp = ops.pack_conv_weight(g(w))
    mode = ops.PRO_AFFINE_SILU if silu else ops.PRO_AFFINE
    lazy = ops.gn_lazy(x, G, gamma, beta)
    assert isinstance(lazy, ops.LazyGN)
    ops.PROFILE.enable()
    out_fold = ops.conv(x, wp, ks, Cout, mfma=True, pro=(lazy, None, mode), bias=g(bias)).to_nchw().cpu()
    names = {r[0] for r in ops.PROFILE.records}
    ops.PROFILE.disable()
    assert 'gn_from_sums' not in names                                 # nothing extra was launched
    sc, sh = lazy.tensors()
    out_mat = ops.conv(x, wp, ks, Cout, mfma=True, pro=(sc, sh, mode), bias=g(bias)).to_nchw().cpu()
    assert maxdiff(out_fold, out_mat) <= 1e-6                          # same arithmetic, different place
    h = gamma.cpu().double()[:, :, None, None] * F.group_norm(xs.double(), G, eps=1e-6) + beta.cpu().double()[:, :, None, None]
    ref = F.conv2d(F.silu(h) if silu else h, w.double(), bias.double(), padding=ks // 2)
    err = maxdiff(out_fold, ref)
    print(f'folded GN + conv{ks}x{ks} {C}->{Cout}: max-abs vs fp64 {err:.2e}')
    assert err <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 20, 37, 48, 96), (16, 64, 64, 384, 256), (8, 128, 128, 192, 64), (1, 64, 64, 512, 256), (2, 33, 70, 320, 64),
                                            (3, 16, 16, 24, 40)])
def test_conv_with_fused_skip_conv(B, H, W, Cin, Cout):
    """mud_conv_args.skip_*: one launch = conv3x3(silu(affine(x))) + bias + time bias (with statistics) AND the residual block's
    1x1 skip conv of the raw x (reference layerspp.py:311-321).  Both outputs against fp64, and the 3x3 output against the
    unfused launch; covers the three tile variants (8x2, 16x1, 4-row) and ragged sizes."""
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / math.sqrt(Cin * 9)
    w2 = torch.randn(Cout, Cin, 1, 1, generator=gen) / math.sqrt(Cin)
    bias, bias_s, b2 = torch.randn(Cout, generator=gen), torch.randn(Cout, generator=gen), torch.randn(B, Cout, generator=gen)
    sc, sh = torch.rand(B, Cin, generator=gen) + 0.5, torch.randn(B, Cin, generator=gen)
    xv, wp, w2p = ops.View.from_nchw(g(x)), ops.pack_conv_weight(g(w)), ops.pack_conv_weight(g(w2))
    assert ops.fused_skip_ok(xv, Cout, ops.PRO_AFFINE_SILU)       # (also where the launch is split over K: both accumulator sets go through the slabs)
    arena = ops.StatsArena(torch.device(DEV))
    out, skip = ops.View.empty(B, H, W, Cout, DEV, arena), ops.View.empty(B, H, W, Cout + 8, DEV).slice(4, Cout)
    pro = (g(sc), g(sh), ops.PRO_AFFINE_SILU)
    ops.conv(xv, wp, 3, Cout, mfma=True, pro=pro, bias=g(bias), bias2=g(b2), out=out, skip=(w2p, g(bias_s), skip))
    plain = ops.conv(xv, wp, 3, Cout, mfma=True, pro=pro, bias=g(bias), bias2=g(b2))
    assert maxdiff(out.to_nchw(), plain.to_nchw()) <= 2e-5         # (the two launches may pick different tiles / K splits: another summation order)
    h = F.silu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None])
    ref = F.conv2d(h, w.double(), bias.double(), padding=1) + b2.double()[:, :, None, None]
    ref_s = F.conv2d(x.double(), w2.double(), bias_s.double())
    e1, e2 = maxdiff(out.to_nchw(), ref), maxdiff(skip.to_nchw(), ref_s)
    print(f'fused skip conv {B}x{H}x{W} {Cin}->{Cout}: 3x3 {e1:.2e}, 1x1 skip {e2:.2e} vs fp64')
    assert e1 <= 1e-4 and e2 <= 1e-4
    if ops.conv_prec_supported(xv, Cout, ops.PRO_AFFINE_SILU, ops.PREC_FP8X, skip=True):
        # the same launch under the fp16 + e4m3-cross-term plan (the skip conv's own products stay fp16 x 3: same bits as above)
        we = ops.fp8x_weight_exponent(w)
        out8, skip8 = ops.View.empty(B, H, W, Cout, DEV), ops.View.empty(B, H, W, Cout, DEV)
        ops.conv(xv, ops.pack_conv_weight(g(w), prec=ops.PREC_FP8X, w_exp=we), 3, Cout, mfma=True, pro=pro, bias=g(bias), bias2=g(b2), out=out8,
                 skip=(w2p, g(bias_s), skip8), prec=ops.PREC_FP8X, w_exp=we)
        e8 = maxdiff(out8.to_nchw(), ref)
        print(f'   under MUD_PREC_FP8X: 3x3 {e8:.2e} vs fp64')
        assert e8 <= 3e-4 and torch.equal(skip8.to_nchw(), skip.to_nchw())
    assert maxdiff(out.stats[..., 0], ref.sum(dim=(2, 3))) <= 2e-6 * float(ref.abs().sum(dim=(2, 3)).max())
    if ops.conv3x3_would_split_k(xv, Cout):                         # the fused launch split over K: the same bits whoever arrives last
        first, first_s = out.to_nchw().clone(), skip.to_nchw().clone()
        out2, skip2 = ops.View.empty(B, H, W, Cout, DEV), ops.View.empty(B, H, W, Cout, DEV)
        for _ in range(3):
            ops.conv(xv, wp, 3, Cout, mfma=True, pro=pro, bias=g(bias), bias2=g(b2), out=out2, skip=(w2p, g(bias_s), skip2))
            assert torch.equal(out2.to_nchw(), first) and torch.equal(skip2.to_nchw(), first_s)
        assert int(ops.splitk_counters(torch.device(DEV)).abs().sum()) == 0


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(1, 64, 64, 256, 256), (1, 64, 64, 512, 256), (1, 32, 32, 384, 128), (1, 40, 24, 320, 64), (4, 64, 64, 256, 256)])
def test_conv_split_k_small_grids(B, H, W, Cin, Cout):
    """Small grids (one slice at a time) deal the K chunks of a tile to several workgroups (raw partial slabs + a fixed-order
    reduce that applies the epilogue).  Checked against fp64 with every epilogue term on (bias, time bias, residual, scale,
    statistics), and against the unsplit kernel (no workspace passed -> never split)."""
    import ctypes as C_
    import mudiff_hip
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, generator=gen) / math.sqrt(Cin * 9)
    bias, b2 = torch.randn(Cout, generator=gen), torch.randn(B, Cout, generator=gen)
    res = torch.randn(B, Cout, H, W, generator=gen)
    sc, sh = torch.rand(B, Cin, generator=gen) + 0.5, torch.randn(B, Cin, generator=gen)
    xv, rv, wp = ops.View.from_nchw(g(x)), ops.View.from_nchw(g(res)), ops.pack_conv_weight(g(w))
    arena = ops.StatsArena(torch.device(DEV))
    out = ops.View.empty(B, H, W, Cout, DEV, arena)
    kw = dict(mfma=True, pro=(g(sc), g(sh), ops.PRO_AFFINE_SILU), bias=g(bias), bias2=g(b2), res=rv, out_scale=ops.INV_SQRT2)
    ops.PROFILE.enable()
    ops.conv(xv, wp, 3, Cout, out=out, **kw)
    ops.PROFILE.disable()
    y = out.to_nchw().cpu()
    # was this launch split?  (the query the wrapper used)
    a = mudiff_hip.ConvArgs()
    a.x, a.B, a.H, a.W, a.Cin, a.ldx, a.ks, a.stride, a.pad = xv.ptr, B, H, W, Cin, Cin, 3, 1, 1
    a.out, a.Cout, a.ldo, a.res, a.ldr = out.ptr, Cout, Cout, rv.ptr, Cout
    nws = mudiff_hip.load().mud_conv2d_mfma_splitk_bytes(C_.byref(a))
    print(f'split-K workspace for {B}x{H}x{W} {Cin}->{Cout}: {nws} bytes')
    if B == 1:
        assert nws > 0                                                  # every B=1 case of this table is a small grid with a long reduction
    h = F.silu(x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None])
    ref = (F.conv2d(h, w.double(), bias.double(), padding=1) + b2.double()[:, :, None, None] + res.double()) * ops.INV_SQRT2
    err = maxdiff(y, ref)
    print(f'  max-abs vs fp64 {err:.2e}')
    assert err <= 1e-4
    s_ref, q_ref = ref.sum(dim=(2, 3)), (ref * ref).sum(dim=(2, 3))
    assert maxdiff(out.stats[..., 0], s_ref) <= 2e-6 * float(ref.abs().sum(dim=(2, 3)).max()) and maxdiff(out.stats[..., 1], q_ref) <= 2e-6 * float(q_ref.max())
    if nws > 0:     # unsplit launch of the same problem through the raw C ABI (no workspace): same result to rounding
        out2 = ops.View.empty(B, H, W, Cout, DEV)
        a.w, a.pro_scale, a.pro_shift, a.pro_ld, a.pro_mode = C_.c_void_p(wp.data_ptr()), C_.c_void_p(g(sc).data_ptr()), C_.c_void_p(g(sh).data_ptr()), Cin, ops.PRO_AFFINE_SILU
        scd, shd, bd, b2d = g(sc), g(sh), g(bias), g(b2)
        a.pro_scale, a.pro_shift = C_.c_void_p(scd.data_ptr()), C_.c_void_p(shd.data_ptr())
        a.bias, a.bias2, a.bias2_ld, a.out_scale, a.out = C_.c_void_p(bd.data_ptr()), C_.c_void_p(b2d.data_ptr()), Cout, ops.INV_SQRT2, out2.ptr
        assert mudiff_hip.load().mud_conv2d_mfma(C_.byref(a), mudiff_hip.stream_ptr()) == 0
        torch.cuda.synchronize()
        assert maxdiff(out2.to_nchw(), y) <= 2e-5
        # the wrapper's launch reduced the slabs itself (arrival counters, one launch); the same problem with a workspace but WITHOUT
        # counters takes the two-launch path - same slabs, same order, same epilogue arithmetic: bit-identical outputs
        cnt = ops.splitk_counters(torch.device(DEV))
        assert int(cnt.abs().sum()) == 0                                # every launch leaves the counters at zero
        out3, ws = ops.View.empty(B, H, W, Cout, DEV), torch.empty(nws, device=DEV, dtype=torch.uint8)
        a.out, a.splitk_ws, a.splitk_ws_bytes = out3.ptr, C_.c_void_p(ws.data_ptr()), nws
        assert mudiff_hip.load().mud_conv2d_mfma(C_.byref(a), mudiff_hip.stream_ptr()) == 0
        torch.cuda.synchronize()
        assert torch.equal(out3.to_nchw().cpu(), y)
        out4 = ops.View.empty(B, H, W, Cout, DEV)
        for _ in range(3):                                              # whichever workgroup arrives last, the sum is the same
            ops.conv(xv, wp, 3, Cout, out=out4, **kw)
            assert torch.equal(out4.to_nchw().cpu(), y)
        assert int(cnt.abs().sum()) == 0


def test_fir_against_reference_golden():
    ops, S, L, UD, *_ = _imports()
    from utils.op import upfirdn2d
    gd = load_golden('fir.npz')
    for tag in 'abc':
        x = g(gd[f'{tag}.x'])
        assert maxdiff(UD.upsample_2d(x, (1, 3, 3, 1), factor=2), gd[f'{tag}.up']) < 1e-6
        assert maxdiff(UD.downsample_2d(x, (1, 3, 3, 1), factor=2), gd[f'{tag}.down']) < 1e-6
        assert maxdiff(UD.conv_downsample_2d(x, g(gd[f'{tag}.w']), k=(1, 3, 3, 1)), gd[f'{tag}.convdown']) < 3e-6
    for tag, (u, d, pad) in (('g1', (1, 1, (2, 1))), ('g2', (2, 1, (2, 1))), ('g3', (1, 2, (1, 1))), ('g4', (2, 2, (3, 0)))):
        assert maxdiff(upfirdn2d(g(gd['g.x']), g(gd['g.k']), up=u, down=d, pad=pad), gd[f'{tag}.out']) < 3e-6
    # NHWC form with the AdaGN+SiLU prologue and the dual output
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 10, 14, generator=gen)
    sc, sh = torch.randn(2, 12, generator=gen), torch.randn(2, 12, generator=gen)
    for mode, fn in (('up', O.upsample_2d), ('down', O.downsample_2d)):
        kk, up, down, pad = UD.fir_params(mode, (1, 3, 3, 1))
        oh, ox = ops.fir_nhwc(ops.View.from_nchw(g(x)), kk, up, down, pad, pro=(g(sc), g(sh), ops.PRO_AFFINE_SILU), want_h=True, want_x=True)
        assert maxdiff(ox.to_nchw(), fn(x)) < 1e-6
        assert maxdiff(oh.to_nchw(), fn(F.silu(x * sc[:, :, None, None] + sh[:, :, None, None]))) < 2e-6


def test_softmax_and_gates():
    ops, *_ = _imports()
    gen = torch.Generator().manual_seed(9)
    for rows, n in ((7, 64), (33, 4096), (5, 1000)):
        s = torch.randn(rows, n, generator=gen) * 3
        assert maxdiff(ops.softmax_rows_(g(s).clone(), n), F.softmax(s, dim=-1)) < 1e-6
    a, b, c = (torch.randn(2, 16, 6, 5, generator=gen) for _ in range(3))
    va, vb, vc = (ops.View.from_nchw(g(t)) for t in (a, b, c))
    assert maxdiff(ops.mul(va, vb).to_nchw(), a * b) < 1e-7
    arena = ops.StatsArena(torch.device(DEV))
    out = ops.View.empty(2, 6, 5, 16, DEV, arena)
    ref = a * b + (1 - a) * c
    assert maxdiff(ops.gate_mix(va, vb, vc, out).to_nchw(), ref) < 1e-6
    assert maxdiff(out.stats[..., 0], ref.sum(dim=(2, 3))) < 1e-4 and maxdiff(out.stats[..., 1], (ref * ref).sum(dim=(2, 3))) < 1e-4
    for C in (48, 192):          # channel counts that do not divide the block size
        a, b, c = (torch.randn(2, C, 9, 7, generator=gen) for _ in range(3))
        out = ops.View.empty(2, 9, 7, C, DEV, arena)
        ref = a * b + (1 - a) * c
        assert maxdiff(ops.gate_mix(*(ops.View.from_nchw(g(t)) for t in (a, b, c)), out).to_nchw(), ref) < 1e-6
        assert maxdiff(out.stats[..., 0], ref.sum(dim=(2, 3))) < 1e-4


@pytest.mark.parametrize('B,H,W,C', [(2, 8, 8, 16), (1, 6, 6, 32), (2, 16, 16, 64), (1, 64, 64, 256), (2, 12, 10, 128), (1, 8, 8, 48)])
def test_attention_block_fused_and_unfused(B, H, W, C):
    """AttnBlockpp on the fused flash kernel (C in 16/32/64/128/256) and on the unfused GEMM + softmax path (C = 48),
    against the oracle; key counts that are not multiples of the 32-key tile exercise the tail masking."""
    ops, S, L, *_ = _imports()
    gen = torch.Generator().manual_seed(C + H)
    m = L.AttnBlockpp(C, skip_rescale=True, init_scale=0.)
    sd = {}
    for k, v in m.state_dict().items():
        if v.dim() == 2:
            sd[k] = (torch.rand(v.shape, generator=gen) * 2 - 1) * math.sqrt(3.0 / v.shape[0])
        else:
            sd[k] = 0.1 * torch.randn(v.shape, generator=gen) + (1.0 if k.endswith('GroupNorm_0.weight') else 0.0)
    m.load_state_dict(sd)
    x = torch.randn(B, C, H, W, generator=gen) * 1.5
    ref = O.attn_block({'m.' + k: v for k, v in sd.items()}, 'm', x)
    err = maxdiff(m.to(DEV)(g(x)), ref)
    print(f'attention {B}x{H}x{W} C={C}: {err:.2e}')
    assert err < 5e-5


@pytest.mark.parametrize('B,N,C', [(1, 4096, 256), (1, 1440, 64), (3, 1000, 32), (1, 260, 128)])
def test_attention_key_split_matches_unsplit(B, N, C):
    """Few (batch x query-block) workgroups: the keys are split over several workgroups and merged (mud_attention with a
    workspace).  Same result as the unsplit kernel (ws = NULL) up to the merge's rounding, and as fp64 softmax attention."""
    import mudiff_hip
    ops, *_ = _imports()
    lib = mudiff_hip.load()
    gen = torch.Generator().manual_seed(N + C)
    qkv = g(torch.randn(B, N, 3 * C, generator=gen))
    nws = lib.mud_attention_ws_bytes(B, N, C)
    assert nws > 0
    v = ops.View(qkv, B, 1, N, 3 * C)
    split = ops.attention(v, C, C ** -0.5).tensor().reshape(B, N, C)
    unsplit = torch.empty(B, N, C, device=DEV)
    rc = lib.mud_attention(qkv.data_ptr(), B, N, C, 3 * C, C ** -0.5, unsplit.data_ptr(), C, None, None)
    assert rc == 0
    q, k, vv = (t.double().cpu() for t in qkv.split(C, dim=2))
    ref = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1) @ vv
    print(f'attention split B={B} N={N} C={C}: split-unsplit {maxdiff(split, unsplit):.2e}, split-fp64 {maxdiff(split, ref):.2e}')
    assert maxdiff(split, unsplit) <= 2e-6 and maxdiff(split, ref) <= 2e-5
    assert lib.mud_attention_ws_bytes(16, 4096, 256) == 0          # enough workgroups: no split, no workspace


def test_blocks_against_reference_golden():
    ops, S, L, UD, *_ = _imports()
    import torch.nn as nn
    gd = load_golden('blocks.npz')
    act = nn.SiLU()

    def load(mod, prefix):
        sd = {k[len(prefix) + 1:]: v for k, v in gd.items() if k.startswith(prefix + '.')}
        mod.load_state_dict(sd, strict=True)
        return mod.to(DEV)
    zemb, temb = g(gd['zemb']), g(gd['temb'])
    zd, td = zemb.shape[1], temb.shape[1]
    for tag, cin, cout, up, down in (('plain', 8, 8, 0, 0), ('skip', 8, 16, 0, 0), ('up', 12, 12, 1, 0), ('down', 8, 8, 0, 1), ('cat', 24, 16, 0, 0)):
        m = load(L.ResnetBlockBigGANpp_Adagn(act, cin, cout, temb_dim=td, zemb_dim=zd, up=bool(up), down=bool(down), dropout=0.0, fir=True,
                                             fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.), f'res_{tag}.sd')
        err = maxdiff(m(g(gd[f'res_{tag}.x']), temb, zemb), gd[f'res_{tag}.y'])
        print(f'resblock.{tag}: {err:.2e}')
        assert err < 5e-5, tag
    m = load(L.AdaptiveGroupNorm(4, 16, zd), 'adagn.sd')
    assert maxdiff(m(g(gd['adagn.x']), zemb), gd['adagn.y']) < 1e-5
    for tag, c in (('c16', 16), ('c32', 32)):
        m = load(L.AttnBlockpp(c, skip_rescale=True, init_scale=0.), f'attn_{tag}.sd')
        err = maxdiff(m(g(gd[f'attn_{tag}.x'])), gd[f'attn_{tag}.y'])
        print(f'attn.{tag}: {err:.2e}')
        assert err < 5e-5
    x1 = g(gd['feat.x'])
    assert maxdiff(load(L.ConvFeatBlock(act, in_ch=1, out_ch=16), 'feat.sd')(x1), gd['feat.y']) < 5e-5
    assert maxdiff(load(L.ConvBlock(act, in_ch=1, out_ch=16, zemb_dim=zd), 'ada.sd')(x1, zemb), gd['ada.y']) < 5e-5
    assert maxdiff(load(L.ConvBlock_GAP(act, in_ch=1, out_ch=16, zemb_dim=zd), 'gap.sd')(x1), gd['gap.y']) < 5e-5
    for tag, cin, cout in (('p1', 1, 8), ('p8', 8, 16)):
        m = load(L.Downsample(in_ch=cin, out_ch=cout, with_conv=True, fir=True, fir_kernel=(1, 3, 3, 1)), f'pyr_{tag}.sd')
        assert maxdiff(m(g(gd[f'pyr_{tag}.x'])), gd[f'pyr_{tag}.y']) < 1e-5


@pytest.mark.parametrize('tag,nc,ngf,td', [('d8', 2, 8, 32), ('d16', 2, 16, 64), ('d8b8', 2, 8, 32)])
def test_critic_against_reference_golden(tag, nc, ngf, td):
    """Discriminator_large(x, t, x_t) -> (logit, mid_feat) on the HIP kernels vs the reference."""
    import torch.nn as nn
    from backbones.discriminator import Discriminator_large
    gd = load_golden('critic.npz')
    d = Discriminator_large(nc=nc, ngf=ngf, t_emb_dim=td, act=nn.LeakyReLU(0.2))
    sd = O.make_discriminator_state_dict(nc, ngf, td, 1234)
    assert list(d.state_dict().keys()) == list(sd.keys())
    d.load_state_dict(sd)
    logit, mid = d.to(DEV)(g(gd[f'{tag}.x']), g(gd[f'{tag}.t']), g(gd[f'{tag}.xt']))
    e1, e2 = maxdiff(logit, gd[f'{tag}.logit']), maxdiff(mid, gd[f'{tag}.mid'])
    print(f'critic {tag}: logit {e1:.2e} (|logit| ~ {float(gd[f"{tag}.logit"].abs().mean()):.2f}), mid_feat {e2:.2e}')
    assert e1 <= 1e-3 and e2 <= 1e-4


def _build(cfg, seed=1234):
    *_, NCSNpp, NCSNpp_adaptive = _imports()
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    g1.load_state_dict(O.make_state_dict(cfg, 'g1', seed))
    g2.load_state_dict(O.make_state_dict(cfg, 'g2', seed))
    return g1.to(DEV).eval(), g2.to(DEV).eval()


@pytest.mark.parametrize('tag', list(SMALL_CFGS))
def test_small_models_every_step_vs_reference(tag):
    ops, S, *_ = _imports()
    gd = load_golden('small_models.npz')
    cfg = O.default_config(**SMALL_CFGS[tag])
    g1, g2 = _build(cfg)
    conds = [g(c) for c in small_conds(cfg)]
    x_init, zs, noises = sampler_inputs(cfg, 2)
    coef = S.Posterior_Coefficients(cfg, DEV)
    x, steps = S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], cfg.num_timesteps, g(x_init), None, cfg,
                                   zs=[g(z) for z in zs], noises=[g(n) for n in noises], return_steps=True)
    worst = 0.0
    for k, st in enumerate(steps):
        for nm, v in zip(('x01', 'x02', 'xnew'), st):
            worst = max(worst, maxdiff(v, gd[f'{tag}.step{k}.{nm}']))
    print(f'{tag}: worst per-step max-abs vs reference = {worst:.2e}')
    assert worst <= 1e-3


@pytest.mark.parametrize('tag', ['s32', 's16t8'])
def test_reference_shaped_loop_under_autocast_vs_reference(tag):
    """Drop-in under the reference's OWN calling convention (engine/test.py:180-199): plain per-step generator calls written
    here exactly like the reference's loop - inside torch.autocast (the reference wraps the generators in
    torch.cuda.amp.autocast(), fp16 on a GPU), `x_0_1[:, [0], :]` advanced-index copies as pseudo-target and posterior
    inputs, `.detach()` - against the outputs of the reference's fp32 CPU run.  The modules must ignore the ambient
    autocast: fp32 out, <= 1e-3 per step."""
    ops, S, *_ = _imports()
    gd = load_golden('small_models.npz')
    cfg = O.default_config(**SMALL_CFGS[tag])
    generator1, generator2 = _build(cfg)
    cond1, cond2, cond3 = (g(c) for c in small_conds(cfg))
    x_init, zs, noises = sampler_inputs(cfg, 2)
    coefficients = S.Posterior_Coefficients(cfg, DEV)
    n_time = cfg.num_timesteps
    x = g(x_init)
    worst = 0.0
    with torch.no_grad():
        for k, i in enumerate(reversed(range(n_time))):
            t = torch.full((x.size(0),), i, dtype=torch.int64).to(x.device)
            latent_z = g(zs[k])
            with torch.autocast('cuda', dtype=torch.float16):
                x_0_1 = generator1(x, cond1, cond2, cond3, t, latent_z)
                x_0_2 = generator2(x, cond1, cond2, cond3, t, latent_z, x_0_1[:, [0], :])
                x_new = S.sample_posterior_combine(coefficients, x_0_1[:, [0], :], x_0_2[:, [0], :], x, t, noise=g(noises[k]))
            assert x_0_1.dtype == x_0_2.dtype == x_new.dtype == torch.float32
            for nm, v in zip(('x01', 'x02', 'xnew'), (x_0_1, x_0_2, x_new)):
                worst = max(worst, maxdiff(v, gd[f'{tag}.step{k}.{nm}']))
            x = x_new.detach()
    print(f'{tag} reference-shaped loop under fp16 autocast: worst per-step max-abs vs reference = {worst:.2e}')
    assert worst <= 1e-3


def test_graph_sampler_two_channel_images_use_channel_zero():
    """num_channels > 1: the reference loop feeds x_0_1[:, [0], :] to G2 and to the posterior (engine/test.py:193-195), which only
    type-checks for 1-channel x_t; GraphSampler must slice the same way instead of passing the whole tensor (G2 then rejects the
    2-channel pseudo-target exactly like the reference's conv would).  Checked on G1 alone through the eager loop's contract:
    a 2-channel G1 output sliced to channel 0 equals the oracle's."""
    ops, S, *_ = _imports()
    from helpers import VARIANT_BASE, VARIANTS
    cfg = O.default_config(**{**VARIANT_BASE, **VARIANTS['two_channels']})
    g1, _ = _build(cfg)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 2, 32, 32, generator=gen)
    cs = [torch.tanh(torch.randn(2, 2, 32, 32, generator=gen)) for _ in range(3)]
    z = torch.randn(2, cfg.nz, generator=gen)
    t = torch.full((2,), 1, dtype=torch.int64)
    out = g1(g(x), *(g(c) for c in cs), g(t), g(z))
    ref = O.g1_forward(O.make_state_dict(cfg, 'g1', 1234), cfg, x, *cs, t, z)
    assert out.shape == (2, 2, 32, 32) and maxdiff(out[:, [0], :], ref[:, [0], :]) <= 1e-3


def test_sampler_rejects_half_injected_draws():
    ops, S, *_ = _imports()
    cfg = O.default_config(**SMALL_CFGS['s32na'])
    g1, g2 = _build(cfg)
    smp = S.GraphSampler(S.Posterior_Coefficients(cfg, DEV), g1, g2, cfg, 1, 32, 32, DEV)
    c = torch.zeros(1, 1, 32, 32, device=DEV)
    with pytest.raises(ValueError):
        smp.sample(c, c, c, c, cfg.num_timesteps, zs=[torch.zeros(1, cfg.nz, device=DEV)] * cfg.num_timesteps)


def test_operands_on_different_gpus_are_rejected_and_foreign_device_launches_are_guarded():
    """A launch goes to the device its operands live on, whatever the process's current device is; operands on two GPUs
    are refused (the kernels take raw pointers).  With one visible GPU only the refusal of a CPU operand and the
    same-device path can be exercised; with two, the cross-device cases run as well."""
    import mudiff_hip
    ops, S, *_ = _imports()
    with pytest.raises(mudiff_hip.MudiffHipError):
        ops.pixel_norm(torch.zeros(2, 8))
    if torch.cuda.device_count() < 2:
        return
    a, b = torch.randn(4, 16, device='cuda:0'), torch.randn(4, 16, device='cuda:1')
    with pytest.raises(mudiff_hip.MudiffHipError):
        mudiff_hip.require_gpu(a, b)
    with torch.cuda.device(0):                      # current device 0, operands on device 1
        out = ops.pixel_norm(b)
    ref = b.cpu() * torch.rsqrt((b.cpu() ** 2).mean(1, keepdim=True) + 1e-8)
    assert out.device == b.device and maxdiff(out, ref) <= 1e-6


def test_config2_full_size_every_step_vs_reference():
    """BASELINE config 2: 256x256, nf=64, ch_mult 1-2-4, 4 steps, dual generator, injected noise."""
    ops, S, *_ = _imports()
    gd = load_golden('full_cfg2.npz')
    cfg = O.default_config()
    g1, g2 = _build(cfg)
    conds = [g(c) for c in demo_conds()]
    x_init, zs, noises = sampler_inputs(cfg, 1)
    coef = S.Posterior_Coefficients(cfg, DEV)
    x, steps = S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], 4, g(x_init), None, cfg,
                                   zs=[g(z) for z in zs], noises=[g(n) for n in noises], return_steps=True)
    for k, st in enumerate(steps):
        errs = [maxdiff(v, gd[f'step{k}.{nm}']) for nm, v in zip(('x01', 'x02', 'xnew'), st)]
        print(f'cfg2 step {k}: max-abs x01 {errs[0]:.2e} x02 {errs[1]:.2e} xnew {errs[2]:.2e}')
        assert max(errs) <= 1e-3
    # PSNR / SSIM of the final sample against the demo target, build vs reference: within +-0.05 dB / +-0.001
    from helpers import preprocess_demo
    tgt = preprocess_demo(load_golden('demo_inputs_u8.npz')['t1ce'].numpy())[0, 0].numpy()
    to01 = lambda a: (np.asarray(a, np.float64) + 1) / 2
    ours, ref = x.cpu()[0, 0].numpy(), gd['step3.xnew'][0, 0].numpy()
    dp = O.psnr(to01(tgt), to01(ours)) - O.psnr(to01(tgt), to01(ref))
    ds = O.ssim(to01(tgt), to01(ours)) - O.ssim(to01(tgt), to01(ref))
    print(f'cfg2: dPSNR {dp:+.4f} dB, dSSIM {ds:+.5f}')
    assert abs(dp) <= 0.05 and abs(ds) <= 0.001


def test_config5_isles_shaped_8_steps_vs_reference():
    """BASELINE config 5: 256x256, 8 timesteps, ch_mult 1-1-2-2-4 so that attn_resolutions=16
    fires in the down and up paths (N=256 keys, C=256) as well as in the middle; golden made by the reference itself."""
    ops, S, *_ = _imports()
    gd = load_golden('full_cfg5.npz')
    cfg = O.default_config(ch_mult=[1, 1, 2, 2, 4], num_timesteps=8, attn_resolutions=(16,))
    g1, g2 = _build(cfg)
    conds = [g(c) for c in demo_conds()]
    x_init, zs, noises = sampler_inputs(cfg, 1)
    coef = S.Posterior_Coefficients(cfg, DEV)
    x, steps = S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], 8, g(x_init), None, cfg,
                                   zs=[g(z) for z in zs], noises=[g(n) for n in noises], return_steps=True)
    for k, st in enumerate(steps):
        errs = {nm: maxdiff(v, gd[f'step{k}.{nm}']) for nm, v in zip(('x01', 'x02', 'xnew'), st) if f'step{k}.{nm}' in gd}
        print(f'cfg5 step {k}: max-abs ' + ' '.join(f'{nm} {e:.2e}' for nm, e in errs.items()))
        assert max(errs.values()) <= 1e-3


@pytest.mark.parametrize('H,W,B,kw', [(40, 40, 3, dict(num_channels_dae=32, ch_mult=[1, 2, 4], attn_resolutions=(10,))),
                                       (24, 56, 2, dict(num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(), num_res_blocks=1)),
                                       (240, 240, 1, dict(num_channels_dae=32, ch_mult=[1, 2, 4], attn_resolutions=(60,), num_res_blocks=1))])
def test_ragged_sizes_against_oracle(H, W, B, kw):
    """Image sizes that are not multiples of the 8x32 conv tile / 32-key attention tile / 128-query block (240x240 is
    what the reference trains on), odd batch sizes, non-square slices: one G1 + G2 + posterior step vs the oracle."""
    ops, S, *_ = _imports()
    cfg = O.default_config(image_size=H, **kw)
    g1, g2 = _build(cfg, seed=5)
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 5), O.make_state_dict(cfg, 'g2', 5)
    gen = torch.Generator().manual_seed(H * 7 + W)
    x, c1, c2, c3 = (torch.tanh(torch.randn(B, 1, H, W, generator=gen)) for _ in range(4))
    z, noise = torch.randn(B, cfg.nz, generator=gen), torch.randn(B, 1, H, W, generator=gen)
    t = torch.randint(0, cfg.num_timesteps, (B,), generator=gen)
    y1 = g1(g(x), g(c1), g(c2), g(c3), g(t), g(z))
    y2 = g2(g(x), g(c1), g(c2), g(c3), g(t), g(z), y1)
    xn = S.sample_posterior_combine(S.Posterior_Coefficients(cfg, DEV), y1, y2, g(x), g(t), g(noise))
    r1 = O.g1_forward(sd1, cfg, x, c1, c2, c3, t, z)
    r2 = O.g2_forward(sd2, cfg, x, c1, c2, c3, t, z, r1)
    rn = O.sample_posterior_combine(O.PosteriorCoefficients(cfg), r1, r2, x, t, noise)
    errs = (maxdiff(y1, r1), maxdiff(y2, r2), maxdiff(xn, rn))
    print(f'{H}x{W} B={B}: {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}')
    assert max(errs) <= 1e-3


def test_graph_sampler_matches_eager_and_batches():
    ops, S, *_ = _imports()
    cfg = O.default_config(**SMALL_CFGS['s32'])
    g1, g2 = _build(cfg)
    B = 3
    gen = torch.Generator().manual_seed(77)
    conds = [g(torch.tanh(torch.randn(B, 1, 32, 32, generator=gen))) for _ in range(3)]
    x_init = g(torch.randn(B, 1, 32, 32, generator=gen))
    zs = [g(torch.randn(B, cfg.nz, generator=gen)) for _ in range(4)]
    noises = [g(torch.randn(B, 1, 32, 32, generator=gen)) for _ in range(4)]
    coef = S.Posterior_Coefficients(cfg, DEV)
    eager = S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], 4, x_init, None, cfg, zs=zs, noises=noises)
    gs = S.GraphSampler(coef, g1, g2, cfg, B, 32, 32, DEV)
    graphed = gs.sample(conds[0], conds[1], conds[2], x_init, 4, zs=zs, noises=noises)
    # fp64 atomics in the GroupNorm statistics make the last bit of a scale order-dependent; the fp16 hi/lo split
    # turns such a 1-ulp input change into a ~2^-17 local change, so two runs agree to ~1e-5, not bitwise
    assert maxdiff(eager, graphed) < 1e-4
    # slices are independent: sample 1 alone == sample 1 inside the batch (data-parallel sharding is exact)
    solo = S.sample_from_model(coef, g1, conds[0][1:2], g2, conds[1][1:2], conds[2][1:2], 4, x_init[1:2], None, cfg,
                               zs=[z[1:2] for z in zs], noises=[n[1:2] for n in noises])
    assert maxdiff(solo, eager[1:2]) < 1e-4    # same arithmetic per slice; only the grouping of the statistics partial sums differs
    # oracle on the same batch
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    ref = O.sample_from_model(O.PosteriorCoefficients(cfg), sd1, sd2, cfg, *[c.cpu() for c in conds], x_init.cpu(),
                              [z.cpu() for z in zs], [n.cpu() for n in noises])
    assert maxdiff(eager, ref) <= 1e-3


def test_two_samplers_replayed_concurrently_have_their_own_splitk_counters():
    """a captured graph can be replayed on any stream; two one-slice samplers (whose 64x64 convolutions split over K and
    reduce in-launch through arrival counters) replayed side by side on two streams must not share a counter array."""
    ops, S, *_ = _imports()
    cfg = O.default_config()
    g1, g2 = _build(cfg)
    coef = S.Posterior_Coefficients(cfg, DEV)
    sa, sb = (S.GraphSampler(coef, g1, g2, cfg, 1, 256, 256, DEV) for _ in range(2))
    assert sa._splitk.data_ptr() != sb._splitk.data_ptr()
    conds = [g(c) for c in demo_conds()]
    xa, zs_a, ns_a = sampler_inputs(cfg, 1, seed_x=42)
    xb, zs_b, ns_b = sampler_inputs(cfg, 1, seed_x=77)
    run = lambda sm, x, zs, ns: sm.sample(*conds, g(x), 2, zs=[g(z) for z in zs[:2]], noises=[g(n) for n in ns[:2]])
    ref_a, ref_b = run(sa, xa, zs_a, ns_a).clone(), run(sb, xb, zs_b, ns_b).clone()
    torch.cuda.synchronize()
    st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(st_a):
            out_a = run(sa, xa, zs_a, ns_a)
        with torch.cuda.stream(st_b):
            out_b = run(sb, xb, zs_b, ns_b)
        torch.cuda.synchronize()
        # (run-to-run the outputs agree to ~3e-5 here - the GroupNorm sums are fp64 atomics whose order varies; a tile reduced early or never
        #  would be off by orders of magnitude more)
        assert maxdiff(out_a, ref_a) <= 1e-4 and maxdiff(out_b, ref_b) <= 1e-4
    assert int(sa._splitk.abs().sum()) == 0 and int(sb._splitk.abs().sum()) == 0      # every launch leaves its counters at zero


def test_loop_cache_changes_nothing():
    """sample_from_model / GraphSampler hoist what depends on the condition images alone out of the reverse loop
    (begin_loop_cache): same images as plain per-step forward calls (which never cache), and a second sampling run with OTHER
    conditions must not see the first run's cache."""
    ops, S, *_ = _imports()
    cfg = O.default_config(**SMALL_CFGS['s32'])
    g1, g2 = _build(cfg)
    coef = S.Posterior_Coefficients(cfg, DEV)
    x_init, zs, noises = sampler_inputs(cfg, 2)
    zs, noises = [g(z) for z in zs], [g(n) for n in noises]
    for seed in (1, 2):
        gen = torch.Generator().manual_seed(seed)
        conds = [g(torch.tanh(torch.randn(2, 1, 32, 32, generator=gen))) for _ in range(3)]
        cached = S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], 4, g(x_init), None, cfg, zs=zs, noises=noises)
        assert g1._loop_cache is None and g2._loop_cache is None
        x = g(x_init)
        for k, i in enumerate(reversed(range(4))):
            t = torch.full((2,), i, dtype=torch.int64, device=DEV)
            y1 = g1(x, *conds, t, zs[k])
            y2 = g2(x, *conds, t, zs[k], y1)
            x = S.sample_posterior_combine(coef, y1, y2, x, t, noises[k])
        assert maxdiff(cached, x) <= 1e-4
    sampler = S.GraphSampler(coef, g1, g2, cfg, 2, 32, 32, DEV)
    assert maxdiff(sampler.sample(*conds, g(x_init), 4, zs=zs, noises=noises), x) <= 1e-4
    other = [torch.flip(c, dims=(3,)) for c in conds]
    want = S.sample_from_model(coef, g1, other[0], g2, other[1], other[2], 4, g(x_init), None, cfg, zs=zs, noises=noises)
    assert maxdiff(sampler.sample(*other, g(x_init), 4, zs=zs, noises=noises), want) <= 1e-4


def test_cpu_tensors_fail_loudly():
    *_, NCSNpp, _ = _imports()
    import mudiff_hip
    cfg = O.default_config(**SMALL_CFGS['s32na'])
    m = NCSNpp(cfg)
    z = torch.zeros(1, 1, 32, 32)
    with pytest.raises(mudiff_hip.MudiffHipError):
        m(z, z, z, z, torch.zeros(1, dtype=torch.int64), torch.zeros(1, cfg.nz))


def test_c_abi_rejects_bad_arguments_without_launching():
    """The C entry points validate on the host before any launch (a faulting kernel can reset the device): null
    pointers, misaligned / too-narrow views, unsupported sizes return MUD_ERR_* with a message in mud_last_error(), and
    the Python layer turns them into MudiffHipError (a RuntimeError, like the reference's TORCH_CHECK failures)."""
    import ctypes
    import mudiff_hip
    ops, *_ = _imports()
    lib = mudiff_hip.load()
    x = torch.zeros(1, 8, 8, 16, device=DEV)
    out = torch.zeros(1, 8, 8, 16, device=DEV)
    k = torch.ones(4, 4, device=DEV)
    # upfirdn2d: null input, zero-sized output, oversized filter
    assert lib.mud_upfirdn2d(None, 1, 8, 8, k.data_ptr(), 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, out.data_ptr(), None) == 1
    assert b'null' in lib.mud_last_error()
    assert lib.mud_upfirdn2d(x.data_ptr(), 1, 2, 2, k.data_ptr(), 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, out.data_ptr(), None) == 1
    assert lib.mud_upfirdn2d(x.data_ptr(), 1, 8, 8, k.data_ptr(), 9, 9, 1, 1, 1, 1, 4, 4, 4, 4, out.data_ptr(), None) == 1
    # attention: unsupported head dim is reported, not launched
    assert lib.mud_attention_supported(24) == 0 and lib.mud_attention_supported(256) == 1
    assert lib.mud_attention(x.data_ptr(), 1, 64, 24, 72, ctypes.c_float(1.0), out.data_ptr(), 24, None, None) != 0
    # conv: channel pitch smaller than the channel count, misaligned base pointer
    xv = ops.View(x, 1, 8, 8, 16)
    w = ops.pack_conv_weight(torch.zeros(64, 16, 3, 3, device=DEV))
    bad = ops.View(x, 1, 8, 8, 16)
    bad.ld = 8
    with pytest.raises(mudiff_hip.MudiffHipError):
        ops.conv(bad, w, 3, 64, mfma=True)
    with pytest.raises(mudiff_hip.MudiffHipError):
        ops.fir_nhwc(ops.View(torch.zeros(1, 4, 4, 6, device=DEV), 1, 4, 4, 6), [[1.0]], 1, 1, (0, 0))      # C % 4 != 0
    with pytest.raises(mudiff_hip.MudiffHipError):
        ops.resize_bilinear(torch.zeros(1, 1, 4, 4, device=DEV), (0, 4))
    torch.cuda.synchronize()
    assert float(ops.conv(xv, w, 3, 64, mfma=True).tensor().abs().max()) == 0.0      # the library still works afterwards


# ----------------------------------------------------------------------------------------------
# rows f1 (uncertainty map) and f3 (volume pipeline)
# ----------------------------------------------------------------------------------------------
def test_resize_bilinear_and_range_kernels():
    """F.interpolate(mode='bilinear', align_corners=False) as the reference calls it; fp32 separable weights in the same
    order, products un-contracted: <= 1e-6 of the reference's outputs (bit-exact in practice)."""
    ops, *_ = _imports()
    gd = load_golden('volume.npz')
    for tag in ('down', 'up', 'x8', 'brats', 'same'):
        want = gd[f'resize.{tag}.out']
        got = ops.resize_bilinear(g(gd[f'resize.{tag}.in']), want.shape[-2:])
        assert got.shape == want.shape and maxdiff(got, want) <= 1e-6, tag
    x = torch.randn(3, 1, 17, 5) * 1.5
    assert torch.equal(ops.to_range_0_1(g(x)).cpu(), ((x + 1.0) / 2.0).clamp(0.0, 1.0))
    assert ops.resize_bilinear(g(torch.zeros(0, 1, 4, 4)), (8, 8)).shape == (0, 1, 8, 8)


def test_uncertainty_map_vs_reference():
    from backbones.discriminator import uncertainty_map, conv2d
    gd = load_golden('volume.npz')
    att = conv2d(64 * 8, 1, 1, padding=0)
    att.weight.data.copy_(gd['att.w']); att.bias.data.copy_(gd['att.b'])
    att = att.to(DEV)
    out = uncertainty_map(att, g(gd['att.feat']), (64, 64))
    assert out.shape == gd['att.out'].shape and maxdiff(out, gd['att.out']) <= 1e-4
    att.weight.mul_(2.0)                                     # in-place update (as an optimizer does it) invalidates the packed copy
    out2 = uncertainty_map(att, g(gd['att.feat']), (64, 64))
    assert maxdiff(out2, O.uncertainty_map(gd['att.feat'], 2 * gd['att.w'], gd['att.b'], (64, 64))) <= 1e-4


def _volume_case():
    gd = load_golden('volume.npz')
    cfg = O.default_config(image_size=16, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(8,), num_res_blocks=1)
    vols = [gd[f'vol{m}'].numpy() for m in range(3)]          # [20,24,11] float64: resized to 16x16 on the way in
    half = 2                                                  # slices 3..7 -> n = 5
    n, T, S = 5, cfg.num_timesteps, cfg.image_size
    gen = torch.Generator().manual_seed(77)
    x_inits = torch.randn(n, 1, S, S, generator=gen)
    zs = [torch.randn(n, cfg.nz, generator=gen) for _ in range(T)]
    noises = [torch.randn(n, 1, S, S, generator=gen) for _ in range(T)]
    return cfg, vols, half, x_inits, zs, noises


@pytest.mark.parametrize('use_graph', [False, True])
def test_volume_slices_batched_vs_per_slice_oracle(use_graph):
    """Row f3: all slices of a volume batched (batch 2 -> 3 launches, the last one padded) through the HIP sampler vs the
    oracle's per-slice restatement of engine/test_volume.py:262-283 with the same per-slice Gaussian draws."""
    from mudiff_hip import volume as V
    cfg, vols, half, x_inits, zs, noises = _volume_case()
    g1, g2 = _build(cfg, seed=9)
    stacks = [np.stack(V.extract_center_slices(V.robust_minmax_to_minus1_1(v), half)[0], 0) for v in vols]
    got = V.predict_slices(cfg, g1, g2, stacks, DEV, batch_size=2, x_inits=x_inits, zs=zs, noises=noises, use_graph=use_graph)
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 9), O.make_state_dict(cfg, 'g2', 9)
    want, s0, s1 = O.predict_volume(O.PosteriorCoefficients(cfg), sd1, sd2, cfg, vols, half,
                                    [x_inits[i:i + 1] for i in range(5)],
                                    [[z[i:i + 1] for z in zs] for i in range(5)], [[e[i:i + 1] for e in noises] for i in range(5)])
    assert (s0, s1) == (3, 7) and got.shape == (5, 16, 16)
    err = max(float(np.abs(got[i] - want[i]).max()) for i in range(5))
    print(f'volume slices (graph={use_graph}): max-abs {err:.2e}')
    assert err <= 1e-3 and got.min() >= 0.0 and got.max() <= 1.0
    if use_graph:       # one sampler reused across volumes (capture paid once): what a sampler built per call gives (to the fp64-atomic GroupNorm sums' run-to-run jitter)
        from mudiff_hip import sampling as S
        smp = S.GraphSampler(S.Posterior_Coefficients(cfg, DEV), g1, g2, cfg, 2, cfg.image_size, cfg.image_size, DEV)
        for _ in range(2):
            again = V.predict_slices(cfg, g1, g2, stacks, DEV, batch_size=32, x_inits=x_inits, zs=zs, noises=noises, sampler=smp)
            assert float(np.abs(again - got).max()) <= 1e-4
        other, _ = _build(cfg, seed=10)
        with pytest.raises(ValueError):
            V.predict_slices(cfg, other, g2, stacks, DEV, x_inits=x_inits, zs=zs, noises=noises, sampler=smp)
        with pytest.raises(ValueError):      # a captured sampler cannot be combined with use_graph=False (used to be ignored silently)
            V.predict_slices(cfg, g1, g2, stacks, DEV, x_inits=x_inits, zs=zs, noises=noises, sampler=smp, use_graph=False)
        # a SEEDED run draws per slice by global index: the same seed gives the same volume whatever the batching
        # (batch 2 through the reused sampler, batch 5 and batch 3 through samplers of their own)
        seeded = [V.predict_slices(cfg, g1, g2, stacks, DEV, batch_size=bs, seed=123, sampler=sm) for bs, sm in ((2, smp), (5, None), (3, None))]
        assert max(float(np.abs(v - seeded[0]).max()) for v in seeded[1:]) <= 1e-4
        assert float(np.abs(V.predict_slices(cfg, g1, g2, stacks, DEV, batch_size=2, seed=124, sampler=smp) - seeded[0]).max()) > 1e-3


def test_predict_volume_end_to_end_nifti(tmp_path):
    """The CLI entry point on files: three NIfTI inputs + DDP-style checkpoints ('module.' prefix) -> predicted_t1ce.nii.gz
    with the inputs' geometry, zero outside the centre window, [0,1] inside, reproducible for a fixed --seed; in-plane size
    != image_size is refused unless --resize_back."""
    from mudiff_hip import volume as V
    cfg = O.default_config(image_size=16, num_channels_dae=16, ch_mult=[1, 2], attn_resolutions=(4,), num_res_blocks=1)
    exp = tmp_path / 'results' / 'exp0'
    exp.mkdir(parents=True)
    for which, name in (('g1', 'gen_diffusive_1'), ('g2', 'gen_diffusive_2')):
        sd = O.make_state_dict(cfg, which, 9)
        torch.save({'module.' + k: v for k, v in sd.items()}, str(exp / f'{name}.pth'))
    rng = np.random.default_rng(0)
    aff = np.diag([1.0, 1.0, 2.5, 1.0]); aff[:3, 3] = (-8, -8, 3)
    paths = {}
    for m in ('flair', 't2', 't1'):
        v = (100 + 50 * rng.random((16, 16, 9))) * (rng.random((16, 16, 9)) > 0.2)
        paths[m] = str(tmp_path / f'{m}.nii.gz')
        V.write_nifti(paths[m], v.astype(np.float32), aff)
    common = ['--target_modality', 'T1CE', '--exp', 'exp0', '--output_path', str(tmp_path / 'results'), '--image_size', '16',
              '--num_channels_dae', '16', '--ch_mult', '1', '2', '--attn_resolutions', '4', '--num_res_blocks', '1',
              '--slice_half_range', '2', '--batch_size', '4', '--input_flair', paths['flair'], '--input_t2', paths['t2'],
              '--input_t1', paths['t1']]
    outs = []
    for run in range(2):
        out = V.predict_volume(V.build_argparser(common + ['--output_dir', str(tmp_path / f'out{run}')]))
        data, a, _ = V.read_nifti(out)
        assert out.endswith('predicted_t1ce.nii.gz') and data.shape == (16, 16, 9) and np.allclose(a, aff)
        assert not data[:, :, :2].any() and not data[:, :, 7:].any() and data[:, :, 2:7].any()
        assert data.min() >= 0.0 and data.max() <= 1.0
        outs.append(data)
    assert np.abs(outs[0] - outs[1]).max() <= 1e-4      # same draws; GroupNorm sums are fp64 atomics, so not bit-for-bit
    with pytest.raises(ValueError):                                                     # missing modality
        V.predict_volume(V.build_argparser(common[:-2] + ['--output_dir', str(tmp_path / 'o')]))
    big = common + ['--output_dir', str(tmp_path / 'o2')]
    big[big.index('--image_size') + 1] = '32'
    with pytest.raises(ValueError):
        V.predict_volume(V.build_argparser(big))
    out = V.predict_volume(V.build_argparser(big + ['--resize_back']))
    assert V.read_nifti(out)[0].shape == (16, 16, 9)


# ----------------------------------------------------------------------------------------------
# row f4: alternate configurations
# ----------------------------------------------------------------------------------------------
from helpers import VARIANT_BASE, VARIANTS      # noqa: E402


@pytest.mark.parametrize('name', list(VARIANTS))
def test_variant_generators_vs_reference(name):
    """Every configuration the reference can construct and run (output_skip / input_skip sum|cat / no input pyramid /
    fir=False resamplers / Fourier time embedding / unconditional / no skip rescale / [0,1] inputs without tanh / two image
    channels / the two-condition twins): G1 and G2 forward against the reference's own outputs."""
    gd = load_golden('variants.npz')
    cfg = O.default_config(**{**VARIANT_BASE, **VARIANTS[name]})
    if name == 'healthy':
        from backbones import ncsnpp_generator_adagn_feat_healthy as H
        G1, G2, nc = H.NCSNpp, H.NCSNpp_adaptive, 2
    else:
        *_, G1, G2 = _imports()
        nc = 3
    x, c1, c2, t, z = (g(gd[f'{name}.{k}']) for k in ('x', 'c1', 'c2', 't', 'z'))
    conds = [c1, c2] + ([g(gd[f'{name}.c3'])] if nc == 3 else [])
    m1 = G1(cfg)
    m1.load_state_dict(O.make_state_dict(cfg, 'g1', 77, n_cond=nc))
    y1 = m1.to(DEV).eval()(x, *conds, t, z)
    e1 = maxdiff(y1, gd[f'{name}.g1'])
    e2 = 0.0
    if f'{name}.g2' in gd:
        m2 = G2(cfg)
        m2.load_state_dict(O.make_state_dict(cfg, 'g2', 77, n_cond=nc))
        y2 = m2.to(DEV).eval()(x, *conds, t, z, g(gd[f'{name}.g1'])[:, [0], :].contiguous())
        e2 = maxdiff(y2, gd[f'{name}.g2'])
    print(f'variant {name}: max-abs G1 {e1:.2e} G2 {e2:.2e}')
    assert max(e1, e2) <= 1e-3


def test_fourier_embedding_and_naive_resamplers():
    ops, S, layerspp, ud, *_ = _imports()
    gen = torch.Generator().manual_seed(3)
    W = torch.randn(16, generator=gen) * 16.0
    t = torch.rand(5, generator=gen) * 3 + 0.01
    xp = torch.log(t)[:, None] * W[None, :] * 2 * np.pi
    want = torch.cat([torch.sin(xp), torch.cos(xp)], -1)
    got = ops.fourier_embedding(g(t), g(W))
    assert maxdiff(got, want) <= 2e-5          # |arg| up to ~500: one ulp of the argument is 3e-5
    x = torch.randn(2, 3, 6, 10, generator=gen)
    assert torch.equal(ud.naive_upsample_2d(g(x)).cpu(), O.naive_upsample_2d(x))
    assert maxdiff(ud.naive_downsample_2d(g(x)), O.naive_downsample_2d(x)) <= 1e-6


def test_config3_brats_shaped_batch32_vs_reference():
    """BASELINE config 3: a batch of 32 BraTS-shaped 256x256 slices through the captured
    hipGraph sampler (the batched driver's path).  The reference sampled the 4 distinct synthetic slices of the fixture
    (B=4); here they fill a batch of 32 eight times over with the same per-slice Gaussian draws, so every replica must
    reproduce the reference's final image, and PSNR / SSIM against the synthetic target must agree with the reference's to
    +-0.05 dB / +-0.001."""
    ops, S, *_ = _imports()
    gd = load_golden('batch_cfg3.npz')
    cfg = O.default_config()
    g1, g2 = _build(cfg)
    sl = gd['slices_u8'].float() / 255.0 * 2.0 - 1.0                     # [4, 4, 256, 256]: 3 conditions + target
    rep = lambda t: t.repeat(8, *([1] * (t.dim() - 1)))
    conds = [g(rep(sl[:, c:c + 1].contiguous())) for c in range(3)]
    x_init, zs, noises = sampler_inputs(cfg, 4, seed_x=314)
    sampler = S.GraphSampler(S.Posterior_Coefficients(cfg, DEV), g1, g2, cfg, 32, 256, 256, DEV)
    out = sampler.sample(*conds, g(rep(x_init)), 4, zs=[g(rep(z)) for z in zs], noises=[g(rep(n)) for n in noises]).cpu()
    ref = gd['final']
    err = max(maxdiff(out[r * 4:(r + 1) * 4], ref) for r in range(8))
    to01 = lambda a: (np.asarray(a, np.float64) + 1) / 2
    dps, dss = [], []
    for i in range(4):
        tgt = to01(sl[i, 3].numpy())
        dps.append(O.psnr(tgt, to01(out[i, 0].numpy())) - O.psnr(tgt, to01(ref[i, 0].numpy())))
        dss.append(O.ssim(tgt, to01(out[i, 0].numpy())) - O.ssim(tgt, to01(ref[i, 0].numpy())))
    print(f'cfg3 B=32: max-abs {err:.2e}, dPSNR {np.mean(dps):+.4f} dB (max {np.abs(dps).max():.4f}), dSSIM {np.mean(dss):+.6f}')
    assert err <= 1e-3 and np.abs(dps).max() <= 0.05 and np.abs(dss).max() <= 0.001


def test_config3_wide_16_slices_four_target_orderings_every_step_vs_reference():
    """BASELINE config 3 as SURVEY.md section 8(d) item 3 specifies it: a BraTS-shaped split over ALL FOUR target orderings
    (dataset/dataset_brats.py:29-34), batch 32 = 16 distinct slices x 2 through the captured sampler; every reverse step's x_new
    of every slice against the reference's own B=4 runs (<= 1e-3), PSNR / SSIM of every slice against its synthetic target
    within +-0.05 dB / +-0.001 of the reference's (tools/metric_calc.py:28-53)."""
    ops, S, *_ = _imports()
    from helpers import wide_cfg3_case
    cfg = O.default_config()
    g1, g2 = _build(cfg)
    case = wide_cfg3_case(cfg, copies=2)
    sampler = S.GraphSampler(S.Posterior_Coefficients(cfg, DEV), g1, g2, cfg, 32, 256, 256, DEV)
    out, steps = sampler.sample(*[g(c) for c in case['conds']], g(case['x_init']), 4, zs=[g(z) for z in case['zs']],
                                noises=[g(n) for n in case['noises']], return_steps=True)
    worst = []
    for k, st in enumerate(steps):
        xn = st[2].cpu()
        per_slice = (xn.view(2, 16, -1) - case['refs'][k].view(1, 16, -1)).abs().amax(dim=2).amax(dim=0)      # both replicas
        by_group = [float(per_slice[4 * i:4 * i + 4].max()) for i in range(4)]
        print(f'cfg3 wide step {k}: max-abs per target ordering ' + ' '.join(f'{n} {e:.2e}' for n, e in zip(case['groups'], by_group)))
        worst.append(max(by_group))
    to01 = lambda a: (np.asarray(a, np.float64) + 1) / 2
    ref_final, ours = case['refs'][-1], out.cpu()
    dps, dss = [], []
    for i in range(16):
        tgt = to01(case['targets'][i].numpy())
        dps.append(O.psnr(tgt, to01(ours[i, 0].numpy())) - O.psnr(tgt, to01(ref_final[i, 0].numpy())))
        dss.append(O.ssim(tgt, to01(ours[i, 0].numpy())) - O.ssim(tgt, to01(ref_final[i, 0].numpy())))
    print(f'cfg3 wide: worst per-step max-abs {max(worst):.2e}; dPSNR max |{np.abs(dps).max():.5f}| dB, dSSIM max |{np.abs(dss).max():.6f}| over 16 slices')
    assert max(worst) <= 1e-3 and np.abs(dps).max() <= 0.05 and np.abs(dss).max() <= 0.001


def test_deterministic_switch_gives_bit_stable_outputs():
    """MUD_DETERMINISTIC=1 (read at import, so a child process): no fp64 atomics - every GroupNorm takes the fixed-order two-pass
    reduction - and two runs of the sampler on the same inputs are bit-identical (the default path agrees to ~1e-6 only), while
    staying within the parity bar against the reference's outputs."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "tests")
from helpers import SMALL_CFGS, load_golden, sampler_inputs, small_conds
from oracle import mudiff_oracle as O
from mudiff_hip import ops, sampling as S
from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
assert ops.DETERMINISTIC
cfg = O.default_config(**SMALL_CFGS["s32"])
g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
g1.load_state_dict(O.make_state_dict(cfg, "g1", 1234)); g2.load_state_dict(O.make_state_dict(cfg, "g2", 1234))
g1, g2 = g1.cuda().eval(), g2.cuda().eval()
conds = [c.cuda() for c in small_conds(cfg)]
x_init, zs, noises = sampler_inputs(cfg, 2)
coef = S.Posterior_Coefficients(cfg, "cuda:0")
run = lambda: S.sample_from_model(coef, g1, conds[0], g2, conds[1], conds[2], cfg.num_timesteps, x_init.cuda(), None, cfg,
                                  zs=[z.cuda() for z in zs], noises=[n.cuda() for n in noises])
ops.PROFILE.enable(); a = run(); names = {r[0] for r in ops.PROFILE.records}; ops.PROFILE.disable()
outs = [run() for _ in range(4)]
assert all(torch.equal(a, o) for o in outs), "outputs differ between runs"
assert "gn_from_sums" not in names and "gn_scale_shift" in names
gd = load_golden("small_models.npz")
err = float((a.cpu() - gd["s32.step3.xnew"]).abs().max())
assert err <= 1e-3, err
print("DETERMINISTIC-OK", err)
'''
    env = dict(os.environ, MUD_DETERMINISTIC='1')
    from conftest import PKG, REPO
    env['PYTHONPATH'] = os.pathsep.join([REPO, PKG, env.get('PYTHONPATH', '')])
    p = subprocess.run([sys.executable, '-c', code], cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and 'DETERMINISTIC-OK' in p.stdout, p.stderr[-3000:]
