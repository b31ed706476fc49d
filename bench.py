#!/usr/bin/env python3
"""bench.py - MU-Diff reverse-diffusion sampling throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic slices: the complete 4-step
dual-generator reverse sampling (4 x [G1 -> G2 -> posterior]) of B 256x256 slices per GPU (BASELINE
config 2 shapes: nf=64, ch_mult 1-2-4, 2 res blocks, nz=100; random-init weights - no trained weights
exist offline; noise drawn on the device; like mudiff_hip.sampling.sample_from_model, the captured
sampler computes what depends on the condition images alone once per slice, not once per reverse step).  value = slices/s over ALL ranks = N*B*K / max-over-
ranks wall time, inputs resident in HBM, fp32 in / fp32 out.

One JSON line on rank 0, with
  roofline      the dominant kernel (3x3 implicit-GEMM conv on split-bf16 MFMA): algorithmic FLOPs /
                per-launch HIP-event time, measured in an instrumented pass of the same workload;
  cpu_baseline  the CPU oracle (port of the reference's PyTorch-CPU path) timed on this host's cores on
                a bounded sample (1 slice), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'mu-diff_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FLOP_PER_SLICE = 3456.4e9        # 4 x (378.87 + 485.24) GFLOP, B=1
FUSED_BYTES_PER_SLICE = 19.9e9   # algorithmic HBM bytes, fully fused ideal
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('MUDIFF_BENCH_BATCH', '16')), help='slices per GPU per step')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    return ap.parse_args()


def bench_config():
    """BASELINE config 2 (demo.ipynb cell 3 / SURVEY.md section 8): the attribute bag the generators' constructors read."""
    from types import SimpleNamespace
    return SimpleNamespace(
        num_timesteps=4, beta_min=0.1, beta_max=20.0, centered=True, use_geometric=False, num_channels=1, num_channels_dae=64,
        n_mlp=3, ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=(16,), dropout=0.0, resamp_with_conv=True, conditional=True,
        fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan', progressive='none',
        progressive_input='residual', progressive_combine='sum', embedding_type='positional', fourier_scale=16.0,
        not_use_tanh=False, image_size=256, nz=100, z_emb_dim=256, t_emb_dim=256)


def random_weights_(module, seed):
    """Random-init weights of the architecture (there are no trained weights offline).  MUDIFF_BENCH_WEIGHTS=uniform
    (default): every matrix / filter U(+-sqrt(3/fan_avg)) (the reference's default_init at scale 1, applied to ALL tensors -
    including Conv_1, NIN_3 and the output conv that the reference starts at ~0, which would make half the network
    multiply by zero), biases 0.1*N(0,1), norm gains 1 + 0.1*N(0,1).  =ctor: the constructors' own initialisers with only
    the ~0 tensors re-drawn."""
    import math
    mode = os.environ.get('MUDIFF_BENCH_WEIGHTS', 'uniform')
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                if mode == 'ctor' and float(p.abs().max()) >= 1e-6:
                    continue
                rf = p[0][0].numel() if p.dim() > 2 else 1
                bound = math.sqrt(3.0 / ((p.shape[0] + p.shape[1]) * rf / 2.0))
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            elif mode == 'ctor':
                if float(p.abs().max()) == 0.0:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith('.W'):
                continue                                            # Fourier frequencies keep their own scale
            else:
                t = 0.1 * torch.randn(p.shape, generator=g)
                if name.endswith('style.bias'):
                    t[:p.shape[0] // 2] += 1.0                      # AdaGN gamma half
                elif name.endswith('.weight'):
                    t += 1.0                                        # GroupNorm gains
                p.copy_(t)


def build_models(cfg, dev, rank, world):
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    torch.manual_seed(1234 + rank)
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    if rank == 0:
        random_weights_(g1, 1)
        random_weights_(g2, 2)
    g1, g2 = g1.to(dev).eval(), g2.to(dev).eval()
    if world > 1:
        # parameters live on rank 0 (checkpoint reader); one flattened RCCL broadcast per generator over xGMI
        from mudiff_hip.distributed import broadcast_parameters
        broadcast_parameters(g1, src=0)
        broadcast_parameters(g2, src=0)
    return g1, g2


def synthetic_batch(cfg, B, dev, seed):
    """BraTS-shaped synthetic slices: smooth z-scored fields clamped to [-1,1] inside a disc, zero (-1)
    background (dataset/dataset_brats.py:83,91 contract)."""
    g = torch.Generator().manual_seed(seed)
    H = cfg.image_size
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, H), indexing='ij')
    mask = ((yy ** 2 + xx ** 2) < 0.8).float()
    out = []
    for _ in range(3):
        f = torch.randn(B, 1, H // 8, H // 8, generator=g)
        f = torch.nn.functional.interpolate(f, size=(H, H), mode='bilinear', align_corners=False)
        out.append((torch.clamp(f * 1.5, -3, 3) / 3 * mask + (mask - 1)).to(dev))
    return out


def cpu_baseline(cfg):
    from oracle import mudiff_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, int(os.environ.get('MUDIFF_CPU_THREADS', '16'))))   # the 1-GPU box's CPU share is 16
    torch.set_num_threads(ncores)
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    g = torch.Generator().manual_seed(5)
    H = cfg.image_size
    conds = [torch.tanh(torch.randn(1, 1, H, H, generator=g)) for _ in range(3)]
    x0 = torch.randn(1, 1, H, H, generator=g)
    zs = [torch.randn(1, cfg.nz, generator=g) for _ in range(cfg.num_timesteps)]
    ns = [torch.randn(1, 1, H, H, generator=g) for _ in range(cfg.num_timesteps)]
    coef = O.PosteriorCoefficients(cfg)
    with torch.no_grad():
        O.g1_forward(sd1, cfg, x0, *conds, torch.zeros(1, dtype=torch.int64), zs[0])      # warm-up (thread pool, oneDNN primitives)
        t0 = time.perf_counter()
        O.sample_from_model(coef, sd1, sd2, cfg, *conds, x0, zs, ns)
        dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 5), unit='slices/s', cores=ncores, kind='port',
                sample=f'1 slice (B=1, {cfg.num_timesteps} steps, both generators, 256x256, nf=64) after a 1-forward warm-up; '
                       f'{dt:.2f} s; torch {torch.__version__} CPU fp32')


def traffic_from_profiles():
    """HBM bytes per launch of the dominant kernel from the latest committed PMC passes (profiles/rNN_*_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command, gfx950 correction applied).  rocprofv3 cannot run
    inside the timed process, so the committed measurement is reported, or null when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_traffic.json')))
    if not files:
        return None
    with open(files[-1]) as f:
        return json.load(f).get('dominant_hbm_bytes_per_launch')


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def main():
    a = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU fallback for the HIP path'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', init_method='env://', device_id=dev)   # nccl == RCCL on ROCm

    from mudiff_hip import ops, sampling as S
    cfg = bench_config()
    B, K, W = a.batch, a.steps, a.warmup
    H = cfg.image_size
    g1, g2 = build_models(cfg, dev, rank, world)
    coef = S.Posterior_Coefficients(cfg, dev)
    c1, c2, c3 = synthetic_batch(cfg, B, dev, seed=100 + rank)     # each rank owns its shard of slices
    x_init = torch.randn(B, 1, H, H, device=dev)

    log(f'models built, rank {rank}/{world}, B={B}')
    if a.no_graph:
        def one_step():
            return S.sample_from_model(coef, g1, c1, g2, c2, c3, cfg.num_timesteps, x_init, None, cfg)
    else:
        sampler = S.GraphSampler(coef, g1, g2, cfg, B, H, H, dev)

        def one_step():
            return sampler.sample(c1, c2, c3, x_init, cfg.num_timesteps)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    log('sampler ready (hipGraph captured)' if not a.no_graph else 'eager mode')
    for _ in range(W):
        out = one_step()
    barrier()
    log('warm-up done')
    t0 = time.perf_counter()
    for _ in range(K):
        out = one_step()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    from mudiff_hip.distributed import max_over_ranks
    dt = max_over_ranks(dt, dev)

    slices = world * B * K
    value = slices / dt
    log(f'timed region: {dt:.3f} s for {slices} slices -> {value:.2f} slices/s')
    line = {
        'metric': '256x256 slices/sec (4-step dual-gen reverse)', 'value': round(value, 3), 'unit': 'slices/s',
        'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(1e3 * dt / K, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32 (convs/attention: bf16 hi+lo split MFMA x3, fp32 accumulate)',
        'data': 'synthetic',
        'config': {'workload': f'BASELINE config 2 shapes: 4-step dual-generator sampling, 256x256, nf=64, ch_mult 1-2-4, '
                               f'{B} slices per GPU per step, batch-sharded over {world} GPU(s), weights replicated',
                   'slices_per_gpu_per_step': B, 'hipgraph': not a.no_graph},
        'end_to_end': {'fp32_flop_frac': round(value * FLOP_PER_SLICE / 1e12 / world / PEAK_FP32_TFLOPS, 4),
                       'bf16x3_issued_frac': round(3 * value * FLOP_PER_SLICE / 1e12 / world / PEAK_BF16_TFLOPS, 4),
                       'fused_hbm_frac': round(value * FUSED_BYTES_PER_SLICE / 1e9 / world / PEAK_HBM_GBS, 4)},
    }

    if rank == 0 and not a.no_roofline:
        # instrumented eager pass of the same workload: HIP events around every launch of the dominant kernel
        ops.PROFILE.enable()
        S.sample_from_model(coef, g1, c1, g2, c2, c3, cfg.num_timesteps, x_init, None, cfg)
        torch.cuda.synchronize()
        prof = ops.PROFILE.summary()
        ops.PROFILE.disable()
        k = prof.get('conv_mfma_k3')
        if k:
            ach = k['flops'] / (k['ms'] * 1e-3) / 1e12
            line['roofline'] = {'kernel': 'k_conv_mfma<3> (3x3 implicit GEMM, split-bf16 MFMA)', 'bound': 'mfma',
                                'achieved': round(ach, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                                'traffic': traffic_from_profiles(),
                                'launches': k['n'], 'avg_launch_us': round(1e3 * k['ms'] / k['n'], 2),
                                'algorithmic_gflop_per_launch': round(k['flops'] / k['n'] / 1e9, 3),
                                'algorithmic_bytes_per_launch': int(k['bytes'] / k['n']),
                                'issued_bf16_tflops': round(3 * ach, 2), 'issued_frac': round(3 * ach / PEAK_BF16_TFLOPS, 4),
                                'vs_fp32_peak_157.3': round(ach / PEAK_FP32_TFLOPS, 3),
                                'share_of_gpu_time': round(k['ms'] / sum(v['ms'] for v in prof.values()), 3)}
            line['kernel_time_ms_per_batch'] = {n: round(v['ms'], 3) for n, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
    if rank == 0 and world == 1 and B != 1 and not a.no_roofline:
        # latency case of BASELINE config 2 read literally (batch = 1): one slice at a time through its own hipGraph
        s1 = S.GraphSampler(coef, g1, g2, cfg, 1, H, H, dev)
        x1 = torch.randn(1, 1, H, H, device=dev)
        for _ in range(2):
            s1.sample(c1[:1], c2[:1], c3[:1], x1, cfg.num_timesteps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(6):
            s1.sample(c1[:1], c2[:1], c3[:1], x1, cfg.num_timesteps)
        torch.cuda.synchronize()
        line['batch1'] = {'slices_per_s': round(6 / (time.perf_counter() - t1), 2), 'note': 'same path, 1 slice per step (latency case)'}
        del s1
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log('timing the CPU oracle on one slice ...')
        line['cpu_baseline'] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
