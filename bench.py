#!/usr/bin/env python3
"""bench.py - MU-Diff reverse-diffusion sampling throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--total-slices M] [--sweep 1,2,4,8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic slices: the complete 4-step
dual-generator reverse sampling (4 x [G1 -> G2 -> posterior]) of B 256x256 slices per GPU (BASELINE
config 2 shapes: nf=64, ch_mult 1-2-4, 2 res blocks, nz=100; random-init weights - no trained weights
exist offline; noise drawn on the device; like mudiff_hip.sampling.sample_from_model, the captured
sampler computes what depends on the condition images alone once per slice, not once per reverse step).
value = slices/s over ALL ranks = slices sampled / max-over-ranks wall time, inputs resident in HBM, fp32 in /
fp32 out.

Ranks.  One process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a
launcher: it starts N fresh rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
MASTER_PORT set) before it makes any GPU call, relays rank 0's JSON line and exits non-zero if any rank fails
(the reference's analogue: engine/train.py:1454-1470).  Under torch.distributed.run the ranks already exist and
WORLD_SIZE must equal --gpus.

Scaling.  Default: weak (B slices per GPU per step, so N GPUs sample N*B slices per step).  `--total-slices M`:
strong - a step is one pass over M slices sharded contiguously over the ranks (mudiff_hip.distributed.shard_range;
SURVEY.md section 8(d) item 4, M = 512 there).  `--sweep 1,2,4,8`: runs the strong-scaling bench at every listed GPU
count that fits the visible devices and prints ONE line {n: slices/s} with the CPU baseline beside it.

One JSON line on rank 0, with
  roofline        the dominant kernel (3x3 implicit-GEMM conv on split-precision MFMA, both arithmetic plans): algorithmic FLOPs /
                  per-launch HIP-event time, measured in an instrumented pass of the same workload;
  cpu_baseline    the CPU oracle (port of the reference's PyTorch-CPU path) timed on this host's cores on
                  a bounded sample (1 slice), rank 0, N=1 only;
  parity          BASELINE config 2 with the committed reference-made fixture (tests/golden/full_cfg2.npz: the
                  reference's demo images, injected draws) through the same captured sampler: per-step max-abs,
                  dPSNR, dSSIM against the reference's own outputs (N=1 only);
  batch1/batch32  the same path at 1 (config 2 read literally) and 32 (config 3) slices per step;
  pcie_inclusive  H2D of the three condition batches + sampling + D2H of the result inside the timed region.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, 'mu-diff_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402   (importing torch does not touch the GPU)

FLOP_PER_SLICE = 3456.4e9        # 4 x (378.87 + 485.24) GFLOP, B=1, the reference's graph
FUSED_BYTES_PER_SLICE = 19.9e9   # algorithmic HBM bytes, fully fused ideal
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
METRIC = '256x256 slices/sec (4-step dual-gen reverse)'


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('MUDIFF_BENCH_BATCH', '32')),
                    help='slices per GPU per captured reverse step (default 32 = BASELINE config 3)')
    ap.add_argument('--total-slices', type=int, default=0, help='strong scaling: a step is one pass over this many slices sharded over the ranks')
    ap.add_argument('--sweep', default='', help='comma-separated GPU counts, e.g. 1,2,4,8: strong-scaling curve + CPU baseline in one line')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-full', action='store_true',
                    help='BASELINE.md section 3 in full: adds the 1-thread figure and B=8 to the default (all cores, B=1, 2 warm-up + 3 timed slices); ~10 min')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the parity / batch1 / batch32 / pcie legs')
    return ap.parse_args(argv)


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------
# launcher (no GPU call happens in this process)
# ---------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def last_json_line(text):
    for ln in reversed(text.strip().splitlines()):
        ln = ln.strip()
        if ln.startswith('{') and ln.endswith('}'):
            try:
                return json.loads(ln)
            except ValueError:
                continue
    return None


def launch_ranks(n, argv, extra_env=None, timeout=None):
    """Start n rank processes of this script (one per GPU) and wait.  -> (returncode, rank-0 stdout).
    Any rank failing terminates the others (exactly the PIDs started here) and the result is non-zero.
    timeout: seconds (default MUDIFF_BENCH_TIMEOUT, 1500): a rank hung in the RCCL rendezvous or in a barrier must not
    hold the GPUs until an outer driver kills the launcher - the ranks started here are stopped and the result is 124."""
    if timeout is None:
        timeout = float(os.environ.get('MUDIFF_BENCH_TIMEOUT', '1500'))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    t0 = time.time()
    rc = 0
    live = set(range(n))
    out0 = None
    try:
        while live:
            for r in sorted(live):
                if r == 0 and out0 is None and procs[0].poll() is not None:
                    out0 = procs[0].stdout.read()
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = rc or code
                    log(f'rank {r} exited with code {code}; stopping the other ranks')
                    for o in sorted(live):
                        procs[o].terminate()
            if timeout is not None and time.time() - t0 > timeout:
                rc = rc or 124
                log(f'launcher timeout after {timeout} s; stopping the ranks')
                for o in sorted(live):
                    procs[o].terminate()
                timeout = None
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    if out0 is None:
        out0 = procs[0].stdout.read() or ''
    return rc, out0


def visible_gpus():
    n = os.environ.get('MUDIFF_BENCH_VISIBLE_GPUS')      # test hook; device_count() does not initialise the GPU on this image
    return int(n) if n else torch.cuda.device_count()


def run_sweep(a, argv):
    """Strong-scaling curve: the same M slices at every GPU count that fits, CPU baseline beside it, ONE line."""
    counts = [int(c) for c in a.sweep.split(',') if c.strip()]
    have = visible_gpus()
    total = a.total_slices or 512
    clean, skip = [], False                                # the caller's flags minus the ones each run sets itself
    for x in argv:
        if skip:
            skip = False
        elif x in ('--sweep', '--gpus', '--total-slices'):
            skip = True
        elif not x.startswith(('--sweep=', '--gpus=', '--total-slices=')):
            clean.append(x)
    curve, lines, skipped, rc_all = {}, {}, [], 0
    have_extras = False                                    # cpu_baseline + roofline ride on the first run that SUCCEEDS
    for n in counts:
        if n > have:
            skipped.append(n)
            continue
        args = clean + ['--gpus', str(n), '--total-slices', str(total), '--no-extras']
        if have_extras or a.no_cpu_baseline:
            args.append('--no-cpu-baseline')
        if have_extras:
            args.append('--no-roofline')
        log(f'sweep: {n} GPU(s) ...')
        if n == 1:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), *args], stdout=subprocess.PIPE, text=True,
                               env={k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')})
            rc, out = p.returncode, p.stdout
        else:
            rc, out = launch_ranks(n, args)
        line = last_json_line(out or '')
        if rc != 0 or line is None:
            rc_all = rc_all or rc or 1
            log(f'sweep: the {n}-GPU run failed (rc {rc})')
            continue
        curve[str(n)] = line['value']
        lines[str(n)] = line
        have_extras = True
    first = next(iter(lines.values())) if lines else {}
    out = {'metric': METRIC, 'unit': 'slices/s', 'higher_is_better': True, 'scaling': 'strong', 'data': 'synthetic',
           'total_slices': total, 'slices_per_s_by_gpus': curve,
           'value': max(curve.values()) if curve else None, 'n_gpus': max((int(k) for k in curve), default=0),
           'ms_per_step_by_gpus': {k: v['ms_per_step'] for k, v in lines.items()},
           'ranks_seen_by_gpus': {k: v.get('ranks_seen') for k, v in lines.items()},
           'per_rank_by_gpus': {k: v.get('per_rank') for k, v in lines.items() if v.get('per_rank')},
           'param_broadcast_by_gpus': {k: v.get('param_broadcast') for k, v in lines.items() if v.get('param_broadcast')},
           # strong scaling: value(n) / (n x value(1)), only when the 1-GPU point ran in this sweep (the driver computes its own)
           'efficiency_vs_1gpu': ({k: round(v / (int(k) * curve['1']), 4) for k, v in curve.items()} if curve.get('1') else None),
           'skipped_gpu_counts': skipped, 'visible_gpus': have, 'steps': a.steps, 'warmup': a.warmup,
           'dtype': first.get('dtype'), 'config': first.get('config'), 'vs_baseline': None,
           'cpu_baseline': first.get('cpu_baseline'), 'roofline': first.get('roofline')}
    print(json.dumps(out), flush=True)
    return rc_all


# ---------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------
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
    bcast = None
    if world > 1:
        # parameters live on rank 0 (checkpoint reader); one flattened RCCL broadcast per generator over xGMI
        bcast = timed_broadcast([g1, g2], dev)
    return g1, g2, bcast


def timed_broadcast(modules, dev):
    """The load-time parameter broadcast (north_star: "RCCL broadcast of params over xGMI at load only"; reference
    engine/train.py:188-190 sends every tensor separately): one flattened message per module.  -> bytes, wall ms (MAX over
    ranks, barrier on both sides; the first collective of a process group also pays the communicator set-up) and GB/s."""
    import torch.distributed as dist
    from mudiff_hip.distributed import broadcast_parameters, max_over_ranks
    cuda = torch.device(dev).type == 'cuda'
    dist.barrier()
    if cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes = sum(broadcast_parameters(m, src=0) for m in modules)
    if cuda:
        torch.cuda.synchronize()
    dist.barrier()
    ms = 1e3 * max_over_ranks(time.perf_counter() - t0, dev)
    return {'bytes': int(nbytes), 'messages': len(modules), 'ms': round(ms, 3), 'gb_per_s': round(nbytes / ms / 1e6, 3) if ms > 0 else None,
            'note': 'one flattened broadcast per generator from rank 0, includes communicator set-up of the first collective'}


def per_rank_rates(slices_local, dt_local, dev, world):
    """Every rank's own slices/s over its own timed region (all_gather), next to the MAX-over-ranks headline."""
    import torch.distributed as dist
    mine = torch.tensor([slices_local / dt_local], dtype=torch.float64, device=dev)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    rates = [round(float(t.item()), 3) for t in got]
    return {'slices_per_s': rates, 'min': min(rates), 'max': max(rates)}


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


def cpu_baseline(cfg, full=False):
    """SURVEY.md section 8(d) / BASELINE.md section 3: the CPU oracle (the reference's PyTorch-CPU path restated; checker code,
    timed here and nowhere shipped) on this host's cores.  Default: all cores, B=1, 2 warm-up slices + 3 timed slices
    (value = 3 / their total time).  full=True adds the 1-thread figure (1 warm-up forward + 1 timed slice) and B=8
    (one timed batch after a 1-forward warm-up)."""
    import platform
    from oracle import mudiff_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, int(os.environ.get('MUDIFF_CPU_THREADS', '16'))))   # the 1-GPU box's CPU share is 16
    sd1, sd2 = O.make_state_dict(cfg, 'g1', 1234), O.make_state_dict(cfg, 'g2', 1234)
    H, T = cfg.image_size, cfg.num_timesteps
    coef = O.PosteriorCoefficients(cfg)

    def draws(B, seed):
        g = torch.Generator().manual_seed(seed)
        conds = [torch.tanh(torch.randn(B, 1, H, H, generator=g)) for _ in range(3)]
        x0 = torch.randn(B, 1, H, H, generator=g)
        return conds, x0, [torch.randn(B, cfg.nz, generator=g) for _ in range(T)], [torch.randn(B, 1, H, H, generator=g) for _ in range(T)]

    def run(B, seed):
        conds, x0, zs, ns = draws(B, seed)
        t0 = time.perf_counter()
        O.sample_from_model(coef, sd1, sd2, cfg, *conds, x0, zs, ns)
        return time.perf_counter() - t0

    def warm_forward(B):
        conds, x0, zs, _ = draws(B, 4)
        O.g1_forward(sd1, cfg, x0, *conds, torch.zeros(B, dtype=torch.int64), zs[0])

    cpu_model = platform.processor() or ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), cpu_model)
    except OSError:
        pass
    n_warm, n_timed = int(os.environ.get('MUDIFF_CPU_WARMUP', '2')), int(os.environ.get('MUDIFF_CPU_SLICES', '3'))
    with torch.no_grad():
        torch.set_num_threads(ncores)
        for i in range(n_warm):
            run(1, 100 + i)
        times = [run(1, 5 + i) for i in range(n_timed)]
        out = dict(value=round(n_timed / sum(times), 5), unit='slices/s', cores=ncores, kind='port',
                   sample=f'{n_timed} slices one at a time (B=1, {T} steps, both generators, 256x256, nf=64) after {n_warm} warm-up slices; '
                          f'{", ".join(f"{t:.2f}" for t in times)} s; torch {torch.__version__} CPU fp32; {cpu_model}',
                   s_per_slice=[round(t, 3) for t in times], cpu_model=cpu_model, torch=torch.__version__)
        if full:
            warm_forward(8)
            t8 = run(8, 9)
            out['b8'] = dict(value=round(8 / t8, 5), unit='slices/s', cores=ncores, sample=f'one batch of 8 after a 1-forward warm-up; {t8:.2f} s')
            torch.set_num_threads(1)
            warm_forward(1)
            t1 = run(1, 5)
            out['one_thread'] = dict(value=round(1 / t1, 5), unit='slices/s', cores=1, sample=f'1 slice after a 1-forward warm-up; {t1:.2f} s')
            torch.set_num_threads(ncores)
    return out


def csrc_digest():
    """Content hash of the kernel sources: ties a committed PMC measurement to the kernels it was taken on (the GPU box has no
    .git, so `git log -1 -- mu-diff_amd/csrc` is not available there)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(REPO, 'mu-diff_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(REPO, 'mu-diff_amd', 'csrc', '*.h'))):
        with open(fn, 'rb') as f:
            h.update(os.path.basename(fn).encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def traffic_from_profiles():
    """-> (HBM bytes per launch of the dominant kernel, source file) from the latest committed PMC passes
    (profiles/rNN_*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command in separate passes, gfx950
    correction applied).  rocprofv3 cannot run inside the timed process, so a committed measurement is reported - and only
    when it was taken on the kernel sources of this tree (`csrc_digest` recorded in the file); otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    src = os.path.relpath(files[-1], REPO)
    if d.get('csrc_digest') != csrc_digest():
        return None, f'{src} is stale (taken on other kernel sources)'
    return d.get('dominant_hbm_bytes_per_launch'), src


# ---------------------------------------------------------------------------------------------------
# reference-made fixture of BASELINE config 2 (tests/golden/full_cfg2.npz) through the captured sampler
# ---------------------------------------------------------------------------------------------------
def _demo_cond(u8):
    """demo.ipynb cell 4 of the reference: percentile-1/99 clip over non-zero pixels, min-max, (x-0.5)/0.5, rot90(k=-1)."""
    import numpy as np
    img = np.asarray(u8)
    low, high = np.percentile(img[img > 0], [1, 99])
    img = np.clip(img, low, high)
    img = (img - img.min()) / (img.max() - img.min())
    img = (img - 0.5) / 0.5
    return torch.rot90(torch.tensor(img, dtype=torch.float32)[None, None], k=-1, dims=(2, 3)).contiguous()


def parity_leg(cfg, dev):
    """The reference's own run of config 2 (its demo JPEGs as conditions, seeded weights, recorded x_init / z / noise draws;
    tests/golden/make_golden.py) against the HIP path through GraphSampler: what the -m gpu parity test checks, in the line."""
    import numpy as np
    from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
    from mudiff_hip import driver, sampling as S
    from mudiff_hip.weights import seeded_state_dict
    gold = os.path.join(REPO, 'tests', 'golden')
    with np.load(os.path.join(gold, 'full_cfg2.npz'), allow_pickle=False) as z:
        ref = {k: torch.from_numpy(z[k]) for k in z.files}
    with np.load(os.path.join(gold, 'demo_inputs_u8.npz'), allow_pickle=False) as z:
        conds = [_demo_cond(z[n]).to(dev) for n in ('flair', 't2', 't1')]
        target = _demo_cond(z['t1ce'])[0, 0].numpy()
    g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
    g1.load_state_dict(seeded_state_dict(g1, 'g1', 1234, cfg.fourier_scale))
    g2.load_state_dict(seeded_state_dict(g2, 'g2', 1234, cfg.fourier_scale))
    g1, g2 = g1.to(dev).eval(), g2.to(dev).eval()
    H = cfg.image_size
    gen = torch.Generator().manual_seed(42)                       # the draws of tests/golden/make_golden.py (tests/helpers.py::sampler_inputs)
    x_init = torch.randn(1, 1, H, H, generator=gen)
    st = torch.get_rng_state()
    torch.manual_seed(43)
    zs, noises = [], []
    for _ in range(cfg.num_timesteps):
        zs.append(torch.randn(1, cfg.nz))
        noises.append(torch.randn(1, 1, H, H))
    torch.set_rng_state(st)
    sampler = S.GraphSampler(S.Posterior_Coefficients(cfg, dev), g1, g2, cfg, 1, H, H, dev)
    x, steps = sampler.sample(*conds, x_init.to(dev), cfg.num_timesteps, zs=[t.to(dev) for t in zs], noises=[t.to(dev) for t in noises],
                              return_steps=True)
    per_step = []
    for k, stp in enumerate(steps):
        per_step.append(max(float((v.cpu() - ref[f'step{k}.{nm}']).abs().max()) for nm, v in zip(('x01', 'x02', 'xnew'), stp)))
    to01 = lambda t: (np.asarray(t, np.float64) + 1) / 2          # noqa: E731
    ours, theirs = x.cpu()[0, 0].numpy(), ref[f'step{cfg.num_timesteps - 1}.xnew'][0, 0].numpy()
    dp = driver.psnr(to01(target), to01(ours)) - driver.psnr(to01(target), to01(theirs))
    ds = driver.ssim(to01(target), to01(ours)) - driver.ssim(to01(target), to01(theirs))
    out = {'fixture': 'tests/golden/full_cfg2.npz (outputs of the reference itself on its demo images, B=1, injected draws)',
           'max_abs_per_step': [float(f'{e:.3e}') for e in per_step], 'tolerance': 1e-3, 'dpsnr_db': round(dp, 5), 'dssim': round(ds, 6),
           'ok': bool(max(per_step) <= 1e-3 and abs(dp) <= 0.05 and abs(ds) <= 0.001)}
    del sampler
    wide = parity_wide_cfg3(cfg, dev, g1, g2)
    out['config3_wide'] = wide
    out['ok'] = bool(out['ok'] and wide['ok'])
    return out


def parity_wide_cfg3(cfg, dev, g1, g2):
    """BASELINE config 3 (tests/golden/wide_cfg3.npz): 16 distinct BraTS-shaped slices, 4 per target ordering
    (dataset/dataset_brats.py:29-34), the reference's own B=4 runs with every step's x_new - here as ONE batch of 32 (each slice
    twice) through the captured sampler, i.e. exactly the headline workload.  Worst per-step max-abs over the 16 slices, and the
    worst PSNR / SSIM difference to the reference per slice (tools/metric_calc.py:28-53 definitions, same code on both sides)."""
    import numpy as np
    from mudiff_hip import driver, sampling as S
    orders = {'T1CE': ['FLAIR', 'T2', 'T1', 'T1CE'], 'FLAIR': ['T1CE', 'T1', 'T2', 'FLAIR'], 'T2': ['T1CE', 'T1', 'FLAIR', 'T2'], 'T1': ['FLAIR', 'T1CE', 'T2', 'T1']}
    mods = ['FLAIR', 'T2', 'T1', 'T1CE']
    with np.load(os.path.join(REPO, 'tests', 'golden', 'wide_cfg3.npz'), allow_pickle=False) as z:
        gd = {k: torch.from_numpy(z[k]) for k in z.files}
    sl = gd['slices_u8'].float() / 255.0 * 2.0 - 1.0
    H, T = cfg.image_size, cfg.num_timesteps
    conds, xs, zs, ns, tg = [[], [], []], [], [[] for _ in range(T)], [[] for _ in range(T)], []
    for gi, order in enumerate(orders.values()):
        idx = slice(4 * gi, 4 * gi + 4)
        for c in range(3):
            conds[c].append(sl[idx, mods.index(order[c])][:, None])
        tg.append(sl[idx, mods.index(order[3])])
        gen = torch.Generator().manual_seed(314 + gi)            # the draws of tests/golden/make_golden.py::golden_cfg3_wide (tests/helpers.py::sampler_inputs)
        xs.append(torch.randn(4, 1, H, H, generator=gen))
        st = torch.get_rng_state()
        torch.manual_seed(315 + gi)
        for k in range(T):
            zs[k].append(torch.randn(4, cfg.nz))
            ns[k].append(torch.randn(4, 1, H, H))
        torch.set_rng_state(st)
    rep = lambda parts: torch.cat(parts, 0).repeat(2, *([1] * (parts[0].dim() - 1))).contiguous().to(dev)      # noqa: E731
    sampler = S.GraphSampler(S.Posterior_Coefficients(cfg, dev), g1, g2, cfg, 32, H, H, dev)
    x, steps = sampler.sample(rep(conds[0]), rep(conds[1]), rep(conds[2]), rep(xs), T, zs=[rep(z) for z in zs], noises=[rep(n) for n in ns], return_steps=True)
    per_step = []
    for k, stp in enumerate(steps):
        ref = torch.cat([gd[f'{name}.step{k}.xnew'] for name in orders], 0)
        per_step.append(float((stp[2].cpu().view(2, 16, -1) - ref.view(1, 16, -1)).abs().max()))
    ref_final = torch.cat([gd[f'{name}.step{T - 1}.xnew'] for name in orders], 0)
    targets = torch.cat(tg, 0)
    to01 = lambda t: (np.asarray(t, np.float64) + 1) / 2          # noqa: E731
    dps, dss = [], []
    ours = x.cpu()
    for i in range(16):
        t = to01(targets[i].numpy())
        dps.append(driver.psnr(t, to01(ours[i, 0].numpy())) - driver.psnr(t, to01(ref_final[i, 0].numpy())))
        dss.append(driver.ssim(t, to01(ours[i, 0].numpy())) - driver.ssim(t, to01(ref_final[i, 0].numpy())))
    return {'fixture': 'tests/golden/wide_cfg3.npz (the reference\'s B=4 runs of 16 synthetic slices over the 4 target orderings; here one batch of 32)',
            'max_abs_per_step': [float(f'{e:.3e}') for e in per_step], 'dpsnr_db_max_abs': round(float(np.abs(dps).max()), 5),
            'dssim_max_abs': round(float(np.abs(dss).max()), 6),
            'ok': bool(max(per_step) <= 1e-3 and np.abs(dps).max() <= 0.05 and np.abs(dss).max() <= 0.001)}


def timed_batches(sampler_fn, n_iter):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_iter):
        sampler_fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------
def dryrun_worker(a, rank, world):
    """Launcher self-test (tests/test_bench_launcher.py): the rank plumbing of the real worker - env wiring, process
    group, barrier, MAX-over-ranks timing, ranks_seen, rank-0 line - over gloo on CPU tensors with a stand-in timed region."""
    import torch.distributed as dist
    from mudiff_hip.distributed import max_over_ranks, shard_range
    if os.environ.get('MUDIFF_BENCH_FAIL_RANK') == str(rank):
        sys.exit(3)
    if world > 1:
        dist.init_process_group(backend='gloo', init_method='env://')
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    lo, hi = shard_range(a.total_slices, rank, world) if a.total_slices else (rank * a.batch, (rank + 1) * a.batch)
    bcast = per_rank = None
    if world > 1:
        torch.manual_seed(rank)                               # ranks start with DIFFERENT parameters; rank 0's must win
        mods = [torch.nn.Linear(8, 4), torch.nn.Linear(4, 2)]
        bcast = timed_broadcast(mods, 'cpu')
        ref = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(ref, sum(p.double().sum() for m in mods for p in m.parameters()).float().reshape(1))
        assert all(torch.equal(r, ref[0]) for r in ref), 'parameters differ between ranks after the broadcast'
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, 'cpu')
    if world > 1:
        per_rank = per_rank_rates(hi - lo, dt_local, 'cpu', world)
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, (rank, lo, hi))
    else:
        seen = [(rank, lo, hi)]
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'dryrun': True, 'n_gpus': world, 'ranks_seen': sorted(s[0] for s in seen),
                          'shards': [list(s[1:]) for s in sorted(seen)], 'value': round((a.total_slices or world * a.batch) / dt, 3),
                          'ms_per_step': round(1e3 * dt, 3), 'scaling': 'strong' if a.total_slices else 'weak',
                          **({'per_rank': per_rank, 'param_broadcast': bcast} if world > 1 else {})}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def worker(a):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch exactly --gpus ranks '
                         f'(python bench.py --gpus {a.gpus} starts them itself when WORLD_SIZE is unset)')
    if os.environ.get('MUDIFF_BENCH_DRYRUN') == '1':
        return dryrun_worker(a, rank, world)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU fallback for the HIP path'
    if os.environ.get('MUDIFF_BENCH_SAME_GPU') == '1':      # rehearsal of the N-rank path on a one-GPU box (with MUDIFF_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ranks_seen = [0]
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('MUDIFF_BENCH_BACKEND', 'nccl')                       # nccl == RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', init_method='env://', device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method='env://')
        assert dist.get_world_size() == a.gpus
        got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank], dtype=torch.int64, device=dev))
        ranks_seen = sorted(int(t.item()) for t in got)
        assert ranks_seen == list(range(world)), ranks_seen

    from mudiff_hip import ops, sampling as S
    from mudiff_hip.distributed import max_over_ranks, shard_range
    cfg = bench_config()
    B, K, W = a.batch, a.steps, a.warmup
    H = cfg.image_size
    g1, g2, bcast = build_models(cfg, dev, rank, world)
    coef = S.Posterior_Coefficients(cfg, dev)
    strong = a.total_slices > 0
    if strong:
        lo, hi = shard_range(a.total_slices, rank, world)         # this rank's contiguous shard of the M slices
        n_local = hi - lo
        nb = (n_local + B - 1) // B
        conds = synthetic_batch(cfg, max(nb * B, B), dev, seed=100 + rank)     # resident in HBM; the tail batch is padded
        batches = [tuple(c[i * B:(i + 1) * B] for c in conds) for i in range(nb)]
    else:
        n_local = B
        batches = [tuple(synthetic_batch(cfg, B, dev, seed=100 + rank))]       # each rank owns its shard of slices
    c1, c2, c3 = batches[0] if batches else synthetic_batch(cfg, B, dev, seed=100 + rank)
    x_init = torch.randn(B, 1, H, H, device=dev)

    log(f'models built, rank {rank}/{world}, B={B}' + (f', slices [{lo}, {hi}) of {a.total_slices}' if strong else ''))
    if a.no_graph:
        def one_batch(cs):
            return S.sample_from_model(coef, g1, cs[0], g2, cs[1], cs[2], cfg.num_timesteps, x_init, None, cfg)
    else:
        sampler = S.GraphSampler(coef, g1, g2, cfg, B, H, H, dev)

        def one_batch(cs):
            return sampler.sample(cs[0], cs[1], cs[2], x_init, cfg.num_timesteps)

    def one_step():
        out = None
        for cs in batches:
            out = one_batch(cs)
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    log('sampler ready (hipGraph captured)' if not a.no_graph else 'eager mode')
    out = None
    for _ in range(W):
        out = one_step()
    barrier()
    log('warm-up done')
    t0 = time.perf_counter()
    for _ in range(K):
        out = one_step()
    barrier()
    dt_local = dt = time.perf_counter() - t0
    assert out is None or torch.isfinite(out).all()
    dt = max_over_ranks(dt, dev)
    per_rank = per_rank_rates(n_local * K, dt_local, dev, world) if world > 1 else None

    slices = (a.total_slices if strong else world * B) * K
    value = slices / dt
    log(f'timed region: {dt:.3f} s for {slices} slices -> {value:.2f} slices/s')
    which = 'BASELINE config 3 (BraTS-shaped test-split batches of 32; model and slice shapes of config 2)' if B == 32 else \
            ('BASELINE config 2 read literally (one slice at a time)' if B == 1 else 'BASELINE config 2 shapes')
    workload = (f'{which}: 4-step dual-generator sampling, 256x256, nf=64, ch_mult 1-2-4, '
                + (f'{a.total_slices} slices per step sharded contiguously over {world} GPU(s) in batches of {B}'
                   if strong else f'{B} slices per GPU per step, batch-sharded over {world} GPU(s)') + ', weights replicated')
    line = {
        'metric': METRIC, 'value': round(value, 3), 'unit': 'slices/s',
        'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(1e3 * dt / K, 3), 'higher_is_better': True,
        'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f32 (convs/attention: fp16 hi+lo split MFMA x3; 3x3 convs that fill the chip: fp16 hi.hi + e4m3 cross terms; fp32 accumulate)',
        'data': 'synthetic', 'ranks_seen': ranks_seen,
        **({'per_rank': per_rank, 'param_broadcast': bcast} if world > 1 else {}),
        'config': {'workload': workload, 'slices_per_gpu_per_step': n_local, 'batch': B, 'hipgraph': not a.no_graph,
                   **({'total_slices': a.total_slices} if strong else {})},
    }
    flop_per_slice = FLOP_PER_SLICE

    if rank == 0 and not a.no_roofline:
        # instrumented eager pass of the same workload: HIP events around every launch of the dominant kernel
        ops.PROFILE.enable()
        S.sample_from_model(coef, g1, c1, g2, c2, c3, cfg.num_timesteps, x_init, None, cfg)
        torch.cuda.synchronize()
        prof = ops.PROFILE.summary()
        ops.PROFILE.disable()
        # the dominant kernel = k_conv_mfma<3> under both of its arithmetic plans (ops.conv names the launches conv_mfma_k3 / conv_mfma_k3_fp8x)
        parts = {n: v for n, v in prof.items() if n.startswith('conv_mfma_k3')}
        if parts:
            k = {f: sum(v[f] for v in parts.values()) for f in ('n', 'ms', 'flops', 'bytes')}
            x8 = parts.get('conv_mfma_k3_fp8x', dict(n=0, ms=0.0, flops=0.0))
            ach = k['flops'] / (k['ms'] * 1e-3) / 1e12
            # 16-bit-MFMA cycles issued per algorithmic FLOP: 3 under the fp16 x 3 plan, 7/3 under fp16 + e4m3 cross terms (224 / 96 cycles
            # per 3-tap group; the e4m3 MFMA counted at its cycles, i.e. in 16-bit equivalents)
            issued = (3.0 * (k['flops'] - x8['flops']) + 7.0 / 3.0 * x8['flops']) / (k['ms'] * 1e-3) / 1e12
            traffic, tsrc = traffic_from_profiles()
            line['roofline'] = {'kernel': 'k_conv_mfma<3> (3x3 implicit GEMM; split-fp16 MFMA x3, or fp16 hi.hi + e4m3 cross terms)', 'bound': 'mfma',
                                'achieved': round(ach, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                                'traffic': traffic, 'traffic_source': tsrc,
                                'launches': k['n'], 'avg_launch_us': round(1e3 * k['ms'] / k['n'], 2),
                                'algorithmic_gflop_per_launch': round(k['flops'] / k['n'] / 1e9, 3),
                                'algorithmic_bytes_per_launch': int(k['bytes'] / k['n']),
                                'issued_16bit_equiv_tflops': round(issued, 2), 'issued_frac': round(issued / PEAK_BF16_TFLOPS, 4),
                                'fp8x_plan': {'launches': x8['n'], 'share_of_flops': round(x8['flops'] / k['flops'], 3), 'ms': round(x8['ms'], 3)},
                                'vs_fp32_peak_157.3': round(ach / PEAK_FP32_TFLOPS, 3),
                                'share_of_gpu_time': round(k['ms'] / sum(v['ms'] for v in prof.values()), 3)}
            line['kernel_time_ms_per_batch'] = {n: round(v['ms'], 3) for n, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
            # FLOPs the path actually issues per slice (the loop cache skips what depends on the condition images alone)
            flop_per_slice = sum(v['flops'] for v in prof.values()) / B
    per_gpu = value / world
    line['end_to_end'] = {'flop_per_slice_issued': round(flop_per_slice / 1e9, 1), 'flop_per_slice_reference_graph': round(FLOP_PER_SLICE / 1e9, 1),
                          'fp32_flop_frac': round(per_gpu * flop_per_slice / 1e12 / PEAK_FP32_TFLOPS, 4),
                          'x3_issued_frac': round(3 * per_gpu * flop_per_slice / 1e12 / PEAK_BF16_TFLOPS, 4),
                          'fused_hbm_frac': round(per_gpu * FUSED_BYTES_PER_SLICE / 1e9 / PEAK_HBM_GBS, 4)}

    extras = rank == 0 and world == 1 and not a.no_extras and not a.no_graph
    if extras:
        # PCIe-inclusive rate (SURVEY.md section 8(d) "Metric"): H2D of the three condition batches + sampling + D2H of the result
        hc = [c.cpu().pin_memory() for c in (c1, c2, c3)]
        hout = torch.empty(B, 1, H, H).pin_memory()

        def pcie_step():
            d = [c.to(dev, non_blocking=True) for c in hc]
            hout.copy_(sampler.sample(d[0], d[1], d[2], x_init, cfg.num_timesteps), non_blocking=True)
        pcie_step()
        tp = timed_batches(pcie_step, 3)
        line['pcie_inclusive'] = {'slices_per_s': round(3 * B / tp, 2), 'note': 'pinned host buffers; 3 x H2D + sample + D2H per batch inside the timed region'}
        del sampler
        # latency case of BASELINE config 2 read literally (batch = 1) and config 3's batch of 32, same path, own hipGraphs
        for nb_, key, iters in ((1, 'batch1', 20), (16, 'batch16', 5), (32, 'batch32', 3)):
            if nb_ == B:
                line[key] = {'slices_per_s': line['value'], 'note': 'the headline run'}
                continue
            cs = synthetic_batch(cfg, nb_, dev, seed=7)
            sb = S.GraphSampler(coef, g1, g2, cfg, nb_, H, H, dev)
            xb = torch.randn(nb_, 1, H, H, device=dev)
            fn = lambda: sb.sample(cs[0], cs[1], cs[2], xb, cfg.num_timesteps)    # noqa: E731
            fn(); fn()
            line[key] = {'slices_per_s': round(iters * nb_ / timed_batches(fn, iters), 2), 'timed_batches': iters,
                         'note': f'same path, {nb_} slice(s) per step' + (' (BASELINE config 2 read literally)' if nb_ == 1 else '')}
            del sb, cs, xb
        log('parity leg: config 2 and wide config 3 fixtures through the captured sampler ...')
        line['parity'] = parity_leg(cfg, dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log('timing the CPU oracle (2 warm-up + 3 timed slices, all cores' + (', then B=8 and 1 thread' if a.cpu_baseline_full else '') + ') ...')
        line['cpu_baseline'] = cpu_baseline(cfg, full=a.cpu_baseline_full)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.sweep:
        sys.exit(run_sweep(a, argv))
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # this process becomes the launcher: it never initialises the GPU
        rc, out0 = launch_ranks(a.gpus, argv)
        line = last_json_line(out0 or '')
        if line is not None:
            print(json.dumps(line), flush=True)
        elif rc == 0:
            rc = 1
            log('rank 0 printed no JSON line')
        sys.exit(rc)
    worker(a)


if __name__ == '__main__':
    main()
