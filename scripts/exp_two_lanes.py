"""Two half-batch samplers replayed on two streams against one full-batch sampler: do the HBM-bound launches of one lane hide
under the clock-limited convolutions of the other?    python scripts/exp_two_lanes.py [B]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
import bench
from mudiff_hip import sampling as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
cfg = bench.bench_config()
g1, g2 = bench.build_models(cfg, dev, 0, 1)
coef = S.Posterior_Coefficients(cfg, dev)
H, T = cfg.image_size, cfg.num_timesteps


def make(b, seed):
    c = bench.synthetic_batch(cfg, b, dev, seed=seed)
    x = torch.randn(b, 1, H, H, device=dev)
    return S.GraphSampler(coef, g1, g2, cfg, b, H, H, dev), c, x


def timeit(fn, n=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


full, cf, xf = make(B, 100)
t_full = timeit(lambda: full.sample(*cf, xf, T))
print(f'one sampler, batch {B}: {B / t_full:.2f} slices/s')
lanes = [make(B // 2, 100 + i) for i in range(2)]
streams = [torch.cuda.Stream() for _ in lanes]


def two():
    cur = torch.cuda.current_stream()
    for st, (smp, c, x) in zip(streams, lanes):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            smp.sample(*c, x, T)
    for st in streams:
        cur.wait_stream(st)


t_two = timeit(two)
print(f'two samplers of batch {B // 2} on two streams: {B / t_two:.2f} slices/s ({t_full / t_two:.3f}x)')
t_seq = timeit(lambda: [smp.sample(*c, x, T) for smp, c, x in lanes])
print(f'the same two samplers one after the other: {B / t_seq:.2f} slices/s')
