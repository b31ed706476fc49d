"""Throughput of the widened rows on one MI355X: f1 critic forward (+ uncertainty map) and f3 volume pipeline."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import numpy as np
import torch
from oracle import mudiff_oracle as O                    # weights-from-seed only (bench helper, not a product path)
from backbones.discriminator import Discriminator_large, conv2d, uncertainty_map
from backbones.ncsnpp_generator_adagn_feat import NCSNpp, NCSNpp_adaptive
from mudiff_hip import volume as V

dev = 'cuda:0'
torch.manual_seed(0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---- f1: critic as the reference builds it (engine/train.py:448-466: nc=2, ngf=64, t_emb_dim=256), 256x256
B = 16
D = Discriminator_large(nc=2, ngf=64, t_emb_dim=256).to(dev).eval()
att = conv2d(64 * 8, 1, 1, padding=0).to(dev)
x, xt = torch.randn(B, 1, 256, 256, device=dev), torch.randn(B, 1, 256, 256, device=dev)
t = torch.randint(0, 4, (B,), device=dev)


def critic():
    out, mid = D(x, t, xt)
    return uncertainty_map(att, mid, (256, 256))


dt = timeit(critic)
print(f'f1 critic + uncertainty map: {dt * 1e3:.2f} ms per batch of {B} -> {B / dt:.1f} samples/s (ngf=64, 256x256)')

# ---- f3: one BraTS-sized volume (240x240x155 -> 155 slices resized to 256), config 2 generators
cfg = O.default_config()
g1, g2 = NCSNpp(cfg), NCSNpp_adaptive(cfg)
g1.load_state_dict(O.make_state_dict(cfg, 'g1', 1234)); g2.load_state_dict(O.make_state_dict(cfg, 'g2', 1234))
g1, g2 = g1.to(dev).eval(), g2.to(dev).eval()
rng = np.random.default_rng(0)
vols = [(rng.random((240, 240, 155)) * 1000 * (rng.random((240, 240, 155)) > 0.3)).astype(np.float64) for _ in range(3)]
t0 = time.perf_counter()
stacks = [np.stack(V.extract_center_slices(V.robust_minmax_to_minus1_1(v), 80)[0], 0) for v in vols]
t_pre = time.perf_counter() - t0
V.predict_slices(cfg, g1, g2, [s[:32] for s in stacks], dev, batch_size=32, seed=1)       # warm-up: graph capture, weight packing
torch.cuda.synchronize()
t0 = time.perf_counter()
pred = V.predict_slices(cfg, g1, g2, stacks, dev, batch_size=32, seed=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'f3 volume: {pred.shape[0]} slices 240x240 -> 256x256 in {dt:.2f} s ({pred.shape[0] / dt:.1f} slices/s incl. upload, resize, graph capture, '
      f'download); host normalisation of 3 volumes {t_pre:.2f} s')
