#!/bin/bash
# round-2 GPU session 21: key splits of the attention kernel chosen by rounds / splits: attention tests, timings over batch sizes, bench at 4 / 12
set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "attention or config2 or config5 or small_models" 2>&1 | tail -3
python - <<'P'
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
dev = 'cuda:0'
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    qkv = ops.View(torch.randn(B, 64, 64, 768, device=dev), B, 64, 64, 768)
    t = timeit(lambda: ops.attention(qkv, 256, 256 ** -0.5))
    print(f'B={B:2d} N=4096 C=256: {t:8.1f} us  {4.0 * B * 4096 * 4096 * 256 / t / 1e6:6.1f} TF')
P
for b in 4 12; do python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, d['value'])"; done
