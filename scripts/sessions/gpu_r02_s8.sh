#!/bin/bash
# round-2 GPU session 8: batch-1 micro-experiments (head conv blocks per image, residual prefetch in the 4-wave tiles)
set -o pipefail
mkdir -p gpurun_out/r02
V=mu-diff_amd/mudiff_hip/variants
for cap in 2048 256 128 64; do
  echo "== MUD_HEAD_CAP=$cap"
  MUD_HEAD_CAP=$cap python - <<'P'
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import torch
from mudiff_hip import ops
dev = 'cuda:0'
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (1, 2, 4):
    x = ops.View(torch.randn(B, 256, 256, 1, device=dev), B, 256, 256, 1)
    w = ops.direct_weight(torch.randn(64, 1, 3, 3, device=dev)); bias = torch.randn(64, device=dev)
    arena = ops.StatsArena(dev); outs = ops.View.empty(B, 256, 256, 64, dev, arena)
    print(f'B={B} head + stats {timeit(lambda: ops.conv(x, w, 3, 64, mfma=False, bias=bias, out=outs)):.1f} us')
P
done
for lib in base preres; do
  MUDIFF_HIP_LIB=$V/lib_$lib.so python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib batch1', d['value'])"
done
for lib in base preres; do
  MUDIFF_HIP_LIB=$V/lib_$lib.so python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib batch1', d['value'])"
done
