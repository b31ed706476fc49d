#!/bin/bash
# round-2 GPU session 11: channel-split pair attention kernel: tests + A/B against the one-wave-per-column kernel
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "attention or config2 or small_models or config5" > gpurun_out/r02/gpu_tests_11.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02/gpu_tests_11.log
for single in 1 0; do
MUD_ATT_SINGLE=$single python - <<'P'
import os, sys
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
for B, N, C in ((16, 4096, 256), (32, 4096, 256), (1, 4096, 256), (16, 1024, 128), (16, 256, 256)):
    H = int(N ** 0.5)
    qkv = ops.View(torch.randn(B, H, H, 3 * C, device=dev), B, H, H, 3 * C)
    t = timeit(lambda: ops.attention(qkv, C, C ** -0.5))
    print(f'MUD_ATT_SINGLE={os.environ["MUD_ATT_SINGLE"]} B={B} N={N} C={C}: {t:8.1f} us  {4.0 * B * N * N * C / t / 1e6:6.1f} TF')
P
done
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02/bench_11.json 2> gpurun_out/r02/bench_11.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r02/bench_11.json')); print(d['value'], d['kernel_time_ms_per_batch'])"
