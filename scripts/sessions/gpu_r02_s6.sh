#!/bin/bash
# round-2 GPU session 6: fused skip conv - tests, bench A/B (MUD_FUSE_SKIP=0/1), layer table
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x > gpurun_out/r02/gpu_tests_6.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r02/gpu_tests_6.log
MUD_FUSE_SKIP=0 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r02/bench_6_nofuse.json 2> gpurun_out/r02/bench_6.err; echo "bench rc=$?"
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_6_fuse.json 2>> gpurun_out/r02/bench_6.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ('nofuse', 'fuse'):
    d = json.load(open(f'gpurun_out/r02/bench_6_{f}.json'))
    print(f, d['value'], d['roofline']['achieved'], d.get('batch1'), d.get('batch32'), d.get('parity', {}).get('max_abs_per_step'), d['kernel_time_ms_per_batch'])
P
python scripts/layer_times.py 16 > gpurun_out/r02/layer_times_b16_fuse.txt 2>&1; head -50 gpurun_out/r02/layer_times_b16_fuse.txt
