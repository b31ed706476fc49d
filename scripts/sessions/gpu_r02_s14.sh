#!/bin/bash
# round-2 GPU session 14: chained MLP kernel: full tests + batch-1 / default bench
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x > gpurun_out/r02/gpu_tests_14.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02/gpu_tests_14.log
python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1', d['value'])"
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['kernel_time_ms_per_batch'])"
python bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager (no graph)', d['value'])"
