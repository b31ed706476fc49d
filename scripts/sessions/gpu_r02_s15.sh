#!/bin/bash
# round-2 GPU session 15: fast sigmoid in the MFMA conv epilogue: full tests + bench
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x > gpurun_out/r02/gpu_tests_15.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02/gpu_tests_15.log
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r02/bench_15.json; python -c "import json; d=json.load(open('gpurun_out/r02/bench_15.json')); print('default', d['value'], d['batch1'], d['batch32'], d['parity']['max_abs_per_step'], d['roofline']['achieved'])"
python scripts/layer_times.py 16 2>/dev/null | head -4
