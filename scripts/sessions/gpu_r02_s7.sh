#!/bin/bash
# round-2 GPU session 7: deterministic switch, multi-stream head blocks at small batch (A/B on batch 1), full tests
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q -x > gpurun_out/r02/gpu_tests_7.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r02/gpu_tests_7.log
for br in 0 1; do
MUD_BRANCHES=$br python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/r02/bench_7_b1_branches$br.json 2> gpurun_out/r02/bench_7.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r02/bench_7_b1_branches$br.json')); print('branches=$br batch1', d['value'], d['ms_per_step'])"
done
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_7.json 2>> gpurun_out/r02/bench_7.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r02/bench_7.json')); print(d['value'], d['roofline']['achieved'], d['batch1'], d['batch32'], d['parity'])"
