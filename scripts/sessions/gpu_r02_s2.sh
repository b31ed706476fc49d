#!/bin/bash
# round-2 GPU session 2: full GPU tests, bench, batch-1 layer table after GN fold + split-K
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q > gpurun_out/r02/gpu_tests_2.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02/gpu_tests_2.log
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_2.json 2> gpurun_out/r02/bench_2.err; echo "bench rc=$?"
cat gpurun_out/r02/bench_2.json
python scripts/layer_times.py 1 > gpurun_out/r02/layer_times_b1_splitk.txt 2>&1; echo "layer_times rc=$?"
head -40 gpurun_out/r02/layer_times_b1_splitk.txt
