#!/bin/bash
# round-2 GPU session 1: tests, default bench, ATT probe, batch-1 layer table
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/gpu_tests_1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/gpu_tests_1.log
tail -3 gpurun_out/r02/gpu_tests_1.log
python bench.py > gpurun_out/r02/bench_1.json 2> gpurun_out/r02/bench_1.err; echo "bench rc=$?"
cat gpurun_out/r02/bench_1.json
python bench.py --gpus 1 --total-slices 64 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-roofline > gpurun_out/r02/bench_strong.json 2>> gpurun_out/r02/bench_1.err; echo "strong rc=$?"; cat gpurun_out/r02/bench_strong.json
python scripts/layer_times.py 1 > gpurun_out/r02/layer_times_b1_before.txt 2>&1; echo "layer_times rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --att --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02/att -- python3 $GRAFT_REPO_ROOT/scripts/one_conv.py 64 256 256 3 16 3 > $GRAFT_REPO_ROOT/gpurun_out/r02/att.log 2>&1; echo "att rc=$?"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/r02/att.log
