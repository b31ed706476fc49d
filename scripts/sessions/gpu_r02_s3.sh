#!/bin/bash
# round-2 GPU session 3: full GPU tests, in-process A/B of the stagger / priority variants of the 3x3 conv, bench
set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -q > gpurun_out/r02/gpu_tests_3.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02/gpu_tests_3.log
V=mu-diff_amd/mudiff_hip/variants
python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_stagger.so $V/lib_prio1.so $V/lib_prio2.so $V/lib_stagprio.so > gpurun_out/r02/ab_conv_1.txt 2>&1; echo "ab rc=$?"
cat gpurun_out/r02/ab_conv_1.txt
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_3.json 2> gpurun_out/r02/bench_3.err; echo "bench rc=$?"
cat gpurun_out/r02/bench_3.json
