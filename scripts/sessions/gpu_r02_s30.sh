# sampler reuse across volumes: volume tests, then the widened rows (critic, volume) timing
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s30
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "volume" > gpurun_out/s30/t.log 2>&1; rc=$?; tail -3 gpurun_out/s30/t.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python scripts/bench_widening.py > gpurun_out/s30/widening.txt 2>&1; tail -8 gpurun_out/s30/widening.txt
