# in-launch split-K reduce: parity tests, then B=1 numbers
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s20
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_k or fused_skip or deterministic" > gpurun_out/s20/t1.log 2>&1; rc=$?; tail -5 gpurun_out/s20/t1.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s20/t2.log 2>&1; rc=$?; tail -3 gpurun_out/s20/t2.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline > gpurun_out/s20/bench.json 2> gpurun_out/s20/bench.log && cat gpurun_out/s20/bench.json
MUD_CONV_SPLITK_2LAUNCH=1 python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline > gpurun_out/s20/bench_b1_2launch.json 2>> gpurun_out/s20/bench.log && cat gpurun_out/s20/bench_b1_2launch.json
python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline > gpurun_out/s20/bench_b1.json 2>> gpurun_out/s20/bench.log && cat gpurun_out/s20/bench_b1.json
python scripts/layer_times.py 1 > gpurun_out/s20/layer_b1.txt 2>&1; head -30 gpurun_out/s20/layer_b1.txt
