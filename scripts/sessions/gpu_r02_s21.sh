# in-launch split-K reduce, second take: split-K parity, then B=1 with either path
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s21
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_k or fused_skip or deterministic" > gpurun_out/s21/t1.log 2>&1; rc=$?; tail -3 gpurun_out/s21/t1.log; [ $rc -eq 0 ] || exit $rc
MUD_CONV_SPLITK_2LAUNCH=1 python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s21/bench.log | cut -c1-160
python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s21/bench.log | cut -c1-160
MUD_CONV_SPLITK_2LAUNCH=1 python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s21/bench.log | cut -c1-160
python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s21/bench.log | cut -c1-160
python scripts/layer_times.py 1 > gpurun_out/s21/layer_b1.txt 2>&1; head -12 gpurun_out/s21/layer_b1.txt
