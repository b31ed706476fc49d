# fused skip conv split over K (in-launch reduction of both accumulator sets): parity, then one-slice A/B
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s27
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_k or fused_skip or deterministic" > gpurun_out/s27/t1.log 2>&1; rc=$?; tail -5 gpurun_out/s27/t1.log; [ $rc -eq 0 ] || exit $rc
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s27/bench.log | cut -c68-130; }
run MUD_FUSE_SKIP_SPLIT=0
run MUD_FUSE_SKIP_SPLIT=1
run MUD_FUSE_SKIP_SPLIT=0
run MUD_FUSE_SKIP_SPLIT=1
python scripts/layer_times.py 1 > gpurun_out/s27/layer_b1.txt 2>&1; head -14 gpurun_out/s27/layer_b1.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s27/tests.log 2>&1; tail -2 gpurun_out/s27/tests.log
