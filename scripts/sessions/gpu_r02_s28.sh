# full GPU suite (with the two-real-ranks launcher test), then the end-of-round profile r02_m
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s28
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s28/tests.log 2>&1; rc=$?; tail -4 gpurun_out/s28/tests.log; [ $rc -eq 0 ] || exit $rc
bash scripts/gpu_profile.sh r02_m
