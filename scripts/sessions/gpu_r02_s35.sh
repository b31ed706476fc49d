# end of round: full GPU suite on the final default build, then the evidence pass r02_o
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s35
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/s35/tests.log 2>&1; rc=$?; tail -3 gpurun_out/s35/tests.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/gpu_profile.sh r02_o
