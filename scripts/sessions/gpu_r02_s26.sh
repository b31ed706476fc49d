# two half-batch lanes on two streams vs one full batch
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s26
timeout -k 10 600 python scripts/exp_two_lanes.py 16 > gpurun_out/s26/two_lanes_16.txt 2>&1; tail -4 gpurun_out/s26/two_lanes_16.txt
timeout -k 10 600 python scripts/exp_two_lanes.py 32 > gpurun_out/s26/two_lanes_32.txt 2>&1; tail -4 gpurun_out/s26/two_lanes_32.txt
