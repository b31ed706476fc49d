#!/bin/bash
# round-2 GPU session 12: rehearsal of the N-rank worker path on the one-GPU box (2 ranks share cuda:0, gloo instead of RCCL),
# launcher + torch.distributed.run forms, weak and strong scaling; then the --sweep line with the counts that fit (1)
set -o pipefail
mkdir -p gpurun_out/r02
export MUDIFF_BENCH_BACKEND=gloo MUDIFF_BENCH_SAME_GPU=1
python bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 > gpurun_out/r02/rehearsal_launcher_weak.json 2> gpurun_out/r02/rehearsal.err; echo "launcher weak rc=$?"; cat gpurun_out/r02/rehearsal_launcher_weak.json
python bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 --total-slices 20 > gpurun_out/r02/rehearsal_launcher_strong.json 2>> gpurun_out/r02/rehearsal.err; echo "launcher strong rc=$?"; cat gpurun_out/r02/rehearsal_launcher_strong.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 > gpurun_out/r02/rehearsal_torchrun.json 2>> gpurun_out/r02/rehearsal.err; echo "torchrun rc=$?"; cat gpurun_out/r02/rehearsal_torchrun.json
unset MUDIFF_BENCH_BACKEND MUDIFF_BENCH_SAME_GPU
python bench.py --sweep 1,2,4,8 --total-slices 64 --steps 2 --warmup 1 > gpurun_out/r02/sweep_1gpu.json 2>> gpurun_out/r02/rehearsal.err; echo "sweep rc=$?"; cat gpurun_out/r02/sweep_1gpu.json
tail -5 gpurun_out/r02/rehearsal.err
