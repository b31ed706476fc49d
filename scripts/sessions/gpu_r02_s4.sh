#!/bin/bash
# round-2 GPU session 4: is the 3x3 conv power (clock) limited?  same kernel on random vs all-zero operands + in-kernel clock
set -o pipefail
mkdir -p gpurun_out/r02
V=mu-diff_amd/mudiff_hip/variants
AB_SHAPES=0,5,8,10,13 python scripts/ab_conv.py 16 5 $V/lib_base.so > gpurun_out/r02/dvfs_random.txt 2>&1; echo "rc=$?"
AB_ZERO=1 AB_SHAPES=0,5,8,10,13 python scripts/ab_conv.py 16 5 $V/lib_base.so > gpurun_out/r02/dvfs_zero.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/dvfs_random.txt gpurun_out/r02/dvfs_zero.txt
MUDIFF_HIP_LIB=scripts/exp/libstamp.so python scripts/stamp_conv.py 16 > gpurun_out/r02/stamps_b16.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/stamps_b16.txt
MUDIFF_HIP_LIB=scripts/exp/libstamp.so python scripts/stamp_conv.py 1 > gpurun_out/r02/stamps_b1.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/stamps_b1.txt
