#!/bin/bash
# round-2 GPU session 9: what-if timing probes of the 3x3 conv (wrong results by design): which part bounds which layer?
set -o pipefail
mkdir -p gpurun_out/r02
V=mu-diff_amd/mudiff_hip/variants
python scripts/ab_conv.py 16 4 $V/lib_base.so $V/lib_w1.so $V/lib_w2.so $V/lib_w3.so $V/lib_w4.so $V/lib_w5.so > gpurun_out/r02/ab_whatif_b16.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/ab_whatif_b16.txt
AB_SHAPES=5,8,10,11,13 python scripts/ab_conv.py 1 4 $V/lib_base.so $V/lib_w1.so $V/lib_w2.so $V/lib_w3.so $V/lib_w4.so $V/lib_w5.so > gpurun_out/r02/ab_whatif_b1.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/ab_whatif_b1.txt
