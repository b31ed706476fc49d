#!/bin/bash
# round-2 GPU session 5: timing probe of the 16x16x32 MFMA shape inside the 3x3 conv (results of that build are wrong by design)
set -o pipefail
mkdir -p gpurun_out/r02
V=mu-diff_amd/mudiff_hip/variants
python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fake16.so > gpurun_out/r02/ab_conv_shape16.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/ab_conv_shape16.txt
