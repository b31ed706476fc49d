#!/bin/bash
# round-2 GPU session 10: what the LDS fragment reads cost (stale-fragment timing probes; wrong results by design)
set -o pipefail
mkdir -p gpurun_out/r02
V=mu-diff_amd/mudiff_hip/variants
python scripts/ab_conv.py 16 4 $V/lib_base.so $V/lib_w6.so $V/lib_w7.so $V/lib_w8.so > gpurun_out/r02/ab_whatif_lds_b16.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/ab_whatif_lds_b16.txt
