# timing probe of the tap-packed scaled-fp8 cross terms (CM_WHATIF=11) against the shipped kernel and the 2-MFMA probe
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s31
V=mu-diff_amd/mudiff_hip/variants
timeout -k 10 600 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_w9.so $V/lib_w11.so > gpurun_out/s31/whatif_fp8_cross.txt 2>&1; tail -22 gpurun_out/s31/whatif_fp8_cross.txt
