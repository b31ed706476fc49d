# what would cheaper cross terms buy?  timing probes with 2 and 1.5 MFMAs per product (wrong results by design)
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s29
V=mu-diff_amd/mudiff_hip/variants
timeout -k 10 600 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_w9.so $V/lib_w10.so > gpurun_out/s29/whatif_mfma_count.txt 2>&1; tail -22 gpurun_out/s29/whatif_mfma_count.txt
