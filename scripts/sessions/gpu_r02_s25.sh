# MFMA issue-order A/B (CM_ORDER 1 / 2 against the shipped order), 18 layer shapes at 16 slices
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s25
V=mu-diff_amd/mudiff_hip/variants
timeout -k 10 600 python scripts/ab_conv.py 16 7 $V/lib_base.so $V/lib_ord1.so $V/lib_ord2.so > gpurun_out/s25/ab_order.txt 2>&1; tail -24 gpurun_out/s25/ab_order.txt
