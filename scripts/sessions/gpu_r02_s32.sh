# scaled-fp8 cross-term experiment (CM_FP8X): numerics against fp64, shipped library beside it
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s32
timeout -k 10 300 python scripts/check_fp8x.py > gpurun_out/s32/check_base.txt 2>&1; tail -8 gpurun_out/s32/check_base.txt
MUDIFF_HIP_LIB=mu-diff_amd/mudiff_hip/variants/lib_fp8x.so timeout -k 10 300 python scripts/check_fp8x.py > gpurun_out/s32/check_fp8x.txt 2>&1; tail -8 gpurun_out/s32/check_fp8x.txt
