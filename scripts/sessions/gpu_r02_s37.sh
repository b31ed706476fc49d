# CM_FP8X: which tile for the 64->64 layers once the one-row 8-wave tile is out (16x1 without prefetch against the 4-wave two-row tile)
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s37
V=mu-diff_amd/mudiff_hip/variants
echo "== 16x1"; AB_SHAPES=10,11,13 timeout -k 10 300 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fp8x2.so 2>&1 | grep -v amdgpu | tee gpurun_out/s37/ab_16x1.txt
echo "== MUD_CONV_NO16=1 (4-wave two-row tile)"; MUD_CONV_NO16=1 AB_SHAPES=10,11,13 timeout -k 10 300 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fp8x2.so 2>&1 | grep -v amdgpu | tee gpurun_out/s37/ab_mt2.txt
