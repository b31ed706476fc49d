# CM_FP8X with the residual prefetch and the one-row 8-wave tile switched off (no spills outside the fused-skip variants): per layer, bench line fused / unfused skip
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s36
V=mu-diff_amd/mudiff_hip/variants
MUDIFF_HIP_LIB=$V/lib_fp8x2.so timeout -k 10 300 python scripts/check_fp8x.py > gpurun_out/s36/check.txt 2>&1; tail -7 gpurun_out/s36/check.txt
timeout -k 10 600 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fp8x2.so > gpurun_out/s36/ab.txt 2>&1; tail -19 gpurun_out/s36/ab.txt
for fs in 1 0; do echo "== MUD_FUSE_SKIP=$fs"; MUD_FUSE_SKIP=$fs MUDIFF_HIP_LIB=$V/lib_fp8x2.so timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/s36/bench_fs$fs.json 2> gpurun_out/s36/bench_fs$fs.log; cut -c1-120 gpurun_out/s36/bench_fs$fs.json; grep -o '"max_abs_per_step": \[[^]]*\]\|"batch1": {[^}]*}\|"achieved": [0-9.]*' gpurun_out/s36/bench_fs$fs.json; done
