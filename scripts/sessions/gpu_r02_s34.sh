# scaled-fp8 cross-term experiment, restructured (zero tap in the record padding, per-lane bases): numerics, per-layer A/B, bench line
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s34
V=mu-diff_amd/mudiff_hip/variants
MUDIFF_HIP_LIB=$V/lib_fp8x.so timeout -k 10 300 python scripts/check_fp8x.py > gpurun_out/s34/check_fp8x.txt 2>&1; tail -8 gpurun_out/s34/check_fp8x.txt
timeout -k 10 600 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fp8x.so > gpurun_out/s34/ab_fp8x.txt 2>&1; tail -20 gpurun_out/s34/ab_fp8x.txt
MUDIFF_HIP_LIB=$V/lib_fp8x.so timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/s34/bench_fp8x.json 2> gpurun_out/s34/bench_fp8x.log; cut -c1-160 gpurun_out/s34/bench_fp8x.json; grep -o '"max_abs_per_step": \[[^]]*\]\|"batch1": {[^}]*}' gpurun_out/s34/bench_fp8x.json
