# scaled-fp8 cross-term experiment: per-layer A/B against the shipped kernel, then the whole bench line (parity leg included) on the variant library
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s33
V=mu-diff_amd/mudiff_hip/variants
timeout -k 10 600 python scripts/ab_conv.py 16 5 $V/lib_base.so $V/lib_fp8x.so > gpurun_out/s33/ab_fp8x.txt 2>&1; tail -20 gpurun_out/s33/ab_fp8x.txt
MUDIFF_HIP_LIB=$V/lib_fp8x.so timeout -k 10 900 python bench.py --no-cpu-baseline > gpurun_out/s33/bench_fp8x.json 2> gpurun_out/s33/bench_fp8x.log; cut -c1-400 gpurun_out/s33/bench_fp8x.json; grep -o '"parity": {.*' gpurun_out/s33/bench_fp8x.json | cut -c1-600
