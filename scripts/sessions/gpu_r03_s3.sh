# round 3, session 3: fp16 pieces everywhere (conv + attention), fused skip conv under the fp8 cross-term plan (no spills: rolled term loop)
set -x
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r03/s3_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/s3_gpu_tests.log
tail -n 6 gpurun_out/r03/s3_gpu_tests.log
timeout -k 10 300 python scripts/ab_prec.py 16 5 > gpurun_out/r03/s3_ab_prec_b16.txt 2>&1
for plan in off auto all; do
  MUD_PREC_PLAN=$plan timeout -k 10 400 python scripts/parity_full.py > gpurun_out/r03/s3_parity_$plan.txt 2>&1
done
grep -h "per-step" gpurun_out/r03/s3_parity_*.txt
for plan in off auto all; do
  MUD_PREC_PLAN=$plan timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/s3_bench_$plan.json 2> gpurun_out/r03/s3_bench_$plan.err
done
grep -h -o '"value": [0-9.]*\|"batch1": {[^}]*}\|"batch16": {[^}]*}\|"max_abs_per_step": [^]]*]' gpurun_out/r03/s3_bench_*.json
