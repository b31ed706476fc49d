# round 3, session 4: GPU suite on the fp16-pieces build with the 'auto' plan as default; per-shape A/B incl. the 64->64 layers under fp8x (16 x 1 tile); bench
set -x
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03/s4_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/s4_gpu_tests.log
tail -n 6 gpurun_out/r03/s4_gpu_tests.log
timeout -k 10 300 python scripts/ab_prec.py 16 5 > gpurun_out/r03/s4_ab_prec_b16.txt 2>&1
grep "64->  64\|weighted" gpurun_out/r03/s4_ab_prec_b16.txt
for plan in auto all; do
  MUD_PREC_PLAN=$plan timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r03/s4_bench_$plan.json 2> gpurun_out/r03/s4_bench_$plan.err
done
grep -h -o '"value": [0-9.]*\|"batch1": {[^}]*}\|"max_abs_per_step": [^]]*]' gpurun_out/r03/s4_bench_*.json
