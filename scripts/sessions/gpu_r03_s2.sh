# round 3, session 2: parity of the arithmetic plans on every full-size fixture (shipped bf16 pieces vs an fp16-pieces build), GPU suite
set -x
mkdir -p gpurun_out/r03
for plan in off all auto; do
  MUD_PREC_PLAN=$plan timeout -k 10 400 python scripts/parity_full.py > gpurun_out/r03/s2_parity_$plan.txt 2>&1
done
for plan in off auto; do
  MUDIFF_ALLOW_VARIANT=1 MUDIFF_HIP_LIB=mu-diff_amd/mudiff_hip/variants/lib_f16.so MUD_PREC_PLAN=$plan timeout -k 10 400 python scripts/parity_full.py > gpurun_out/r03/s2_parity_f16_$plan.txt 2>&1
done
grep -h "per-step\|library" gpurun_out/r03/s2_parity_*.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03/s2_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/s2_gpu_tests.log
tail -n 15 gpurun_out/r03/s2_gpu_tests.log
