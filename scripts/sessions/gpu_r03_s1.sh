# round 3, session 1: the rewritten 3x3 kernel (precision plans as a template parameter, residual loaded into the accumulators,
# probes gone) - GPU suite, MFMA subnormal probe, per-shape plan A/B, parity of the plans on every full-size fixture
set -x
mkdir -p gpurun_out/r03
scripts/exp/mfma_denorm_probe > gpurun_out/r03/s1_denorm.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/s1_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03/s1_gpu_tests.log
tail -5 gpurun_out/r03/s1_gpu_tests.log
timeout -k 10 300 python scripts/ab_prec.py 16 5 > gpurun_out/r03/s1_ab_prec_b16.txt 2>&1
for plan in off all auto; do
  MUD_PREC_PLAN=$plan timeout -k 10 300 python scripts/parity_full.py > gpurun_out/r03/s1_parity_$plan.txt 2>&1
done
tail -4 gpurun_out/r03/s1_parity_*.txt gpurun_out/r03/s1_denorm.txt
