#!/bin/bash
# One parameterised GPU session (replaces the per-session one-off scripts; scripts/sessions/README.md lists what every past
# session ran and where its result lives):
#     gpurun --timeout 1200 -- 'bash scripts/gpu_session.sh TAG step [step ...]'
# Output goes to gpurun_out/TAG/.  Steps (each bounded by its own timeout; a failed step stops the session):
#   tests            the whole -m gpu suite            tests-x           ... stopping at the first failure
#   bench[:plan]     python bench.py (no CPU baseline) with MUD_PREC_PLAN=plan (default: the library default)
#   bench-full       python bench.py as the driver runs it (CPU baseline included)
#   bench1           python bench.py --batch 1 (one slice at a time, 20 steps)
#   abprec[:B]       scripts/ab_prec.py B 5            per-shape A/B of the arithmetic plans of the 3x3 kernel
#   parity[:plan]    scripts/parity_full.py            per-step max-abs on every full-size fixture
#   layers[:B]       scripts/layer_times.py B          per-layer kernel times of one G1 + G2 pass
#   denorm           scripts/exp/mfma_denorm_probe     (build it first: see the .hip file's header)
#   profile          scripts/gpu_profile.sh TAG        rocprofv3 kernel stats + FETCH/WRITE passes + layer tables + bench line
#   widening         scripts/bench_widening.py         critic / volume rows
#   run:<script>     python scripts/<script>.py        (anything else under scripts/)
set -o pipefail
TAG=${1:?tag}; shift
O=gpurun_out/$TAG
mkdir -p $O
run() { echo "== $*" >&2; "$@"; local rc=$?; [ $rc -eq 0 ] || { echo "step failed (rc $rc): $*" >&2; exit $rc; }; }
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $name in
    tests)    timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; rc=$?; tail -n 8 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    tests-x)  timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; rc=$?; tail -n 8 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc ;;
    bench)    if [ -n "$arg" ]; then export MUD_PREC_PLAN=$arg; fi
              run timeout -k 10 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench${arg:+_$arg}.json 2> $O/bench${arg:+_$arg}.err
              unset MUD_PREC_PLAN; python -c "import json,sys; d=json.load(open('$O/bench${arg:+_$arg}.json')); print('bench${arg:+ $arg}:', d['value'], 'slices/s; batch1', d.get('batch1', {}).get('slices_per_s'), '; roofline', d.get('roofline', {}).get('achieved'), 'TF; parity', d.get('parity', {}).get('max_abs_per_step'), d.get('parity', {}).get('config3_wide', {}).get('max_abs_per_step'))" ;;
    bench-full) run timeout -k 10 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; cat $O/bench_full.json ;;
    bench1)   run timeout -k 10 300 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_b1.json 2> $O/bench_b1.err; cat $O/bench_b1.json ;;
    abprec)   run timeout -k 10 400 python scripts/ab_prec.py ${arg:-16} 5 > $O/ab_prec_b${arg:-16}.txt 2>&1; tail -n 3 $O/ab_prec_b${arg:-16}.txt ;;
    parity)   if [ -n "$arg" ]; then export MUD_PREC_PLAN=$arg; fi
              run timeout -k 10 500 python scripts/parity_full.py > $O/parity${arg:+_$arg}.txt 2>&1; unset MUD_PREC_PLAN; grep per-step $O/parity${arg:+_$arg}.txt ;;
    layers)   run timeout -k 10 300 python scripts/layer_times.py ${arg:-16} > $O/layer_times_b${arg:-16}.txt 2>&1; head -n 3 $O/layer_times_b${arg:-16}.txt ;;
    denorm)   run scripts/exp/mfma_denorm_probe > $O/denorm.txt 2>&1; cat $O/denorm.txt ;;
    profile)  run bash scripts/gpu_profile.sh $TAG ;;
    widening) run timeout -k 10 600 python scripts/bench_widening.py > $O/widening.txt 2>&1; tail -n 8 $O/widening.txt ;;
    run)      run timeout -k 10 600 python scripts/$arg.py > $O/$arg.txt 2>&1; tail -n 12 $O/$arg.txt ;;
    *) echo "unknown step $step" >&2; exit 2 ;;
  esac
done
