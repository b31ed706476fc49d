# in-launch split-K + side-by-side MLP chains: full GPU suite, default bench line, one-slice bench
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s24
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/s24/tests.log 2>&1; rc=$?; tail -3 gpurun_out/s24/tests.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline > gpurun_out/s24/bench.json 2> gpurun_out/s24/bench.log && cut -c1-2500 gpurun_out/s24/bench.json
python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s24/bench.log | cut -c68-130
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
