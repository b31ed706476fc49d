#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats, FETCH_SIZE and WRITE_SIZE in separate --pmc passes, plus the default bench line.
#   gpurun -- 'bash scripts/gpu_profile.sh r02_d'   then   python scripts/summarize_profile.py r02_d gpurun_out/r02_d/stats gpurun_out/r02_d/fetch gpurun_out/r02_d/write gpurun_out/r02_d/bench.json
set -o pipefail
TAG=${1:-r02_x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; cat $O/bench.json
python scripts/layer_times.py 16 > $O/layer_times_b16.txt 2>&1
python scripts/layer_times.py 1 > $O/layer_times_b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $O/write.log 2>&1; echo "write rc=$?"
ls $O/stats/* | head -5; du -sh $O
