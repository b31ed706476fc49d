set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > $O/log.txt 2>&1; echo rc=$?
tail -2 $O/log.txt
python3 $R/scripts/trace_gaps.py $O 1600
