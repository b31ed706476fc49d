# split-K policy sweep, second pass: per-layer tables for blocks <= 256
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s23
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --batch 1 --no-extras --no-roofline 2>> gpurun_out/s23/bench.log | cut -c68-130; }
run A=0
run MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=512 MUD_SPLITK_MINCHUNKS=8
run MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=768 MUD_SPLITK_MINCHUNKS=8
run MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=1024 MUD_SPLITK_MINCHUNKS=8
run MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=1024 MUD_SPLITK_MINCHUNKS=12
run A=0
MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=512 MUD_SPLITK_MINCHUNKS=8 python scripts/layer_times.py 1 > gpurun_out/s23/l_256_512_8.txt 2>&1
MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=768 MUD_SPLITK_MINCHUNKS=8 python scripts/layer_times.py 1 > gpurun_out/s23/l_256_768_8.txt 2>&1
MUD_SPLITK_MAXBLOCKS=256 MUD_SPLITK_TARGET=1024 MUD_SPLITK_MINCHUNKS=8 python scripts/layer_times.py 1 > gpurun_out/s23/l_256_1024_8.txt 2>&1
python scripts/layer_times.py 1 > gpurun_out/s23/l_base.txt 2>&1
