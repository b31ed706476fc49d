#!/bin/bash
# build the library, then run a command on the MI355X box:  scripts/gpu.sh [timeout] 'cmd'
set -e
T=600
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
make -C /root/repo/mu-diff_amd/csrc -j8 2>&1 | grep -E "error|Error|warning: unused" && exit 1
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
