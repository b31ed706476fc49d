#!/bin/bash
# round-2 GPU session 16: SQ counters of the 3x3 conv (five layer shapes) and of the pair attention kernel
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES -d $O/sq1 -- python3 $R/scripts/pmc_conv.py 16 > $O/sq1.log 2>&1; echo "sq1 rc=$?"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES -d $O/sq2 -- python3 $R/scripts/pmc_conv.py 16 > $O/sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
python scripts/pmc_summary.py $O/sq1 k_ > $O/sq1_summary.txt 2>&1; python scripts/pmc_summary.py $O/sq2 k_ > $O/sq2_summary.txt 2>&1
head -60 $O/sq1_summary.txt; tail -5 $O/sq2.log
