#!/bin/bash
# PMC passes over tools/ntt_pmc.py (run on the GPU box); prints per-engine means of the last 6 launches of each group
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ntt_pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/ntt_pmc.py > /dev/null 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 $REPO/tools/ntt_pmc.py > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 $REPO/tools/ntt_pmc.py > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/p3 -- python3 $REPO/tools/ntt_pmc.py > /dev/null 2> $OUT/p3.err
cd $REPO && python3 tools/ntt_pmc_summary.py $OUT | tee $OUT/summary.txt
