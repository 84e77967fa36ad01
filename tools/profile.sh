#!/bin/bash
# rocprofv3 recipe (run on the GPU box through gpurun): kernel-trace stats, then PMC passes in separate runs.
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline --single-stream "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --no-cpu-baseline --single-stream "$@" > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py --no-cpu-baseline --single-stream "$@" > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --no-cpu-baseline --single-stream "$@" > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --no-cpu-baseline --single-stream "$@" > /dev/null 2> $OUT/pmc_write.err
cd $REPO && python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE) next to the algorithmic bytes -> gpurun_out/prof_<tag>/pmc_traffic_k_ntt_pass.json
python3 tools/pmc_traffic.py $OUT $OUT > $OUT/traffic.txt 2>&1; cat $OUT/traffic.txt
