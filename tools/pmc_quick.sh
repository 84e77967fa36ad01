#!/bin/bash
# one rocprofv3 PMC pass over a short bench run; prints mean counters per kernel.  usage: tools/pmc_quick.sh <tag> "<counters>" [bench args]
set -o pipefail
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT/pmc_sq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/err.log
cd $REPO && python3 tools/summarize_prof.py $OUT | grep -v "at::native\|rocclr\|k_to_mont"
