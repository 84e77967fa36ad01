#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run (full kernel names, so template variants stay apart)
# usage: tools/stats_quick.sh <tag> [bench args]   (LSA_NATIVE_LIB selects a variant library)
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/statsq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --no-cpu-baseline --single-stream --steps 4 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/err.log
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:26]:
    n = r["Name"].replace("lsa::", "").replace("void ", "")
    print("%-64s calls=%-5s avg_us=%-9.1f total_ms=%-9.2f pct=%s" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
