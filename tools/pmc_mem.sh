#!/bin/bash
# memory-path counters of the headline step per kernel (separate rocprofv3 --pmc passes, kernel means):
#   tools/pmc_mem.sh <tag> [bench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# at most two counters of one block per pass (more: "exceeds the capabilities of the hardware", and the aborted tool hangs)
for set in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > /dev/null 2> $OUT/p$i.err || echo "pass $i failed: $set"
done
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "lsa::" not in k and not k.startswith("k_"): continue
        k = k.split("(")[0].replace("void lsa::", "").replace("lsa::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k[:28].ljust(30), "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), " n=%d" % len(next(iter(acc[k].values()))))
    print()
PY
