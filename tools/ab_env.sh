#!/bin/bash
# A/B of a runtime switch: tools/ab_env.sh VAR "v1 v2 ..." [workloads...]
var=$1; vals=$2; shift 2
wls=${@:-ckks_hmult rotate bfv_hmult deep}
for v in $vals; do
  echo "== $var=$v"
  for wl in $wls; do
    env $var=$v python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
  done
done
