#!/bin/bash
# A/B of the whole-limb (single-pass) NTT plan for N = 2^13 / 2^14: never (LSA_NTT_WIDE=0) / per launch (unset) / always (1)
set -e
for w in 0 auto 1; do
  if [ $w = auto ]; then unset LSA_NTT_WIDE; else export LSA_NTT_WIDE=$w; fi
  python tools/probe_wide.py 2>/dev/null
  for wl in bfv_hmult task_bfv task_conv; do
    echo "== LSA_NTT_WIDE=$w $wl"
    python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown') or {}
print(round(d['value'],1), d['unit'], round(d['ms_per_step'],2), {k:(round(v['est_ms_per_step'],2), round(v['achieved_GBps'])) for k,v in kb.items()})"
  done
done
