#!/bin/bash
# A/B of arbitrary extra build flags on the GPU box: each argument is one flag set; prints the headline bench breakdown
for cfg in "$@"; do
  LSA_EXTRA_FLAGS="$cfg" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== flags: $cfg"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('hmult', round(d['value'],1), round(d['ms_per_step'],1), {k:(round(v['est_ms_per_step'],2), round(v['achieved_GBps'])) for k,v in d['kernel_breakdown'].items()})"
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
