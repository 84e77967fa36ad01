#!/bin/bash
# A/B of NTT kernel build variants on the GPU box: "<max_rho> <waves>" pairs
for cfg in "$@"; do
  set -- $cfg
  LSA_EXTRA_FLAGS="-DLSA_NTT_MAX_RHO=$1 -DLSA_NTT_WAVES=$2" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== rho=$1 waves=$2"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('hmult', round(d['value'],1), round(d['ms_per_step'],1), {k:(round(v['est_ms_per_step'],1), round(v['achieved_GBps'])) for k,v in d['kernel_breakdown'].items()})"
  python bench.py --workload ntt --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ntt2^14', round(d['value'],1))"
done
