#!/bin/bash
# A/B of NTT kernel build variants on the GPU box: "<threads> <max_rho> <waves> <tiles_per_wg> <tau>"
for cfg in "$@"; do
  set -- $cfg
  LSA_EXTRA_FLAGS="-DLSA_NTT_THREADS=$1 -DLSA_NTT_MAX_RHO=$2 -DLSA_NTT_WAVES=$3 -DLSA_NTT_TILES_PER_WG=$4 -DLSA_NTT_TAU=$5" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== threads=$1 rho=$2 waves=$3 K=$4 tau=$5"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('hmult', round(d['value'],1), round(d['ms_per_step'],1), {k:(round(v['est_ms_per_step'],1), round(v['achieved_GBps'])) for k,v in d['kernel_breakdown'].items() if k=='k_ntt_pass'})"
  python tools/probe_engines.py 2>/dev/null | tr -d '\n ' ; echo
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
