#!/bin/bash
# A/B of build flag sets on the raw NTT workloads (integer-engine BFV shape, and both engines via probe_wide): restores the default build
for cfg in "$@"; do
  LSA_EXTRA_FLAGS="$cfg" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== flags: [$cfg]"
  python bench.py --workload ntt --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
  LSA_NTT_WIDE=0 python tools/probe_wide.py 2>/dev/null
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
