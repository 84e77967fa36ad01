#!/bin/bash
# A/B of build flag sets over several workloads: tools/ab_build.sh "<flags A>" "<flags B>" ...   (restores the default build)
for cfg in "$@"; do
  LSA_EXTRA_FLAGS="$cfg" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== flags: [$cfg]"
  for wl in ckks_hmult ckks_hmult rotate bfv_hmult deep; do
    python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
  done
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
