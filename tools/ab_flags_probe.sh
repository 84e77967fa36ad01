#!/bin/bash
# A/B of extra build flags: NTT GB/s by engine (tools/probe_engines.py) per flag set
for cfg in "$@"; do
  LSA_EXTRA_FLAGS="$cfg" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== flags: $cfg"
  python tools/probe_engines.py 2>/dev/null | tr -d '\n ' ; echo
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
