#!/bin/bash
for flags in "$@"; do
  LSA_EXTRA_FLAGS="$flags" python lattisense_amd/build.py --force > /dev/null 2>&1
  echo "== flags=$flags"
  python tools/probe_engines.py 2>/dev/null | tr -d '\n ' ; echo
done
LSA_EXTRA_FLAGS="" python lattisense_amd/build.py --force > /dev/null 2>&1
