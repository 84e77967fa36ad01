#!/bin/bash
# runs a probe script under the default build and each prebuilt variant library (lattisense_amd.build --variant):
#   tools/ab_variants.sh tools/probe_engines.py default copyonly computeonly notw
script=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = default ]; then python $script 2>/dev/null | tr -d '\n '; else LSA_NATIVE_LIB=lattisense_amd/variants/lib$v.so python $script 2>/dev/null | tr -d '\n '; fi
  echo
done
