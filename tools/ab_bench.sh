#!/bin/bash
# same-box A/B of prebuilt library variants on bench workloads: tools/ab_bench.sh "default base grid1d" "ckks_hmult ntt"
for v in $1; do
  for wl in ${2:-ckks_hmult}; do
    if [ "$v" = default ]; then unset LSA_NATIVE_LIB; else export LSA_NATIVE_LIB=lattisense_amd/variants/lib$v.so; fi
    echo -n "$v: "; python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
  done
done
