#!/bin/bash
# same-box A/B of NTT builds, interleaved twice so that drift shows: staged kernel (LSA_NTT_R16=0), product build, prebuilt variants
# usage: tools/ab_r16.sh "<variant names>" ["<workloads>"]
for rep in 1 2; do
echo "== staged (LSA_NTT_R16=0)"; LSA_NTT_R16=0 python tools/probe_engines.py 2>/dev/null | tr -d '\n '; echo
for wl in ${2:-ckks_hmult}; do LSA_NTT_R16=0 python bench.py --workload $wl --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
for v in default $1; do
  if [ "$v" = default ]; then unset LSA_NATIVE_LIB; else export LSA_NATIVE_LIB=lattisense_amd/variants/lib$v.so; fi
  echo "== $v"; python tools/probe_engines.py 2>/dev/null | tr -d '\n '; echo
  for wl in ${2:-ckks_hmult}; do python bench.py --workload $wl --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
done
unset LSA_NATIVE_LIB
done
