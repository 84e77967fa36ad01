#!/bin/bash
# same-box A/B at N = 2^17: nine-stage second pass on the staged kernel (LSA_NTT_R8X3=0, LSA_KS_FUSED=0) / as three radix-8 groups
# per point (LSA_KS_FUSED=0) / that plus the fused second pass + key MAC for FP64-engine target limbs (default)
for rep in 1 2 3; do
echo "== staged second pass, unfused MAC"; LSA_NTT_R8X3=0 LSA_KS_FUSED=0 python bench.py --workload deep17 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
echo "== radix-8 x 3 second pass, unfused MAC"; LSA_KS_FUSED=0 python bench.py --workload deep17 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
echo "== default"; python bench.py --workload deep17 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
done
