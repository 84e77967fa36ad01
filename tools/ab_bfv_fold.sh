#!/bin/bash
# same-box A/B of BFV mult+relin: separate element-wise steps / copies (LSA_BFV_FOLD=0) vs folded into the conversions and transforms
for rep in 1 2 3; do
for x in 0 1; do echo "== LSA_BFV_FOLD=$x"; LSA_BFV_FOLD=$x python bench.py --workload bfv_hmult --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
done
for x in 0 1; do echo "== task_bfv, LSA_BFV_FOLD=$x"; LSA_BFV_FOLD=$x python bench.py --workload task_bfv --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
