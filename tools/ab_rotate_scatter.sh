#!/bin/bash
# same-box A/B: the rotation's permutation as its own kernel (LSA_ROT_SCATTER=0) vs on the ModDown tail's store (default)
for rep in 1 2 3; do
for x in 0 1; do echo "== LSA_ROT_SCATTER=$x"; LSA_ROT_SCATTER=$x python bench.py --workload rotate --steps 16 --warmup 3 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
done
for x in 0 1; do echo "== conv task, LSA_ROT_SCATTER=$x"; LSA_ROT_SCATTER=$x python bench.py --workload task_conv --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py; done
