#!/bin/bash
# same-box A/B of the bootstrap's plaintext-MAC workgroup order: batch-fastest without / with the XCD deal
for rep in 1 2; do
echo "== LSA_MACM_NO_XCD=1"
LSA_MACM_NO_XCD=1 python bench.py --workload bootstrap --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
LSA_MACM_NO_XCD=1 python bench.py --workload bootstrap --log-slots 11 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
echo "== default"
python bench.py --workload bootstrap --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
python bench.py --workload bootstrap --log-slots 11 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
done
