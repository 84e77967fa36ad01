for rep in 1 2; do
echo "== single hoisting (LSA_BT_DOUBLE_HOIST=0)"
LSA_BT_DOUBLE_HOIST=0 python bench.py --workload bootstrap --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
LSA_BT_DOUBLE_HOIST=0 python bench.py --workload bootstrap --log-slots 11 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
echo "== double hoisting (default)"
python bench.py --workload bootstrap --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
python bench.py --workload bootstrap --log-slots 11 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py
done
