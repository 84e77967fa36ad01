#!/bin/bash
# shader / memory clocks and package power while a bench workload runs (read-only rocm-smi queries): is the chip clock- or
# power-limited under this kernel mix?   usage: tools/clocks_under_load.sh [workload] [steps]
WL=${1:-ckks_hmult}; STEPS=${2:-300}
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6
python bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6
  sleep 0.7
done
wait $BP
python tools/summarize_line.py < /tmp/clk_bench.json | cut -c1-120
