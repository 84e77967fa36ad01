#!/bin/bash
# headline bench at several (warmup, steps): does the reported rate depend on how long the GPU has been busy?  (it does not)
for cfg in "2 10" "5 20" "40 20" "120 20" "5 100" "5 20"; do set -- $cfg; echo -n "warmup $1 steps $2: "; python bench.py --warmup $1 --steps $2 --no-cpu-baseline 2>/dev/null | python tools/summarize_line.py | cut -c1-110; done
