for rep in 1 2 3; do
for x in 0 1; do echo "== LSA_KSMAC_XCD=$x"; LSA_KSMAC_XCD=$x python bench.py --steps 16 --warmup 3 --no-cpu-baseline --single-stream 2>/dev/null | python tools/summarize_line.py; done
done
