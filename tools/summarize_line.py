"""stdin: one bench.py JSON line -> value, ms per step and the per-kernel (est. ms per step, algorithmic GB/s)."""
import json
import sys

d = json.loads(sys.stdin.read())
kb = d.get("kernel_breakdown") or {}
print(d["config"].get("workload", "?")[:40], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2),
      {k: (round(v["est_ms_per_step"], 2), round(v["achieved_GBps"])) for k, v in kb.items()})
