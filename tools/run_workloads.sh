#!/bin/bash
# every bench workload once on the GPU box; one JSON line each into gpurun_out/workloads_<tag>.log, the summary table on stdout
# (redirect stdout to ANOTHER file, e.g. tools/run_workloads.sh r03 > gpurun_out/workloads_r03_table.log)
TAG=${1:-run}
OUT=gpurun_out/workloads_$TAG.log
: > $OUT
python bench.py 2> gpurun_out/err_ckks_hmult.log >> $OUT || exit 1
for w in rotate bfv_hmult deep deep17 ntt task_ckks task_bfv task_conv bootstrap; do
  python bench.py --workload $w --no-cpu-baseline 2> gpurun_out/err_$w.log >> $OUT || exit 1
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(d["config"]["workload"][:58].ljust(60), "%10.1f %s" % (d["value"], d["unit"]), "ms/step %.2f" % d["ms_per_step"],
          "ntt %.0f GB/s frac %.3f" % (r.get("achieved", 0), r.get("frac", 0)) if r else "")
PY
