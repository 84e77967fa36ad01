#!/bin/bash
# kernel trace of 2 headline steps; prints the NTT launches of the last operator tile (grid, us)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/err.log
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "lsa::" in r["Kernel_Name"]]
# one operator tile = the launches between two k_tensor launches
idx = [i for i, r in enumerate(rows) if "k_tensor" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
for r in rows[a:b]:
    print(r["Kernel_Name"][5:40].ljust(36), r["Grid_Size_X"].rjust(9), r["Grid_Size_Y"].rjust(4), r["Grid_Size_Z"].rjust(3),
          "%8.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
