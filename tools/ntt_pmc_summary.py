import csv, glob, sys, collections
out = sys.argv[1]
def rows(sub, pat):
    fs = glob.glob(f"{out}/{sub}/**/*{pat}*.csv", recursive=True)
    r = []
    for f in fs:
        r += list(csv.DictReader(open(f)))
    return r
tr = [r for r in rows("trace", "kernel_trace") if "k_ntt_pass" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
print("launch us:", [round(x) for x in d])
for sub in ("p1", "p2", "p3"):
    cs = [r for r in rows(sub, "counter_collection") if "k_ntt_pass" in r["Kernel_Name"]]
    per = collections.defaultdict(dict)
    for r in cs:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]) + per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0)
    ids = sorted(per)
    half = len(ids) // 2
    for name, sel in (("fp64", ids[2:half]), ("int", ids[half + 2:])):
        if not sel:
            continue
        keys = sorted(per[sel[0]])
        print(sub, name, {k: "%.4g" % (sum(per[i][k] for i in sel) / len(sel)) for k in keys})
