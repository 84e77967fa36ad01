"""Summarise a tools/profile.sh output directory: per-kernel time stats and per-kernel mean PMC counters."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def short(name):
    n = name.split("(")[0]
    for fam in ("k_ntt_r16", "k_ntt_r8x3", "k_ntt_pass"):   # keep the variant (template arguments)
        if fam in n:
            return n[n.index(fam):]
    for k in ("k_ntt_pass", "k_baseconv", "k_ks_mac", "k_tensor", "k_sub_mul", "k_rescale_prep", "k_permute",
              "k_copy_rows", "k_elementwise", "k_to_mont"):
        if k in n:
            return k
    return n[-50:]


def main(d):
    for f in find(os.path.join(d, "stats"), "*kernel_stats.csv"):
        print("== kernel stats (%s)" % os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:14]:
            print("%-18s calls=%-7s total_ms=%-10.3f avg_us=%-9.3f pct=%s" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                r["Percentage"]))
        ntt = [r for r in rows if "k_ntt_pass" in r["Name"] or "k_ntt_r16" in r["Name"] or "k_ntt_r8x3" in r["Name"]]
        if ntt:   # both variants together: the figure bench.py's roofline.avg_launch_us is compared with
            calls = sum(int(r["Calls"]) for r in ntt)
            tot = sum(float(r["TotalDurationNs"]) for r in ntt)
            print("%-18s calls=%-7d total_ms=%-10.3f avg_us=%-9.3f (all variants)" % ("k_ntt_* (passes)", calls, tot / 1e6, tot / calls / 1e3))
    for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        files = find(os.path.join(d, sub), "*counter_collection.csv")
        if not files:
            continue
        agg = defaultdict(lambda: defaultdict(list))
        for f in files:
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("== %s (mean per dispatch)" % sub)
        for k, cs in agg.items():
            print("%-18s " % k + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())) +
                  "  n=%d" % len(next(iter(cs.values()))))


if __name__ == "__main__":
    main(sys.argv[1])
