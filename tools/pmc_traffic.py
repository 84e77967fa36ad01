"""Turns a tools/profile.sh output directory into profiles/<round>/pmc_traffic_k_ntt_pass.json: HBM bytes per launch from the
rocprofv3 PMC passes (FETCH_SIZE doubled: gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section;
WRITE_SIZE exact; both in KB), per kernel and per k_ntt_pass variant, next to the algorithmic bytes bench.py sampled in the
same configuration, plus a hash of the kernel sources so bench.py can tell when the file no longer describes the build.
usage: python tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/r02 [workload] [batch]"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["lattisense_amd/csrc/kernels.hip", "lattisense_amd/csrc/ntt_core.h", "lattisense_amd/csrc/ntt_r16.h",
                  "lattisense_amd/csrc/modarith.h", "lattisense_amd/csrc/ops.hip"]
NTT_KERNELS = ("k_ntt_r16", "k_ntt_r8x3", "k_ntt_pass")   # the limb-transform passes: radix-16-squared (8- / 7-stage), three radix-8 groups (9-stage), the staged kernel


def sources_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


def counter(d, sub, name):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def short(n):
    n = n.replace("lsa::", "").replace("void ", "")
    return n.split("(")[0]


def main(d, out_dir, workload="ckks_hmult", batch=256):
    fetch, write = counter(d, "pmc_fetch", "FETCH_SIZE"), counter(d, "pmc_write", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if "at::" in k or "rocclr" in k:
            continue
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        kernels[short(k)] = {"launches": len(f), "fetch_size_kb_mean": sum(f) / len(f), "write_size_kb_mean": sum(w) / len(w),
                             "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1e3}
    ntt = {k: v for k, v in kernels.items() if k.startswith(NTT_KERNELS)}
    launches = sum(v["launches"] for v in ntt.values())
    weighted = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in ntt.values()) / max(launches, 1)
    bench = {}
    try:
        bench = json.loads(open(os.path.join(d, "bench_stats.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    alg = {k: v["achieved_GBps"] * 1e9 * v["avg_launch_us"] * 1e-6 for k, v in bench.get("kernel_breakdown", {}).items()}
    ratios = {}
    for name in ("k_baseconv", "k_ks_mac", "k_tensor"):
        vs = [v for k, v in kernels.items() if k.startswith(name)]
        if vs and name in alg:
            n = sum(v["launches"] for v in vs)
            counted = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in vs) / n
            ratios[name] = {"counter_bytes_per_launch": counted, "algorithmic_bytes_per_launch": alg[name], "ratio": counted / alg[name]}
    if "k_ntt_pass" in alg:
        ratios["k_ntt_pass"] = {"counter_bytes_per_launch": weighted, "algorithmic_bytes_per_launch": alg["k_ntt_pass"],
                                "ratio": weighted / alg["k_ntt_pass"]}
    res = {"kernel": "k_ntt_r16 + k_ntt_r8x3 + k_ntt_pass (every limb-transform pass, launch-weighted)", "workload": workload, "batch": int(batch), "variants": ntt,
           "hbm_bytes_per_launch": weighted, "other_kernels": {k: v for k, v in kernels.items() if not k.startswith(NTT_KERNELS)},
           "counter_vs_algorithmic": ratios,
           "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section) -> doubled; WRITE_SIZE exact; units KB",
           "kernel_sources_sha256": sources_hash(), "source": "tools/profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py " + workload}
    os.makedirs(out_dir, exist_ok=True)
    p = os.path.join(out_dir, "pmc_traffic_k_ntt_pass.json")
    json.dump(res, open(p, "w"), indent=1)
    print(json.dumps(ratios, indent=1))
    print("->", p)


if __name__ == "__main__":
    main(*sys.argv[1:])
