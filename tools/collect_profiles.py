#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_gpu.sh on the GPU
box) into small, committed summaries under profiles/:
   profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats
   profiles/<round>_<tag>_summary.json       per-launch averages of the PMC passes
   python tools/collect_profiles.py r01 <tag> [<tag> ...]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "flutl_composite_kernel"


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


def pmc_avg(d):
    acc = collections.defaultdict(list)
    for f in newest(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    for tag in tags:
        src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
        stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
        out = {"tag": tag}
        if stats:
            dst = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_kernel_stats.csv")
            shutil.copy(stats[0], dst)
            for r in csv.DictReader(open(stats[0])):
                if KERNEL in r["Name"]:
                    out["kernel_stats"] = {k: r[k] for k in
                                           ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
        bj = os.path.join(src, "bench_trace.json")
        if os.path.exists(bj):
            lines = [l for l in open(bj) if l.startswith("{")]
            if lines:
                out["bench_line_under_rocprof"] = json.loads(lines[-1])
        pm = {}
        for p in ("pmc_write", "pmc_fetch", "pmc_sq"):
            a, n = pmc_avg(os.path.join(src, p))
            pm.update(a)
        out["pmc_avg_per_launch"] = pm
        if "WRITE_SIZE" in pm and "FETCH_SIZE" in pm:
            # rocprofv3 reports both in KiB (MI355X_MICROARCH.md, HBM section).
            # WRITE_SIZE is exact for 16-byte-per-lane streaming stores; FETCH_SIZE
            # is calibrated only for wide (16 B/lane) reads, where it reads 1/2;
            # this kernel's reads are unaligned 4-byte mask windows and scalar
            # loads, so the read side is reported raw (uncorrected).
            out["hbm_traffic_bytes_per_launch"] = {
                "write": pm["WRITE_SIZE"] * 1024, "fetch_raw": pm["FETCH_SIZE"] * 1024,
                "note": "fetch uncalibrated for 4-byte accesses; x2 would be the "
                        "upper bound given by the wide-read correction"}
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_summary.json"), "w"),
                  indent=1)
        print(tag, json.dumps(out.get("kernel_stats")), json.dumps(out.get("hbm_traffic_bytes_per_launch")))


if __name__ == "__main__":
    main()
