#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile_cmd.sh / profile_gpu.sh on the
GPU box) into small, committed summaries under profiles/:
   profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats
   profiles/<round>_<tag>_summary.json       per kernel: trace stats + per-launch averages of
                                             the PMC passes
   python tools/collect_profiles.py r04 <tag> [<tag> ...]
A tag's kernels: the compositor's for the bench.py tags, the scan converter's two for `scan`.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"scan": ["scan_rows_kernel", "scan_prepare_kernel"]}
DEFAULT = ["flutl_composite_kernel"]


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:] if files else []


def pmc_avg(d, kernel):
    acc = collections.defaultdict(list)
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # (the first launches of a run are warm-up: bench.py's own, the scan tool's 20)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    rnd, tags = sys.argv[1], sys.argv[2:]
    for tag in tags:
        src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
        stats = newest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
        out = {"tag": tag, "kernels": {}}
        if stats:
            shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{rnd}_{tag}_kernel_stats.csv"))
        for kern in KERNELS.get(tag, DEFAULT):
            k = {}
            if stats:
                for r in csv.DictReader(open(stats[0])):
                    if kern in r["Name"]:
                        k["kernel_stats"] = {c: r[c] for c in ("Calls", "AverageNs", "MinNs", "MaxNs", "Percentage")}
            pm, cnt = {}, {}
            for p in ("pmc_write", "pmc_fetch", "pmc_sq", "pmc_sq2"):
                a, n = pmc_avg(os.path.join(src, p), kern)
                pm.update(a)
                cnt.update(n)
            k["pmc_avg_per_launch"] = pm
            k["pmc_launches"] = cnt
            if "WRITE_SIZE" in pm and "FETCH_SIZE" in pm:
                # rocprofv3 reports both in KiB (MI355X_MICROARCH.md, HBM section).  WRITE_SIZE is
                # exact for 16-byte-per-lane streaming stores; FETCH_SIZE is calibrated only for
                # wide (16 B/lane) reads, where it reads 1/2: narrower reads are reported raw.
                k["hbm_traffic_bytes_per_launch"] = {
                    "write": pm["WRITE_SIZE"] * 1024, "fetch_raw": pm["FETCH_SIZE"] * 1024,
                    "note": "fetch uncalibrated for accesses narrower than 16 B per lane; x2 is the "
                            "upper bound given by the wide-read correction"}
            if "TCC_HIT_sum" in pm and pm.get("TCC_HIT_sum", 0) + pm.get("TCC_MISS_sum", 0) > 0:
                k["l2_hit_rate"] = pm["TCC_HIT_sum"] / (pm["TCC_HIT_sum"] + pm["TCC_MISS_sum"])
            if pm.get("SQ_WAVE_CYCLES"):
                k["wait_any_of_wave_cycles"] = pm.get("SQ_WAIT_ANY", 0) / pm["SQ_WAVE_CYCLES"]
            out["kernels"][kern] = k
        # (what bench.py's profiled_traffic() and the older summaries' readers look for)
        first = out["kernels"].get(KERNELS.get(tag, DEFAULT)[0], {})
        for key in ("kernel_stats", "pmc_avg_per_launch", "hbm_traffic_bytes_per_launch"):
            if key in first:
                out[key] = first[key]
        bj = os.path.join(src, "bench_trace.json")
        if os.path.exists(bj):
            lines = [l for l in open(bj) if l.startswith("{")]
            if lines:
                out["line_under_rocprof"] = json.loads(lines[-1])
            else:
                txt = [l.strip() for l in open(bj) if "us per conversion" in l]
                if txt:
                    out["line_under_rocprof"] = txt[-1]
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_summary.json"), "w"), indent=1)
        print(tag, {k: (v.get("kernel_stats") or {}).get("AverageNs") for k, v in out["kernels"].items()},
              json.dumps(out.get("hbm_traffic_bytes_per_launch")))


if __name__ == "__main__":
    main()
